"""CPU oracle for the DGP hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product path (deepgraphpose_amd/) never does.

It is a CPU restatement (numpy + torch-CPU fp32) of the reference algorithm on
the north-star path.  Every function cites the reference lines it follows
(paths relative to /root/reference; PET = src/DeepLabCut/deeplabcut/
pose_estimation_tensorflow, DGP = src/deepgraphpose).

PINNING STATUS
  * The conv / BN / pool / transposed-conv / softmax arithmetic of the
    reference lives in an un-vendored third-party dependency
    (tensorflow==1.13.1/1.15 + tensorflow.contrib.slim, README.md:37-38;
    call sites PET/nnet/pose_net.py:10,14-16,19-25,46-52,
    DGP/models/fitdgp_util.py:50-73,312,368).  TF cannot be imported here and the
    reference holds no golden vectors for it  ->  **parity unpinned** for
    A1-A3; they restate TF-1.15's published semantics and are checked by
    known-answer tests plus an independent naive restatement (oracle/dgp_naive.c).
    What CAN be pinned to TensorFlow is (round 6, tests/test_oracle_cpu.py "TF / TF-slim's
    OWN published known answers"): the expected matrices TF's own unit tests assert for
    slim.conv2d / resnet_utils.conv2d_same / subsample (resnet_v1_test.py
    testConv2DSameEven / Odd, testSubsample*), tf.nn.conv2d itself (conv_ops_test.py
    Conv2DTest: 1x1 / 2x2 / 1x2 filters, stride 2 VALID and SAME, kernel smaller than
    the stride, kernel = input size; tests/_tf_kat.py -- each vector is first re-derived
    from the definition in float64 loops), conv2d_transpose SAME stride 2
    (conv2d_transpose_test.py testConv2DTransposeSame), max-pool SAME (pooling_ops_test.py
    _testMaxPoolSamePadding), the stack_blocks_dense endpoint
    shapes, and slim's atrous invariant (output_stride 16 subsampled == nominal stride 32)
    -- on this oracle and, across implementations, through dgp_forward
    (tests/test_parity_gpu.py).  Padding, alignment and stride bookkeeping are therefore
    held to numbers TF itself is held to; the fp32 accumulation order of TF's kernels is not
    (no TF-produced tensor exists in this image).
  * The pure-numpy pieces (A6 argmax_pose_predict, B10 index helpers, locref
    targets) ARE pinned against outputs of the reference itself: see
    tests/golden/make_golden.py and tests/golden/*.npz.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

MEAN_PIXEL = (123.68, 116.779, 103.939)   # PET/default_config.py:23
BN_EPS = 1e-5                              # slim resnet_arg_scope() default epsilon


# ----------------------------------------------------------------------------
# TF padding arithmetic
# ----------------------------------------------------------------------------
def tf_same_pads(n: int, k: int, s: int, d: int = 1) -> Tuple[int, int, int]:
    """TF 'SAME': out = ceil(n/s); pad_total = max((out-1)*s + k_eff - n, 0);
    pad_before = pad_total // 2 (the extra pixel goes AFTER)."""
    k_eff = (k - 1) * d + 1
    out = -(-n // s)
    total = max((out - 1) * s + k_eff - n, 0)
    return out, total // 2, total - total // 2


def _to_nchw(x: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(x)).permute(0, 3, 1, 2)


def _t(a: np.ndarray, like: torch.Tensor) -> torch.Tensor:
    """numpy parameter -> torch tensor of the activation's dtype (fp32: no change; fp64 anchor: exact widening of the fp32 values)"""
    return torch.from_numpy(np.ascontiguousarray(a)).to(like.dtype)


def _to_nhwc(t: torch.Tensor) -> np.ndarray:
    return t.permute(0, 2, 3, 1).contiguous().numpy()


def conv2d(x: torch.Tensor, w_hwio: np.ndarray, stride: int = 1, rate: int = 1,
           padding: str = "SAME") -> torch.Tensor:
    """slim.conv2d without bias/BN/activation on an NCHW torch tensor.
    TF conv2d is a cross-correlation (no kernel flip), weights HWIO."""
    kh, kw = w_hwio.shape[:2]
    w = _t(w_hwio, x).permute(3, 2, 0, 1).contiguous()
    if padding == "SAME":
        _, pt, pb = tf_same_pads(x.shape[2], kh, stride, rate)
        _, pl, pr = tf_same_pads(x.shape[3], kw, stride, rate)
        x = F.pad(x, (pl, pr, pt, pb))
    return F.conv2d(x, w, stride=stride, dilation=rate)


def conv2d_same(x: torch.Tensor, w_hwio: np.ndarray, stride: int, rate: int = 1) -> torch.Tensor:
    """slim resnet_utils.conv2d_same: stride 1 -> SAME; else explicit symmetric-ish
    zero pad (k_eff-1 total, floor-half first) followed by a VALID conv."""
    k = w_hwio.shape[0]
    if stride == 1:
        return conv2d(x, w_hwio, 1, rate, "SAME")
    k_eff = k + (k - 1) * (rate - 1)
    pad_total = k_eff - 1
    pad_beg = pad_total // 2
    pad_end = pad_total - pad_beg
    x = F.pad(x, (pad_beg, pad_end, pad_beg, pad_end))
    return conv2d(x, w_hwio, stride, rate, "VALID")


def batch_norm(x: torch.Tensor, wts: Dict[str, np.ndarray], scope: str) -> torch.Tensor:
    """slim.batch_norm, is_training=False (PET/nnet/pose_net.py:52): moving stats,
    scale=True, eps 1e-5, evaluated as (x-mean) * (gamma * rsqrt(var+eps)) + beta."""
    g = _t(wts[scope + "/BatchNorm/gamma"], x)
    b = _t(wts[scope + "/BatchNorm/beta"], x)
    m = _t(wts[scope + "/BatchNorm/moving_mean"], x)
    v = _t(wts[scope + "/BatchNorm/moving_variance"], x)
    inv = g * torch.rsqrt(v + BN_EPS)
    return (x - m[None, :, None, None]) * inv[None, :, None, None] + b[None, :, None, None]


def max_pool_same(x: torch.Tensor, k: int, s: int) -> torch.Tensor:
    """slim.max_pool2d(padding='SAME') -- padded cells never win (-inf)."""
    _, pt, pb = tf_same_pads(x.shape[2], k, s)
    _, pl, pr = tf_same_pads(x.shape[3], k, s)
    x = F.pad(x, (pl, pr, pt, pb), value=float("-inf"))
    return F.max_pool2d(x, k, s)


def subsample(x: torch.Tensor, factor: int) -> torch.Tensor:
    """slim resnet_utils.subsample: max_pool2d([1,1], stride=factor)."""
    return x if factor == 1 else x[:, :, ::factor, ::factor]


# ----------------------------------------------------------------------------
# A1: backbone
# ----------------------------------------------------------------------------
def resnet_features(frames_u8: np.ndarray, wts: Dict[str, np.ndarray], depth: int = 50,
                    return_endpoints: bool = False, dtype=np.float32, output_stride: int = 16):
    """PoseNet.extract_features (PET/nnet/pose_net.py:36-54): (x - mean_pixel) ->
    slim resnet_v1_{depth}(global_pool=False, output_stride=16, is_training=False).
    dtype=np.float64: the ACCURACY ANCHOR -- the same fp32 parameters and the same graph evaluated in double precision (what every
    fp32 evaluation order, TF's included, approximates); the reference itself runs fp32.
    output_stride: 16 is what DLC / DGP build (pose_net.py:49); 32 is slim's nominal network -- only the TF-slim atrous-invariance
    known answer uses it (tests/test_oracle_cpu.py, resnet_v1_test.testAtrousFullyConvolutionalValues)."""
    from oracle.resnet_plan import units as resnet_units   # the oracle's OWN restatement of slim's plan
    name = "resnet_v1_%d" % depth
    x = frames_u8.astype(np.float32) - np.asarray(MEAN_PIXEL, dtype=np.float32)[None, None, None, :]
    x = _to_nchw(x.astype(dtype)).contiguous(memory_format=torch.channels_last)
    ends = {}
    with torch.no_grad():
        net = conv2d_same(x, wts[name + "/conv1/weights"], stride=2)
        net = F.relu(batch_norm(net, wts, name + "/conv1"))
        ends["conv1"] = net
        net = max_pool_same(net, 3, 2)
        ends["pool1"] = net
        for u in resnet_units(depth, output_stride):
            if u.has_shortcut_conv:
                sc = conv2d(net, wts[u.scope + "/shortcut/weights"], stride=u.stride, padding="SAME")
                sc = batch_norm(sc, wts, u.scope + "/shortcut")
            else:
                sc = subsample(net, u.stride)
            r = conv2d(net, wts[u.scope + "/conv1/weights"], 1)
            r = F.relu(batch_norm(r, wts, u.scope + "/conv1"))
            r = conv2d_same(r, wts[u.scope + "/conv2/weights"], u.stride, u.rate)
            r = F.relu(batch_norm(r, wts, u.scope + "/conv2"))
            r = conv2d(r, wts[u.scope + "/conv3/weights"], 1)
            r = batch_norm(r, wts, u.scope + "/conv3")
            net = F.relu(sc + r)
            ends[u.scope] = net
    out = _to_nhwc(net)
    if return_endpoints:
        return out, {k: _to_nhwc(v) for k, v in ends.items()}
    return out


# ----------------------------------------------------------------------------
# A2: transposed-conv heads
# ----------------------------------------------------------------------------
def conv2d_transpose_same(x_nhwc: np.ndarray, w: np.ndarray, b: Optional[np.ndarray],
                          stride: int = 2) -> np.ndarray:
    """slim.conv2d_transpose(k=3, stride, 'SAME') + bias (PET/nnet/pose_net.py:18-26,
    DGP/models/fitdgp_util.py:58-73).  Weights [kh,kw,Cout,Cin].  TF defines it as the
    gradient of the SAME/stride-s forward conv (pad 0 before / 1 after for k=3,s=2):
        y[o] = sum_{i,k : s*i + k - pad_before = o} x[i] * w[k]     (no flip)
    Output is exactly stride*H x stride*W."""
    n, h, wd, cin = x_nhwc.shape
    kh, kw, cout, cin2 = w.shape
    assert cin == cin2
    oh, ow = h * stride, wd * stride
    _, pt, _ = tf_same_pads(oh, kh, stride)
    _, pl, _ = tf_same_pads(ow, kw, stride)
    xt = _to_nchw(x_nhwc)
    wt = _t(w, xt).permute(3, 2, 0, 1).contiguous()  # [Cin,Cout,kh,kw]
    with torch.no_grad():
        full = F.conv_transpose2d(xt, wt, stride=stride)   # (H-1)*s + k
        y = full[:, :, pt:pt + oh, pl:pl + ow]
        if y.shape[2] < oh or y.shape[3] < ow:    # only when k < stride
            y = F.pad(y, (0, ow - y.shape[3], 0, oh - y.shape[2]))
        if b is not None:
            y = y + _t(b, xt)[None, :, None, None]
    return _to_nhwc(y)


def pose_heads(features: np.ndarray, wts: Dict[str, np.ndarray], with_locref: bool = False):
    scmap = conv2d_transpose_same(features, wts["pose/part_pred/block4/weights"],
                                  wts["pose/part_pred/block4/biases"])
    locref = None
    if with_locref:
        locref = conv2d_transpose_same(features, wts["pose/locref_pred/block4/weights"],
                                       wts["pose/locref_pred/block4/biases"])
    return scmap, locref


# ----------------------------------------------------------------------------
# A3: DGP 2-D soft-argmax
# ----------------------------------------------------------------------------
def gaussian_taps(sigma: float, truncate: float = 1.0, dtype=np.float32) -> np.ndarray:
    """make_gaussian_2d_kernel (DGP/models/fitdgp_util.py:281-286): radius=int(sigma*truncate)."""
    radius = int(sigma * truncate)
    x = np.arange(-radius, radius + 1).astype(dtype)
    k = np.exp(dtype(-0.5) * np.square(x / dtype(sigma)))
    return (k / k.sum(dtype=dtype)).astype(dtype)


def argmax_2d_from_cm(scmap: np.ndarray, gamma: float = 1.0, gauss_len: int = 2,
                      dtype=np.float32, th=None) -> Tuple[np.ndarray, np.ndarray]:
    """argmax_2d_from_cm (DGP/models/fitdgp_util.py:342-402); th: the thresholding branch (:377-388).

    scmap [N,H,W,C] -> (mu [N,C,2] as (row, col), normalised blurred softmax [N,H,W,C]).
    Steps: softmax over H*W of gamma*s; CONSTANT zero pad by gauss_len (:304-310);
    depthwise conv with outer(g,g) (:312, identity pointwise); divide by sum + 1e-100
    (1e-100 is 0 in fp32); expectation of the (h, w) grid (:318-339, :396)."""
    n, h, w, c = scmap.shape
    s = np.transpose(scmap.astype(dtype), (0, 3, 1, 2)).reshape(n * c, h * w) * dtype(gamma)
    s = s - s.max(axis=1, keepdims=True)
    e = np.exp(s)
    p = (e / e.sum(axis=1, keepdims=True, dtype=dtype)).reshape(n * c, h, w)
    g = gaussian_taps(gauss_len, dtype=dtype)
    k2 = (g[:, None] * g[None, :]).astype(dtype)
    r = (len(g) - 1) // 2
    pad = int(gauss_len)
    pp = np.zeros((n * c, h + 2 * pad, w + 2 * pad), dtype=dtype)
    pp[:, pad:pad + h, pad:pad + w] = p
    oh, ow = h + 2 * pad - 2 * r, w + 2 * pad - 2 * r       # VALID conv; == h, w
    blur = np.zeros((n * c, oh, ow), dtype=dtype)
    for a in range(2 * r + 1):
        for b in range(2 * r + 1):
            blur += k2[a, b] * pp[:, a:a + oh, b:b + ow]
    tot = blur.sum(axis=(1, 2), keepdims=True, dtype=dtype)
    pn = blur / (tot + dtype(1e-100) if dtype == np.float64 else tot)
    if th is not None:          # :377-388: per map, values below th * max -> 0, renormalise
        mst = pn.max(axis=(1, 2), keepdims=True)
        pn = np.where(pn < mst * dtype(th), dtype(0), pn)
        tot = pn.sum(axis=(1, 2), keepdims=True, dtype=dtype)
        pn = pn / (tot + dtype(1e-100) if dtype == np.float64 else tot)
    hh = np.arange(oh, dtype=dtype)[None, :, None]
    ww = np.arange(ow, dtype=dtype)[None, None, :]
    mu_h = (pn * hh).sum(axis=(1, 2), dtype=dtype)
    mu_w = (pn * ww).sum(axis=(1, 2), dtype=dtype)
    mu = np.stack([mu_h, mu_w], axis=1).reshape(n, c, 2)
    pmap = np.transpose(pn.reshape(n, c, oh, ow), (0, 2, 3, 1))
    return mu.astype(dtype), pmap.astype(dtype)


# ----------------------------------------------------------------------------
# A4: likelihood read-out
# ----------------------------------------------------------------------------
def likelihood_window(scmap_hwc: np.ndarray, mu_c2: np.ndarray):
    """DGP/models/eval.py:331-343 for one frame.  scmap fp32 [H,W,C], mu [C,2] (fp32
    values, held as float64 in the reference's `markers` array).
    Returns (idx [C,2] int, likelihood [C] float64)."""
    nj = scmap_hwc.shape[2]
    idx = np.zeros((nj, 2), dtype=np.int64)
    lik = np.zeros(nj, dtype=np.float64)
    for jj in range(nj):
        mu_jj = np.asarray(mu_c2[jj], dtype=np.float64)
        ends_floor = np.floor(mu_jj).astype("int")
        ends_ceil = np.ceil(mu_jj).astype("int") + 1
        m = scmap_hwc[:, :, jj]
        sig = np.exp(m) / (np.exp(m) + 1)
        win = sig[ends_floor[0]:ends_ceil[0], ends_floor[1]:ends_ceil[1]]
        loc = np.unravel_index(np.argmax(win), win.shape)
        idx[jj] = [loc[0] + ends_floor[0], loc[1] + ends_floor[1]]
        lik[jj] = sig[idx[jj, 0], idx[jj, 1]]
    return idx, lik


# ----------------------------------------------------------------------------
# A6: DLC hard arg-max
# ----------------------------------------------------------------------------
def argmax_pose_predict(scmap: np.ndarray, offmat: Optional[np.ndarray], stride: float):
    """PET/nnet/predict.py:62-77.  scmap [H,W,C] (sigmoid probabilities), offmat
    [H,W,C,2] already multiplied by locref_stdev (:45-60) or None.
    Returns pose [C,3] = (x, y, prob) float64 and maxloc [C,2] int."""
    nj = scmap.shape[2]
    pose, locs = [], []
    for j in range(nj):
        maxloc = np.unravel_index(np.argmax(scmap[:, :, j]), scmap[:, :, j].shape)
        offset = 0 if offmat is None else np.array(offmat[maxloc][j])[::-1]
        pos_f8 = np.array(maxloc).astype("float") * stride + 0.5 * stride + offset
        pose.append(np.hstack((pos_f8[::-1], [scmap[maxloc][j]])))
        locs.append(maxloc)
    return np.array(pose), np.array(locs, dtype=np.int64)


def sigmoid_f32(x: np.ndarray) -> np.ndarray:
    """tf.sigmoid on fp32 (PET/nnet/pose_net.py:86)."""
    x = x.astype(np.float32)
    return (np.float32(1) / (np.float32(1) + np.exp(-x))).astype(np.float32)


# ----------------------------------------------------------------------------
# A0: whole inference step (one sess.run + host read-out, batched)
# ----------------------------------------------------------------------------
def infer(frames_u8: np.ndarray, wts: Dict[str, np.ndarray], depth: int = 50,
          stride: float = 8.0, gamma: float = 1.0, gauss_len: int = 1, dtype=np.float32):
    """estimate_pose hot loop (DGP/models/eval.py:306-357) over a batch of frames.
    Returns dict(x, y, likelihoods [T,nj] float64; mu [T,nj,2] f32; idx [T,nj,2]; scmap).
    dtype=np.float64: the accuracy anchor (see resnet_features) -- backbone, heads and soft-argmax in double precision (mu is then
    float64); the tolerance tests measure every fp32 evaluation, the CPU oracle's own included, against it."""
    feats = resnet_features(frames_u8, wts, depth, dtype=dtype)
    scmap, _ = pose_heads(feats, wts, False)
    mu, _ = argmax_2d_from_cm(scmap, gamma, gauss_len, dtype=dtype)
    t, nj = mu.shape[:2]
    idx = np.zeros((t, nj, 2), dtype=np.int64)
    lik = np.zeros((t, nj))
    for i in range(t):
        idx[i], lik[i] = likelihood_window(scmap[i], mu[i])
    markers = mu.astype(np.float64)
    xr = markers[:, :, 1] * stride + 0.5 * stride
    yr = markers[:, :, 0] * stride + 0.5 * stride
    return {"x": xr, "y": yr, "likelihoods": lik, "mu": mu, "idx": idx, "scmap": scmap,
            "features": feats}


def motion_energy(frames_u8: np.ndarray) -> np.ndarray:
    """calculate_motion_energy's per-frame loop (DGP/dataset.py:29-43) on an in-memory uint8 clip [T, H, W, 3]: frame 0 gets 0,
    frame t the np.mean of np.abs(frame_t - frame_{t-1}) -- uint8 arithmetic, so the difference wraps modulo 256 and np.abs is
    the identity, exactly as in the reference."""
    frames_u8 = np.asarray(frames_u8)
    assert frames_u8.dtype == np.uint8
    me = np.zeros(frames_u8.shape[0])
    prev = None
    for t, frame in enumerate(frames_u8):
        if prev is not None:
            me[t] = np.mean(np.abs(frame - prev))
        prev = frame
    return me
