"""Seeded synthetic weights and frames (no checkpoint / dataset is reachable).

Weights use the TF variable names and layouts of the reference graph so the
same dict can be fed to the HIP engine, to the CPU oracle, and -- once a TF
checkpoint reader exists -- be replaced by real DLC/DGP snapshots:

  resnet_v1_50/conv1/weights                              [7,7,3,64]   HWIO
  resnet_v1_50/conv1/BatchNorm/{gamma,beta,moving_mean,moving_variance}
  resnet_v1_50/block{b}/unit_{u}/bottleneck_v1/{shortcut,conv1,conv2,conv3}/...
  pose/part_pred/block4/{weights [3,3,nj,2048], biases [nj]}     (PET/nnet/pose_net.py:18-26)
  pose/locref_pred/block4/{weights [3,3,2nj,2048], biases [2nj]}
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from .arch import BN_EPS as BN_EPS_, resnet_units


def _conv(rng, kh, kw, cin, cout, gain=2.0):
    std = np.sqrt(gain / (kh * kw * cin))
    return (rng.standard_normal((kh, kw, cin, cout)) * std).astype(np.float32)


def _bn(rng, c, prefix, out, gamma_mean=1.0):
    out[prefix + "/BatchNorm/gamma"] = (gamma_mean * (1.0 + 0.1 * rng.standard_normal(c))).astype(np.float32)
    out[prefix + "/BatchNorm/beta"] = (0.1 * rng.standard_normal(c)).astype(np.float32)
    out[prefix + "/BatchNorm/moving_mean"] = (0.1 * rng.standard_normal(c)).astype(np.float32)
    out[prefix + "/BatchNorm/moving_variance"] = (1.0 + 0.2 * rng.random(c)).astype(np.float32)


def make_weights(depth: int = 50, nj: int = 4, with_locref: bool = False,
                 seed: int = 0, head_std: float = 0.01) -> Dict[str, np.ndarray]:
    """He-normal convs, near-identity BN, N(0, head_std) deconv heads."""
    rng = np.random.default_rng(seed)
    name = "resnet_v1_%d" % depth
    w: Dict[str, np.ndarray] = {}
    w[name + "/conv1/weights"] = _conv(rng, 7, 7, 3, 64)
    # frames are 0..255 minus mean: scale the stem BN so activations are O(1)
    _bn(rng, 64, name + "/conv1", w)
    w[name + "/conv1/BatchNorm/moving_variance"] *= np.float32(70.0 ** 2)
    for u in resnet_units(depth):
        if u.has_shortcut_conv:
            w[u.scope + "/shortcut/weights"] = _conv(rng, 1, 1, u.depth_in, u.depth, gain=1.0)
            _bn(rng, u.depth, u.scope + "/shortcut", w)
        w[u.scope + "/conv1/weights"] = _conv(rng, 1, 1, u.depth_in, u.depth_bottleneck)
        _bn(rng, u.depth_bottleneck, u.scope + "/conv1", w)
        w[u.scope + "/conv2/weights"] = _conv(rng, 3, 3, u.depth_bottleneck, u.depth_bottleneck)
        _bn(rng, u.depth_bottleneck, u.scope + "/conv2", w)
        w[u.scope + "/conv3/weights"] = _conv(rng, 1, 1, u.depth_bottleneck, u.depth, gain=1.0)
        # damp the residual branch like a trained net so depth does not blow up the scale
        _bn(rng, u.depth, u.scope + "/conv3", w, gamma_mean=0.5)
    w["pose/part_pred/block4/weights"] = (head_std * rng.standard_normal((3, 3, nj, 2048))).astype(np.float32)
    w["pose/part_pred/block4/biases"] = (0.01 * rng.standard_normal(nj)).astype(np.float32)
    if with_locref:
        w["pose/locref_pred/block4/weights"] = (head_std * rng.standard_normal((3, 3, 2 * nj, 2048))).astype(np.float32)
        w["pose/locref_pred/block4/biases"] = (0.01 * rng.standard_normal(2 * nj)).astype(np.float32)
    return w


def make_stress_weights(depth: int = 50, nj: int = 4, with_locref: bool = False, seed: int = 0, head_std: float = 0.05,
                        dead_frac: float = 0.05, n_outliers: int = 3) -> Dict[str, np.ndarray]:
    """Weights with the statistics of a TRAINED network instead of make_weights' near-identity BatchNorm -- what the H2 activation
    format (one power-of-two scale per tensor) and the per-tensor weight scales have to survive:
      * the folded BN scale gamma * rsqrt(var + eps) of a layer's channels is log-uniform over [2^-8, 2^4], var log-uniform over
        [1e-3, 1e2];
      * the output magnitude of a layer's channels is log-uniform over [2^-6, 2^2] (weight columns compensate the BN scale, so a
        panel's columns differ by up to 2^20);
      * `dead_frac` of the channels are dead (gamma 0, beta < 0: zero after the ReLU) and `n_outliers` channels per layer are
        100 x larger than their layer's typical channel.
    The network stays finite: the pre-BN variance of a channel matches the variance its BN divides by."""
    rng = np.random.default_rng(seed)
    name = "resnet_v1_%d" % depth
    w: Dict[str, np.ndarray] = {}

    def layer(prefix, kh, kw, cin, cout, in_ms, gain=2.0, relu_after=True, damp=1.0):
        """conv + BN whose output channel c has standard deviation ~ a[c]; returns E[x^2] of the (post-ReLU) output per channel"""
        a = damp * 2.0 ** rng.uniform(-6, 2, cout)
        if n_outliers:
            a[rng.choice(cout, min(n_outliers, cout), replace=False)] *= 100.0
        s = 2.0 ** rng.uniform(-8, 4, cout)                         # folded BN scale
        var = 10.0 ** rng.uniform(-3, 2, cout)
        base = rng.standard_normal((kh, kw, cin, cout)) * np.sqrt(gain / (kh * kw * cin))
        pre_std = (a / s)                                           # what the conv must deliver so that BN scales it to a
        wconv = base * (pre_std / np.sqrt(max(float(np.mean(in_ms)), 1e-30) * gain / 2.0))[None, None, None, :]
        gamma = s * np.sqrt(var + BN_EPS_)
        beta = 0.1 * a * rng.standard_normal(cout)
        mean = 0.1 * pre_std * rng.standard_normal(cout)
        dead = rng.random(cout) < dead_frac
        gamma[dead] = 0.0
        beta[dead] = -0.1 * a[dead]
        w[prefix + "/weights"] = wconv.astype(np.float32)
        w[prefix + "/BatchNorm/gamma"] = gamma.astype(np.float32)
        w[prefix + "/BatchNorm/beta"] = beta.astype(np.float32)
        w[prefix + "/BatchNorm/moving_mean"] = mean.astype(np.float32)
        w[prefix + "/BatchNorm/moving_variance"] = var.astype(np.float32)
        ms = np.where(dead, 0.0, a * a * (0.5 if relu_after else 1.0) + beta * beta)
        return ms

    x_ms = layer(name + "/conv1", 7, 7, 3, 64, np.full(3, 70.0 ** 2))
    for u in resnet_units(depth):
        sc_ms = x_ms
        if u.has_shortcut_conv:
            sc_ms = layer(u.scope + "/shortcut", 1, 1, u.depth_in, u.depth, x_ms, gain=1.0, relu_after=False)
        r1 = layer(u.scope + "/conv1", 1, 1, u.depth_in, u.depth_bottleneck, x_ms)
        r2 = layer(u.scope + "/conv2", 3, 3, u.depth_bottleneck, u.depth_bottleneck, r1)
        r3 = layer(u.scope + "/conv3", 1, 1, u.depth_bottleneck, u.depth, r2, gain=1.0, relu_after=False, damp=0.5)
        x_ms = 0.5 * (sc_ms + r3)                                   # relu(shortcut + residual), roughly
    # heads: input magnitudes vary by channel (and the 100 x outlier channels dominate the sums); scale so that logits are O(10)
    hs = 0.1 * head_std / np.sqrt(max(float(np.mean(x_ms)), 1e-30))
    w["pose/part_pred/block4/weights"] = (hs * rng.standard_normal((3, 3, nj, 2048))).astype(np.float32)
    w["pose/part_pred/block4/biases"] = (0.01 * rng.standard_normal(nj)).astype(np.float32)
    if with_locref:
        w["pose/locref_pred/block4/weights"] = (hs * rng.standard_normal((3, 3, 2 * nj, 2048))).astype(np.float32)
        w["pose/locref_pred/block4/biases"] = (0.01 * rng.standard_normal(2 * nj)).astype(np.float32)
    return w


def make_frames(n: int, h: int = 480, w: int = 640, nj: int = 4, seed: int = 0) -> np.ndarray:
    """uint8 [n,h,w,3] frames: a few moving Gaussian blobs on a smooth background
    plus uint8 noise, so scoremaps get distinct, non-degenerate peaks."""
    rng = np.random.default_rng(seed + 1000)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    bg = 96.0 + 32.0 * np.sin(xx / 97.0) * np.cos(yy / 61.0)
    cy = rng.uniform(0.2 * h, 0.8 * h, size=nj)
    cx = rng.uniform(0.2 * w, 0.8 * w, size=nj)
    col = rng.uniform(60, 150, size=(nj, 3)).astype(np.float32)
    frames = np.empty((n, h, w, 3), dtype=np.uint8)
    for t in range(n):
        img = np.repeat(bg[:, :, None], 3, axis=2).copy()
        for j in range(nj):
            py = cy[j] + 0.08 * h * np.sin(0.05 * t + j)
            px = cx[j] + 0.08 * w * np.cos(0.04 * t + 2 * j)
            g = np.exp(-((yy - py) ** 2 + (xx - px) ** 2) / (2.0 * (12.0 + 3 * j) ** 2))
            img += g[:, :, None] * col[j][None, None, :]
        img += rng.integers(-6, 7, size=img.shape).astype(np.float32)
        frames[t] = np.clip(img, 0, 255).astype(np.uint8)
    return frames
