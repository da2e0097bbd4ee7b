"""CPU oracle for the DGP TRAINING step -- TEST INFRASTRUCTURE ONLY (see oracle/dgp_oracle.py).

Differentiable torch-CPU restatement of dgp_loss (DGP/models/fitdgp.py:848-1144) on top of the same network
semantics as dgp_oracle.py, with torch autograd standing in for TF's tf.gradients, plus the optimiser step
(clip_by_global_norm(10) + MomentumOptimizer(0.9), fitdgp.py:708-713).  fp32 by default, fp64 for
gradient checks.  Parity status: unpinned against TF itself (TF cannot run here); the expressions follow the
cited lines one by one.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import dgp_oracle as O


# ------------------------------------------------------------------ differentiable network (A1 + A2)
def _w(wts, name, dtype):
    return wts[name] if isinstance(wts[name], torch.Tensor) else torch.from_numpy(wts[name]).to(dtype)


def _conv(x, w_hwio, stride=1, rate=1, padding="SAME"):
    kh, kw = w_hwio.shape[:2]
    w = w_hwio.permute(3, 2, 0, 1)
    if padding == "SAME":
        _, pt, pb = O.tf_same_pads(x.shape[2], kh, stride, rate)
        _, pl, pr = O.tf_same_pads(x.shape[3], kw, stride, rate)
        x = F.pad(x, (pl, pr, pt, pb))
    return F.conv2d(x, w, stride=stride, dilation=rate)


def _conv_same(x, w_hwio, stride, rate=1):
    k = w_hwio.shape[0]
    if stride == 1:
        return _conv(x, w_hwio, 1, rate, "SAME")
    keff = k + (k - 1) * (rate - 1)
    pb = (keff - 1) // 2
    pe = keff - 1 - pb
    return _conv(F.pad(x, (pb, pe, pb, pe)), w_hwio, stride, rate, "VALID")


def _bn(x, P, scope):
    """inference-mode BN (moving stats frozen) with TRAINABLE gamma/beta (fitdgp.py:708: all trainable vars)."""
    g, b = P[scope + "/BatchNorm/gamma"], P[scope + "/BatchNorm/beta"]
    m, v = P[scope + "/BatchNorm/moving_mean"], P[scope + "/BatchNorm/moving_variance"]
    inv = g * torch.rsqrt(v + O.BN_EPS)
    return (x - m[None, :, None, None]) * inv[None, :, None, None] + b[None, :, None, None]


def _deconv(x, w, b, stride=2):
    """conv2d_transpose SAME (see dgp_oracle.conv2d_transpose_same); w [kh,kw,Cout,Cin]."""
    n, c, h, wd = x.shape
    kh, kw = w.shape[:2]
    oh, ow = h * stride, wd * stride
    _, pt, _ = O.tf_same_pads(oh, kh, stride)
    _, pl, _ = O.tf_same_pads(ow, kw, stride)
    full = F.conv_transpose2d(x, w.permute(3, 2, 0, 1), stride=stride)
    return full[:, :, pt:pt + oh, pl:pl + ow] + b[None, :, None, None]


def network(frames_u8: np.ndarray, P: Dict[str, torch.Tensor], depth: int = 50, dtype=torch.float32):
    """-> (pred [nt,H,W,nj], locref [nt,H,W,2nj]) as NHWC torch tensors attached to the graph of P."""
    from oracle.resnet_plan import units as resnet_units   # the oracle's OWN restatement of slim's plan
    name = "resnet_v1_%d" % depth
    x = torch.from_numpy(frames_u8.astype(np.float32) - np.asarray(O.MEAN_PIXEL, np.float32)).to(dtype)
    x = x.permute(0, 3, 1, 2)
    net = F.relu(_bn(_conv_same(x, P[name + "/conv1/weights"], 2), P, name + "/conv1"))
    _, pt, pb = O.tf_same_pads(net.shape[2], 3, 2)
    _, pl, pr = O.tf_same_pads(net.shape[3], 3, 2)
    net = F.max_pool2d(F.pad(net, (pl, pr, pt, pb), value=float("-inf")), 3, 2)
    for u in resnet_units(depth):
        if u.has_shortcut_conv:
            sc = _bn(_conv(net, P[u.scope + "/shortcut/weights"], u.stride), P, u.scope + "/shortcut")
        else:
            sc = net if u.stride == 1 else net[:, :, ::u.stride, ::u.stride]
        r = F.relu(_bn(_conv(net, P[u.scope + "/conv1/weights"], 1), P, u.scope + "/conv1"))
        r = F.relu(_bn(_conv_same(r, P[u.scope + "/conv2/weights"], u.stride, u.rate), P, u.scope + "/conv2"))
        r = _bn(_conv(r, P[u.scope + "/conv3/weights"], 1), P, u.scope + "/conv3")
        net = F.relu(sc + r)
    pred = _deconv(net, P["pose/part_pred/block4/weights"], P["pose/part_pred/block4/biases"])
    loc = _deconv(net, P["pose/locref_pred/block4/weights"], P["pose/locref_pred/block4/biases"])
    return pred.permute(0, 2, 3, 1), loc.permute(0, 2, 3, 1)


TRAINABLE_SUFFIXES = ("/weights", "/biases", "/BatchNorm/gamma", "/BatchNorm/beta")


def make_params(wts: Dict[str, np.ndarray], dtype=torch.float32):
    """numpy weights -> dict of torch tensors; trainables require grad (moving stats do not)."""
    P = {}
    for k, v in wts.items():
        t = torch.from_numpy(np.asarray(v)).to(dtype).clone()
        if k.endswith(TRAINABLE_SUFFIXES):
            t.requires_grad_(True)
        P[k] = t
    return P


# ------------------------------------------------------------------ differentiable soft-argmax (A3)
def soft_argmax(pred: torch.Tensor, gamma: float, gauss_len: int):
    """pred [N,H,W,C] -> mu [N,C,2]; fitdgp_util.py:342-402 with autograd."""
    n, h, w, c = pred.shape
    s = pred.permute(0, 3, 1, 2).reshape(n * c, h * w) * gamma
    p = torch.softmax(s, dim=1).reshape(n * c, 1, h, w)
    r = int(gauss_len)
    xs = torch.arange(-r, r + 1, dtype=pred.dtype)
    g = torch.exp(-0.5 * (xs / gauss_len) ** 2)
    g = g / g.sum()
    k2 = (g[:, None] * g[None, :])[None, None]
    b = F.conv2d(F.pad(p, (r, r, r, r)), k2)[:, 0]
    b = b / (b.sum(dim=(1, 2), keepdim=True))
    hh = torch.arange(h, dtype=pred.dtype)[None, :, None]
    ww = torch.arange(w, dtype=pred.dtype)[None, None, :]
    return torch.stack([(b * hh).sum(dim=(1, 2)), (b * ww).sum(dim=(1, 2))], 1).reshape(n, c, 2)


# ------------------------------------------------------------------ loss pre-computation (B1)
def limb_statistics(joint_loc_full: np.ndarray, S0: np.ndarray, stride: float, ws: float, ws_max: float):
    """fitdgp.py:875-892 -> (ws [nl], ws_max [nl]) from all labels [n, nj, 2] (NaN = unlabeled)."""
    nj = joint_loc_full.shape[1]
    j1 = np.copy(joint_loc_full).swapaxes(1, 2).reshape(-1, nj)
    j1[np.isnan(j1)] = 1e10
    limb = np.matmul(j1, S0.T)
    limb[np.abs(limb) > 1e5] = 0
    limb = np.reshape(limb, [joint_loc_full.shape[0], 2, -1])
    limb = np.sqrt(np.sum(np.square(limb), 1))
    limb = limb.T * stride + stride / 2
    ws_max_v = np.max(np.nan_to_num(limb), 1) * ws_max
    with np.errstate(invalid="ignore", divide="ignore"):
        mean_nz = np.true_divide(limb.sum(1), (limb != 0).sum(1))
    ws_v = 1 / (np.nan_to_num(mean_nz) + 1e-20) * ws
    return ws_v, ws_max_v


# ------------------------------------------------------------------ dgp_loss (B3-B8)
def _sigmoid_ce(z, x):
    """tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x*z + log(1 + exp(-|x|))."""
    return torch.clamp(x, min=0) - x * z + torch.log1p(torch.exp(-torch.abs(x)))


def dgp_loss(pred, locref_pred, batch: dict, cfg: dict):
    """pred [nt,H,W,nj], locref_pred [nt,H,W,2nj] (torch, graph attached) -> dict of losses.

    batch: targets [nv,nj,2] (NaN unlabeled), locref_map/mask [nt,H,W,2nj], visible_marker, hidden_marker,
           visible_marker_in_targets (int arrays), nt, wt_batch_mask [nt-1], vector_field [nt-1,Hin,Win] or None
    cfg:   nj, S0, ws [nl], ws_max [nl], stride, gamma, gauss_len, lengthscale, gm2, gm3, wn_visible, wn_hidden,
           wt, wt_max, locref_loss_weight, locref_huber_loss, n_frames_total, n_visible_frames_total
    """
    dt = pred.dtype
    nt, H, W, nj = pred.shape
    vm = torch.as_tensor(batch["visible_marker"], dtype=torch.long)
    hm = torch.as_tensor(batch["hidden_marker"], dtype=torch.long)
    vt = torch.as_tensor(batch["visible_marker_in_targets"], dtype=torch.long)
    n_vis_total = float(cfg["n_visible_frames_total"])
    n_hid_total = float(cfg["n_frames_total"]) - n_vis_total
    targets = torch.as_tensor(np.nan_to_num(np.asarray(batch["targets"], dtype=np.float64), nan=0.0)).to(dt)

    mu = soft_argmax(pred, cfg["gamma"], cfg["gauss_len"])                     # :946
    mu_marker = mu.reshape(-1, 2)
    t_all = torch.zeros((nt * nj, 2), dtype=dt)
    t_all = t_all.index_add(0, hm, mu_marker[hm])                             # combine_all_marker :958
    if len(vm):
        t_all = t_all.index_add(0, vm, targets.reshape(-1, 2)[vt])

    hh = torch.arange(H, dtype=dt)[None, :, None]
    ww = torch.arange(W, dtype=dt)[None, None, :]
    d2 = (hh - t_all[:, 0, None, None]) ** 2 + (ww - t_all[:, 1, None, None]) ** 2
    G = torch.exp(-d2 / (2 * cfg["lengthscale"] ** 2))                         # :970
    G = G / (G.amax(dim=(1, 2), keepdim=True) + 1e-5)

    n_h = float(len(hm))
    n_v = float(len(vm))
    n_v_eff = n_v if n_v > 0 else n_h                                           # :983-984
    predm = pred.permute(0, 3, 1, 2).reshape(-1, H, W)
    G_v, G_h, pred_v, pred_h = G[vm], G[hm], predm[vm], predm[hm]

    loss = {}
    loss["visible_loss_pred"] = _sigmoid_ce(G_v, pred_v).mean() if len(vm) else torch.zeros((), dtype=dt)
    scale_h = n_vis_total / n_hid_total * n_h / n_v_eff * cfg["wn_hidden"] / cfg["wn_visible"] if n_h > 0 else 0.0
    if len(hm) == 0:
        loss["hidden_loss_pred"] = torch.zeros((), dtype=dt)
    else:
        if cfg["gm2"] in (1, 2):
            q = torch.sigmoid(pred_h)
            c = q.amax(dim=(1, 2), keepdim=True)
            if cfg["gm2"] == 1:
                G_h = G_h * c
            qs = q * c
            pred_h_s = -torch.log(1 - qs + 1e-20) + torch.log(qs + 1e-20)
        if cfg["gm3"] == 3:
            wgt = (1 - c).expand_as(pred_h_s)
            present = (wgt != 0).sum().to(dt)
            ce = (_sigmoid_ce(G_h, pred_h_s) * wgt).sum()
            loss["hidden_loss_pred"] = (ce / present if present > 0 else ce * 0) * scale_h
        else:
            loss["hidden_loss_pred"] = _sigmoid_ce(G_h, pred_h).mean() * scale_h
    total = loss["visible_loss_pred"] + loss["hidden_loss_pred"]

    # locref (:1041-1055): channels regrouped [nt*nj, 2, H, W], visible markers only
    def regroup(a):
        return a.permute(0, 3, 1, 2).reshape(-1, 2, H, W)
    lmap = torch.as_tensor(np.asarray(batch["locref_map"], dtype=np.float64)).to(dt)
    lmask = torch.as_tensor(np.asarray(batch["locref_mask"], dtype=np.float64)).to(dt)
    lp_v, lm_v, lk_v = regroup(locref_pred)[vm], regroup(lmap)[vm], regroup(lmask)[vm]
    diff = lp_v - lm_v
    if cfg.get("locref_huber_loss", True):
        el = torch.where(diff.abs() < 1.0, 0.5 * diff ** 2, diff.abs() - 0.5)
    else:
        el = diff ** 2
    nz = (lk_v != 0).sum().to(dt)
    lr = (el * lk_v).sum()
    loss["visible_loss_locref"] = cfg["locref_loss_weight"] * (lr / nz if nz > 0 else lr * 0)
    total = total + loss["visible_loss_locref"]

    t3 = t_all.reshape(nt, nj, 2)
    S0 = np.asarray(cfg["S0"])
    if S0.shape[0] > 0:                                                         # :1062-1076
        S = torch.as_tensor(S0).to(dt)
        P_sp = t3.permute(1, 2, 0).reshape(nj, -1) * cfg["stride"] + 0.5 * cfg["stride"]
        dist = torch.sqrt(((S @ P_sp).reshape(S0.shape[0], 2, -1) ** 2).sum(1))
        wmax = torch.as_tensor(np.asarray(cfg["ws_max"])).to(dt)[:, None]
        dist_th = F.relu(dist - wmax) + wmax
        ws = torch.as_tensor(np.asarray(cfg["ws"])).to(dt)[:, None]
        loss["ws_loss"] = (dist_th * ws).sum() / H / W * n_vis_total / n_v_eff / (n_vis_total + n_hid_total) / cfg["wn_visible"]
        total = total + loss["ws_loss"]
    if cfg.get("wt", 0) > 0 and nt > 1 and batch.get("vector_field") is not None:        # :1079-1124
        P_t = t3 * cfg["stride"] + 0.5 * cfg["stride"]
        dif = torch.sqrt(((P_t[:-1] - P_t[1:]) ** 2).sum(2))                             # [nt-1, nj]
        vf = np.asarray(batch["vector_field"], dtype=np.float64)
        mask = np.asarray(batch.get("wt_batch_mask", np.ones(nt - 1)), dtype=np.float64)
        wt_batch = np.ones(nt - 1) * cfg["wt"] * mask
        # the crop boxes are functions of the (differentiable) targets and tf.image.crop_and_resize has a gradient with respect to
        # its boxes (python/ops/image_grad.py _CropAndResizeGrad -> CropAndResizeGradBoxes), so the weight is differentiated too;
        # cfg["wt_weight_grad"] = False gives the weight as a constant (stop-gradient), for comparison
        if cfg.get("wt_weight_grad", True):
            w = temporal_flow_weights_torch(P_t.to(torch.float64), vf, wt_batch, H, W).to(dt)
        else:
            w = torch.as_tensor(temporal_flow_weights(P_t.detach().numpy().astype(np.float64), vf, wt_batch, H, W)).to(dt)
        v = (F.relu(dif - cfg["wt_max"]) + cfg["wt_max"]) * w
        loss["wt_loss"] = torch.sqrt((v ** 2).sum()) * n_vis_total / n_v_eff / (n_vis_total + n_hid_total) / cfg["wn_visible"]
        total = total + loss["wt_loss"]
    loss["total_loss"] = total
    loss["total_loss_visible"] = loss["visible_loss_pred"] + loss["visible_loss_locref"]
    loss["_mu"] = mu
    return loss


def crop_and_resize_mean(img: np.ndarray, box, crop_hw):
    """mean of tf.image.crop_and_resize(img[None,...,None], [box], [0], crop_hw) (bilinear, extrapolation 0).
    box = (y1, x1, y2, x2) normalised as TF defines it (y * (H-1) is the source row)."""
    Himg, Wimg = img.shape
    ch, cw = crop_hw
    y1, x1, y2, x2 = box
    hs = (y2 - y1) * (Himg - 1) / (ch - 1) if ch > 1 else 0.0
    ws_ = (x2 - x1) * (Wimg - 1) / (cw - 1) if cw > 1 else 0.0
    iy = y1 * (Himg - 1) + np.arange(ch) * hs if ch > 1 else np.array([0.5 * (y1 + y2) * (Himg - 1)])
    ix = x1 * (Wimg - 1) + np.arange(cw) * ws_ if cw > 1 else np.array([0.5 * (x1 + x2) * (Wimg - 1)])
    vy = (iy >= 0) & (iy <= Himg - 1)
    vx = (ix >= 0) & (ix <= Wimg - 1)
    iyc, ixc = np.clip(iy, 0, Himg - 1), np.clip(ix, 0, Wimg - 1)
    ty, by = np.floor(iyc).astype(int), np.ceil(iyc).astype(int)
    lx, rx = np.floor(ixc).astype(int), np.ceil(ixc).astype(int)
    fy, fx = (iyc - ty)[:, None], (ixc - lx)[None, :]
    top = img[ty][:, lx] + (img[ty][:, rx] - img[ty][:, lx]) * fx
    bot = img[by][:, lx] + (img[by][:, rx] - img[by][:, lx]) * fx
    out = (top + (bot - top) * fy) * (vy[:, None] & vx[None, :])
    return out.sum() / (ch * cw)


def temporal_flow_weights(P_t: np.ndarray, vector_field: np.ndarray, wt_batch: np.ndarray, H: int, W: int, window=10.0):
    """fitdgp.py:1085-1118: per (frame pair, joint) weight min(min(1/(mean flow + 1e-10), 1)^3, 1) * wt / H / W, the
    mean flow taken over the +-10 px box of the two marker positions resampled to the full frame."""
    ntm1, nj = P_t.shape[0] - 1, P_t.shape[1]
    Hin, Win = vector_field.shape[1:]
    w = np.zeros((ntm1, nj))
    for t in range(ntm1):
        for j in range(nj):
            r0, c0 = P_t[t, j]
            r1, c1 = P_t[t + 1, j]
            box = (max(0.0, min(r0, r1) - window) / Hin, max(0.0, min(c0, c1) - window) / Win,
                   min(float(Hin), max(r0, r1) + window) / Hin, min(float(Win), max(c0, c1) + window) / Win)
            m = crop_and_resize_mean(vector_field[t], box, (Hin, Win))
            inv = min(1.0 / (m + 1e-10), 1.0)
            inv = min(np.exp(np.log(inv) * 3), 1.0)
            w[t, j] = inv * wt_batch[t] / H / W
    return w


def temporal_flow_weights_torch(P_t, vector_field: np.ndarray, wt_batch: np.ndarray, H: int, W: int, window=10.0):
    """temporal_flow_weights with the autograd path TF has (fitdgp.py:1085-1118): P_t [nt, nj, 2] torch float64 (graph attached).
    crop_and_resize_op.cc: a sample at in_y = y1 (Hin - 1) + i (y2 - y1) (crop height = image height), bilinear between floor and
    ceil rows; CropAndResizeGradBoxes differentiates the LERP FRACTIONS with respect to the box (the integer corner indices are
    constants), which is what autograd does on `iy - floor(iy)` below.  reduce_min / reduce_max of the two positions share a tie's
    gradient evenly (math_grad.py _MinOrMaxGrad); maximum(0, v) / minimum(n, v) pass the gradient to v where v is strictly inside."""
    ntm1, nj = P_t.shape[0] - 1, P_t.shape[1]
    Hin, Win = vector_field.shape[1:]
    img_all = torch.as_tensor(np.asarray(vector_field, dtype=np.float64))
    rows = []
    for t in range(ntm1):
        row = []
        for j in range(nj):
            pair_r = torch.stack([P_t[t, j, 0], P_t[t + 1, j, 0]])
            pair_c = torch.stack([P_t[t, j, 1], P_t[t + 1, j, 1]])
            y1 = torch.clamp(torch.amin(pair_r) - window, min=0.0) / Hin
            y2 = torch.clamp(torch.amax(pair_r) + window, max=float(Hin)) / Hin
            x1 = torch.clamp(torch.amin(pair_c) - window, min=0.0) / Win
            x2 = torch.clamp(torch.amax(pair_c) + window, max=float(Win)) / Win
            img = img_all[t]
            iy = y1 * (Hin - 1) + torch.arange(Hin, dtype=torch.float64) * (y2 - y1) if Hin > 1 else (0.5 * (y1 + y2) * (Hin - 1)).reshape(1)
            ix = x1 * (Win - 1) + torch.arange(Win, dtype=torch.float64) * (x2 - x1) if Win > 1 else (0.5 * (x1 + x2) * (Win - 1)).reshape(1)
            vy = ((iy >= 0) & (iy <= Hin - 1)).to(torch.float64)
            vx = ((ix >= 0) & (ix <= Win - 1)).to(torch.float64)
            iyc, ixc = torch.clamp(iy, 0, Hin - 1), torch.clamp(ix, 0, Win - 1)
            ty, by = torch.floor(iyc).long(), torch.ceil(iyc).long()
            lx, rx = torch.floor(ixc).long(), torch.ceil(ixc).long()
            fy, fx = (iy - ty.to(torch.float64))[:, None], (ix - lx.to(torch.float64))[None, :]       # differentiable fractions
            top = img[ty][:, lx] + (img[ty][:, rx] - img[ty][:, lx]) * fx
            bot = img[by][:, lx] + (img[by][:, rx] - img[by][:, lx]) * fx
            m = ((top + (bot - top) * fy) * (vy[:, None] * vx[None, :])).sum() / (Hin * Win)
            inv = torch.clamp(1.0 / (m + 1e-10), max=1.0)
            inv = torch.clamp(torch.exp(torch.log(inv) * 3), max=1.0)
            row.append(inv * float(wt_batch[t]) / H / W)
        rows.append(torch.stack(row))
    return torch.stack(rows)


# ------------------------------------------------------------------ DLC step-0 loss (N2)
def dlc_loss(pred, locref_pred, part_targets, locref_targets, locref_mask, part_weights=None,
             locref_loss_weight: float = 0.05, huber: bool = True):
    """pose_net.train (DeepLabCut nnet/pose_net.py:159-190): tf.losses.sigmoid_cross_entropy (weights 1.0 or the
    part_score_weights map; reduction SUM_BY_NONZERO_WEIGHTS) + locref_loss_weight * huber_loss(..., locref_mask)
    (nnet/losses.py:16-45, same reduction: sum / number of non-zero weights, 0 when there are none)."""
    ce = _sigmoid_ce(part_targets, pred)
    if part_weights is None:
        part = ce.mean()
    else:
        nz = (part_weights != 0).sum()
        part = (ce * part_weights).sum() / nz if nz > 0 else ce.sum() * 0
    out = {"part_loss": part, "total_loss": part}
    if locref_pred is not None:
        d = locref_pred - locref_targets
        ad = d.abs()
        el = torch.where(ad < 1.0, 0.5 * d * d, ad - 0.5) if huber else d * d
        nz = (locref_mask != 0).sum()
        loc = locref_loss_weight * ((el * locref_mask).sum() / nz if nz > 0 else el.sum() * 0)
        out["locref_loss"] = loc
        out["total_loss"] = part + loc
    return out


# ------------------------------------------------------------------ optimiser (B9)
def momentum_step(P: Dict[str, torch.Tensor], V: Dict[str, torch.Tensor], lr: float, momentum: float = 0.9,
                  clip: float = 10.0):
    """clip_by_global_norm(10) then MomentumOptimizer: accum = m*accum + g ; var -= lr*accum (fitdgp.py:708-713)."""
    names = [k for k, t in P.items() if t.requires_grad and t.grad is not None]
    gn = torch.sqrt(sum((P[k].grad.double() ** 2).sum() for k in names)).item()
    scale = clip / max(gn, clip) if clip > 0 else 1.0          # clip <= 0: plain MomentumOptimizer (fit_dlc)
    with torch.no_grad():
        for k in names:
            g = P[k].grad * scale
            V[k] = momentum * V.get(k, torch.zeros_like(g)) + g
            P[k] -= lr * V[k]
            P[k].grad = None
    return gn
