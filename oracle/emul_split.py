"""Where does the fp16-split engine's distance from the fp64 anchor come from?  TEST INFRASTRUCTURE / ANALYSIS ONLY (see dgp_oracle.py).

Evaluates the oracle's graph in float64 with ONE source of rounding switched on at a time:
  w_tensor  : conv weights cut into an fp16 high / low pair on ONE power-of-two scale per tensor (max -> [2^14, 2^15))   [the engine]
  w_column  : the same with one power-of-two scale per output column
  act_h2    : every activation tensor stored as H2 cells (max * 2^e in [2^10, 2^11), fp16 high / low)                     [the engine]
  act_f32   : every activation tensor rounded to fp32 (what ANY fp32 evaluation pays at least)
Usage: python -m oracle.emul_split H W seed..."""
import sys
import numpy as np
import torch
import torch.nn.functional as F
from . import dgp_oracle as O


def split_f16(x: torch.Tensor, scale) -> torch.Tensor:
    xs = x * scale
    hi = xs.to(torch.float16).to(torch.float64)
    lo = (xs - hi).to(torch.float16).to(torch.float64)
    return (hi + lo) / scale


def pow2_for(mx, top):
    """2^(top - E) for max = m 2^E"""
    e = torch.floor(torch.log2(mx.clamp_min(1e-300)))
    return torch.where(mx > 0, 2.0 ** (top - e), torch.ones_like(mx))


def run(frames, wts, mode, depth=50):
    from oracle.resnet_plan import units as resnet_units
    name = "resnet_v1_%d" % depth

    def qw(w_hwio):
        w = torch.from_numpy(np.ascontiguousarray(w_hwio)).double()
        if mode == "w_tensor":
            return split_f16(w, pow2_for(w.abs().max(), 14))
        if mode == "w_column":
            return split_f16(w, pow2_for(w.abs().amax(dim=(0, 1, 2), keepdim=True), 14))
        return w

    def qa(x):
        if mode == "act_h2":
            return split_f16(x, pow2_for(x.abs().max(), 10))
        if mode == "act_f32":
            return x.float().double()
        return x

    def conv(x, w_hwio, stride=1, rate=1, same_explicit=False):
        w = qw(w_hwio).permute(3, 2, 0, 1).contiguous()
        k = w_hwio.shape[0]
        if same_explicit and stride > 1:
            keff = k + (k - 1) * (rate - 1)
            pb = (keff - 1) // 2
            x = F.pad(x, (pb, keff - 1 - pb, pb, keff - 1 - pb))
        else:
            _, pt, pbt = O.tf_same_pads(x.shape[2], k, stride, rate)
            _, pl, pr = O.tf_same_pads(x.shape[3], k, stride, rate)
            x = F.pad(x, (pl, pr, pt, pbt))
        return F.conv2d(x, w, stride=stride, dilation=rate)

    x = frames.astype(np.float32) - np.asarray(O.MEAN_PIXEL, dtype=np.float32)
    x = torch.from_numpy(x.astype(np.float64)).permute(0, 3, 1, 2)
    with torch.no_grad():
        net = F.relu(O.batch_norm(conv(x, wts[name + "/conv1/weights"], 2, same_explicit=True), wts, name + "/conv1"))
        net = qa(O.max_pool_same(net, 3, 2))
        for u in resnet_units(depth):
            if u.has_shortcut_conv:
                sc = O.batch_norm(conv(net, wts[u.scope + "/shortcut/weights"], u.stride), wts, u.scope + "/shortcut")
            else:
                sc = O.subsample(net, u.stride)
            r = qa(F.relu(O.batch_norm(conv(net, wts[u.scope + "/conv1/weights"]), wts, u.scope + "/conv1")))
            r = qa(F.relu(O.batch_norm(conv(r, wts[u.scope + "/conv2/weights"], u.stride, u.rate, same_explicit=True), wts, u.scope + "/conv2")))
            r = O.batch_norm(conv(r, wts[u.scope + "/conv3/weights"]), wts, u.scope + "/conv3")
            net = qa(F.relu(sc + r))
    feats = net.permute(0, 2, 3, 1).contiguous().numpy()
    hw = dict(wts)
    if mode in ("w_tensor", "w_column"):
        w = torch.from_numpy(wts["pose/part_pred/block4/weights"]).double()
        mx = w.abs().max() if mode == "w_tensor" else w.abs().amax(dim=(0, 1, 3), keepdim=True)
        hw["pose/part_pred/block4/weights"] = split_f16(w, pow2_for(mx, 14)).numpy()
    scmap, _ = O.pose_heads(feats, hw, False)
    mu, _ = O.argmax_2d_from_cm(scmap, 1.0, 1, dtype=np.float64)
    return mu, scmap


if __name__ == "__main__":
    from deepgraphpose_amd.synthetic import make_frames, make_stress_weights
    H, W = int(sys.argv[1]), int(sys.argv[2])
    for seed in [int(v) for v in sys.argv[3:]]:
        wts = make_stress_weights(50, 4, False, seed=seed)
        frames = make_frames(2, H, W, 4, seed=seed + 1)
        s_ref, _ = O.pose_heads(O.resnet_features(frames, wts, 50), wts, False)
        wts["pose/part_pred/block4/weights"] = (wts["pose/part_pred/block4/weights"] * np.float32(3.0 / s_ref.std())).astype(np.float32)
        mu0, sc0 = run(frames, wts, "exact")
        r64 = O.infer(frames, wts, 50, 8.0, 1.0, 1, dtype=np.float64)
        print("seed %d: emulator vs oracle fp64 %.2g px" % (seed, np.abs(mu0 - r64["mu"]).max() * 8))
        for mode in ("w_tensor", "w_column", "act_h2", "act_f32"):
            mu, sc = run(frames, wts, mode)
            print("  %-9s px vs fp64 %.3g   scmap rel %.3g" % (mode, np.abs(mu - mu0).max() * 8, np.abs(sc - sc0).max() / np.abs(sc0).max()), flush=True)
