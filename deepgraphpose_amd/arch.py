"""ResNet-v1 (TF-slim flavour) layer plan used by DeepLabCut / DGP.

This is the architecture description of the hot path: a flat list of
convolution layers with the exact TF-1.15 slim geometry, shared by the HIP
engine (which mirrors it in C++, csrc/dgp_net.cpp) and by the CPU oracle.

Reference semantics restated (slim is third-party and not vendored):
  * PET/nnet/pose_net.py:36-54  -> resnet_v1_{50,101}(global_pool=False,
    output_stride=16, is_training=False)
  * slim resnet_v1: root conv2d_same(64, 7, stride 2) + max_pool2d(3, 2, SAME);
    blocks (64,3,s2) (128,4,s2) (256,6|23,s2) (512,3,s1), stride on the LAST
    unit of each block; stack_blocks_dense switches to atrous rate 2 once the
    output stride 16 is reached, so all block4 3x3 convs have rate 2.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Tuple

BLOCKS = {
    50: [(64, 3, 2), (128, 4, 2), (256, 6, 2), (512, 3, 1)],
    101: [(64, 3, 2), (128, 4, 2), (256, 23, 2), (512, 3, 1)],
    152: [(64, 3, 2), (128, 8, 2), (256, 36, 2), (512, 3, 1)],
}

MEAN_PIXEL = (123.68, 116.779, 103.939)  # PET/default_config.py:23
BN_EPS = 1e-5                            # slim resnet_arg_scope default


@dataclass
class UnitPlan:
    scope: str          # e.g. resnet_v1_50/block1/unit_1/bottleneck_v1
    depth_in: int
    depth: int
    depth_bottleneck: int
    stride: int         # stride actually applied (after output_stride logic)
    rate: int           # atrous rate of conv2
    has_shortcut_conv: bool


def resnet_units(depth: int = 50, output_stride: int = 16) -> List[UnitPlan]:
    """Unit list after slim's stack_blocks_dense(output_stride/4) logic."""
    if depth not in BLOCKS:
        raise ValueError("unsupported resnet depth %r" % (depth,))
    name = "resnet_v1_%d" % depth
    target = output_stride // 4       # root block already has stride 4
    current_stride, rate = 1, 1
    units: List[UnitPlan] = []
    depth_in = 64
    for bi, (base, n_units, block_stride) in enumerate(BLOCKS[depth], start=1):
        for ui in range(1, n_units + 1):
            s = block_stride if ui == n_units else 1
            d_out = base * 4
            if current_stride == target:
                unit_stride, unit_rate = 1, rate
                rate *= s
            else:
                unit_stride, unit_rate = s, 1
                current_stride *= s
            units.append(UnitPlan(
                scope="%s/block%d/unit_%d/bottleneck_v1" % (name, bi, ui),
                depth_in=depth_in, depth=d_out, depth_bottleneck=base,
                stride=unit_stride, rate=unit_rate,
                has_shortcut_conv=(depth_in != d_out)))
            depth_in = d_out
    return units


def same_out(n: int, s: int) -> int:
    return -(-n // s)


def feature_hw(h: int, w: int, depth: int = 50) -> Tuple[int, int]:
    """Spatial size of the backbone output (stride 16, ceil at every halving)."""
    for _ in range(4):
        h, w = same_out(h, 2), same_out(w, 2)
    return h, w


def scoremap_hw(h: int, w: int, deconv_stride: int = 2) -> Tuple[int, int]:
    """DGP/dataset.py:348-371 closed form: H_out = 2*ceil(H/16)."""
    fh, fw = feature_hw(h, w)
    return fh * deconv_stride, fw * deconv_stride


def conv_macs_per_frame(h: int, w: int, depth: int = 50, nj: int = 4,
                        with_locref: bool = False) -> int:
    """Algorithmic multiply-accumulates of the conv stack for one frame.

    Counts the reference's convolutions only (no BN/ReLU/pool), with the
    transposed-conv heads counted at their true tap count (9 taps per input
    pixel) -- the figure BASELINE.md section 3 quotes (35.61 GMAC at 640x480).
    """
    h1, w1 = same_out(h, 2), same_out(w, 2)
    macs = h1 * w1 * 7 * 7 * 3 * 64
    hh, ww = same_out(h1, 2), same_out(w1, 2)
    for u in resnet_units(depth):
        ho, wo = same_out(hh, u.stride), same_out(ww, u.stride)
        if u.has_shortcut_conv:
            macs += ho * wo * u.depth_in * u.depth
        macs += hh * ww * u.depth_in * u.depth_bottleneck
        macs += ho * wo * 9 * u.depth_bottleneck * u.depth_bottleneck
        macs += ho * wo * u.depth_bottleneck * u.depth
        hh, ww = ho, wo
    heads = nj * (3 if with_locref else 1)
    macs += hh * ww * 9 * 2048 * heads
    return macs


def launch_algorithmic_bytes(name: str, h: int, w: int, depth: int = 50, batch: int = 1, elem_bytes: float = 4.0) -> float:
    """Algorithmic HBM bytes of ONE backbone launch of the engine, from its profile name ("conv:<scope>[+<scope>...]|<kernel>",
    dgp_net_profile_launch): every tensor that enters or leaves the launch once (4 bytes per value: H2 cells are as large as fp32)
    plus the weights; elem_bytes = 2 for the 16-bit tier's H1 cells and fp16 weight cells.  Tensors that stay inside a fused launch are not counted: the shortcut tensor of conv3+shortcut, X' between
    conv3 and the next unit's conv1 (chain kernel: written once, not re-read), R2 between conv2 and conv3 (unit kernel).
    0.0 for names that are not backbone convs (stem, heads)."""
    body = name.split("|")[0]
    if not body.startswith("conv:"):
        return 0.0
    if body.endswith("/conv1+pool"):      # fused root block: uint8 frames in, pool output out, 7x7x3x64 weights
        h1, w1 = same_out(h, 2), same_out(w, 2)
        return float(batch * h * w * 3 + elem_bytes * batch * same_out(h1, 2) * same_out(w1, 2) * 64 + 4 * 7 * 7 * 3 * 64)
    toks = body[5:].split("+")
    units = {u.scope: (i, u) for i, u in enumerate(resnet_units(depth))}
    plan = resnet_units(depth)
    # geometry of every unit's input
    h1, w1 = same_out(h, 2), same_out(w, 2)
    hh, ww = same_out(h1, 2), same_out(w1, 2)
    geo = []
    for u in plan:
        ho, wo = same_out(hh, u.stride), same_out(ww, u.stride)
        geo.append((batch * hh * ww, batch * ho * wo))
        hh, ww = ho, wo
    convs = []                      # (unit index, which)
    for t in toks:
        if t == "shortcut":
            convs.append((convs[-1][0], "shortcut"))
            continue
        scope, _, which = t.rpartition("/")
        if scope not in units:
            return 0.0
        convs.append((units[scope][0], which))
    have = set(convs)
    total = 0.0
    for ui, which in convs:
        u = plan[ui]
        pin, pout = geo[ui]
        if which == "conv1":
            total += u.depth_in * u.depth_bottleneck                                    # weights
            total += pin * u.depth_bottleneck                                           # R1 out
            if (ui - 1, "conv3") not in have:
                total += pin * u.depth_in                                               # X in (chain: comes from the registers)
        elif which == "conv2":
            total += 9 * u.depth_bottleneck * u.depth_bottleneck
            total += pin * u.depth_bottleneck                                           # R1 in
            if (ui, "conv3") not in have:
                total += pout * u.depth_bottleneck                                      # R2 out (unit kernel: stays in registers)
        elif which == "conv3":
            total += u.depth_bottleneck * u.depth + pout * u.depth                      # weights, X' out
            if (ui, "conv2") not in have:
                total += pout * u.depth_bottleneck                                      # R2 in
            if (ui, "shortcut") in have:
                total += pin * u.depth_in                                               # the shortcut conv's input, read once
            elif u.has_shortcut_conv:
                total += pout * u.depth                                                 # separately computed shortcut tensor
            else:
                total += (pin if u.stride == 1 else pout) * u.depth                     # identity / subsampled residual
        elif which == "shortcut":
            total += u.depth_in * u.depth
            if (ui, "conv3") not in have:
                total += pin * u.depth_in + pout * u.depth
    return elem_bytes * total


def conv_algorithmic_bytes(h: int, w: int, depth: int = 50, batch: int = 1) -> dict:
    """{TF scope of a backbone conv: algorithmic HBM bytes of one launch at `batch` frames}: fp32 input read once
    + output written once + residual read once (conv3) + the weights.  Layer-by-layer execution, no fusion."""
    name = "resnet_v1_%d" % depth
    out = {}
    h1, w1 = same_out(h, 2), same_out(w, 2)
    out["conv:%s/conv1" % name] = 4.0 * (batch * h * w * 4 + batch * h1 * w1 * 64 + 7 * 7 * 4 * 64)
    hh, ww = same_out(h1, 2), same_out(w1, 2)
    for u in resnet_units(depth):
        ho, wo = same_out(hh, u.stride), same_out(ww, u.stride)
        pin, pout = batch * hh * ww, batch * ho * wo
        if u.has_shortcut_conv:
            out["conv:%s/shortcut" % u.scope] = 4.0 * (pin * u.depth_in + pout * u.depth + u.depth_in * u.depth)
        out["conv:%s/conv1" % u.scope] = 4.0 * (pin * u.depth_in + pin * u.depth_bottleneck + u.depth_in * u.depth_bottleneck)
        out["conv:%s/conv2" % u.scope] = 4.0 * (pin * u.depth_bottleneck + pout * u.depth_bottleneck +
                                                9 * u.depth_bottleneck * u.depth_bottleneck)
        res = pout * u.depth if u.has_shortcut_conv else (pin if u.stride == 1 else pout) * u.depth
        out["conv:%s/conv3" % u.scope] = 4.0 * (pout * u.depth_bottleneck + pout * u.depth + res + u.depth_bottleneck * u.depth)
        hh, ww = ho, wo
    return out
