"""The bottleneck chain kernel (csrc/dgp_chain.hip; include/dgp_hip.h dgp_chain_h2): conv3 of unit k + shortcut + ReLU and conv1 of
unit k + 1 in one launch, conv1 fed from conv3's accumulator registers.  Reference graph: PET/nnet/pose_net.py:46-52 -> slim
resnet_v1 `bottleneck` (conv3 without activation, relu(shortcut + residual)), then the next unit's conv1 (1x1 + BN + ReLU).

Layer tests against float64 on the values the H2 cells hold (tolerance 2e-5 relative, as for every other conv layer test), and
against the layer-by-layer kernels (dgp_conv2d_h2 twice) which compute the same products in another order."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [
    # N, Ho, Wo, C, C1, CIN2, res_mode
    (2, 17, 23, 64, 64, 0, 1),        # identity unit of block1 (ragged last tile)
    (3, 30, 40, 64, 64, 64, 0),       # block1 unit_1: conv3 + shortcut conv K-concatenated, then unit_2's conv1
    (2, 15, 20, 64, 128, 0, 2),       # stride-2 unit at the end of block1 -> block2 unit_1's conv1
    (2, 15, 20, 128, 128, 0, 1),      # identity unit of block2
    (2, 8, 10, 128, 256, 0, 2),       # end of block2 -> block3 unit_1's conv1
    (2, 30, 40, 256, 256, 0, 1),      # identity unit of block3 (65-KiB weight chunks: two-slot ring)
    (32, 30, 40, 256, 256, 0, 1),     # ... at batch 32 (480 tiles over 256 persistent workgroups)
    (32, 120, 160, 64, 64, 0, 1),     # full batch-32 640x480 shape of block1 (4800 tiles: persistent workgroups wrap the chunk ring)
]


def _case_tensors(case):
    N, Ho, Wo, C, C1, CIN2, res_mode = case
    seed = abs(hash(case)) % (2 ** 31)
    g = torch.Generator(device="cuda").manual_seed(seed)
    rng = np.random.default_rng(seed)
    C4 = 4 * C
    r2 = torch.relu(torch.randn((N, Ho, Wo, C), device="cuda", generator=g)) * 3.0
    if res_mode == 0:
        src2 = torch.relu(torch.randn((N, Ho, Wo, CIN2), device="cuda", generator=g)) * 2.5
    elif res_mode == 1:
        src2 = torch.relu(torch.randn((N, Ho, Wo, C4), device="cuda", generator=g)) * 2.0
    else:
        src2 = torch.relu(torch.randn((N, 2 * Ho - 1, 2 * Wo, C4), device="cuda", generator=g)) * 2.0
    w3 = (rng.standard_normal((C + CIN2, C4)) / np.sqrt(C + CIN2)).astype(np.float32)
    w1 = (rng.standard_normal((C4, C1)) / np.sqrt(C4)).astype(np.float32)
    s3 = None if res_mode == 0 else (1 + 0.1 * rng.standard_normal(C4)).astype(np.float32)      # (the K-concatenated panel is BN-folded)
    b3 = (0.1 * rng.standard_normal(C4)).astype(np.float32)
    s1 = (1 + 0.1 * rng.standard_normal(C1)).astype(np.float32)
    b1 = (0.1 * rng.standard_normal(C1)).astype(np.float32)
    return r2, src2, w3, s3, b3, w1, s1, b1


@pytest.mark.parametrize("case", CASES)
def test_chain_matches_float64_and_the_layer_kernels(lib_built, case):
    from deepgraphpose_amd import engine
    N, Ho, Wo, C, C1, CIN2, res_mode = case
    r2, src2, w3, s3, b3, w1, s1, b1 = _case_tensors(case)
    C4 = 4 * C
    dd = lambda a: torch.from_numpy(np.asarray(a)).double().cuda()
    if res_mode == 0:                                   # shared scale of the two K sources
        e_r2 = e_s2 = engine.h2_exp_for(max(float(r2.abs().max()), float(src2.abs().max())))
    else:
        e_r2, e_s2 = engine.h2_exp_for(float(r2.abs().max())), engine.h2_exp_for(float(src2.abs().max()))
    r2h, s2h = engine.f32_to_h2(r2, e_r2), engine.f32_to_h2(src2, e_s2)
    r2q, s2q = engine.h2_to_f32(r2h, e_r2).double(), engine.h2_to_f32(s2h, e_s2).double()
    M = N * Ho * Wo
    if res_mode == 0:
        acc = torch.cat([r2q.reshape(M, C), s2q.reshape(M, CIN2)], 1) @ dd(w3) + dd(b3)
    else:
        sc = s2q if res_mode == 1 else s2q[:, ::2, ::2]
        acc = (r2q.reshape(M, C) @ dd(w3)) * dd(s3) + dd(b3) + sc.reshape(M, C4)
    x_ref = torch.relu(acc)
    e_x = engine.h2_exp_for(float(x_ref.max()))
    # conv1 consumes the 22-bit cells of X' (what the kernel feeds its MFMAs and what the layer-by-layer path re-reads)
    xq = engine.h2_to_f32(engine.f32_to_h2(x_ref.float().reshape(N, Ho, Wo, C4), e_x), e_x).double().reshape(M, C4)
    r1_ref = torch.relu((xq @ dd(w1)) * dd(s1) + dd(b1))
    e_r1 = engine.h2_exp_for(float(r1_ref.max()))
    xo, r1, xrng, r1rng = engine.chain_h2(r2h, e_r2, s2h, e_s2, w3, s3, b3, w1, s1, b1, res_mode, e_x, e_r1)
    x_out = engine.h2_to_f32(xo, e_x).double().reshape(M, C4)
    r1_out = engine.h2_to_f32(r1, e_r1).double().reshape(M, C1)
    ex = float((x_out - x_ref).abs().max() / x_ref.abs().max())
    er = float((r1_out - r1_ref).abs().max() / r1_ref.abs().max())
    assert ex < 2e-5 and er < 2e-5, (case, ex, er)
    assert abs(float(xrng.max()) - float(x_ref.max())) <= 1e-4 * float(x_ref.max())
    assert abs(float(r1rng.max()) - float(r1_ref.max())) <= 1e-4 * float(r1_ref.max())
    if M > 100000:
        return
    # the layer-by-layer kernels on the same cells (identity / subsample cases: dgp_conv2d_h2 with an H2 residual, then conv1)
    if res_mode != 0:
        y3, _ = engine.conv2d_h2(r2h, e_r2, w3.reshape(1, 1, C, C4), scale=s3, bias=b3, residual=s2h, res_stride=res_mode, res_is_h2=True,
                                 res_exp=e_s2, relu=True, y_is_h2=True, y_exp=e_x, out_hw=(Ho, Wo))
        y1, _ = engine.conv2d_h2(y3, e_x, w1.reshape(1, 1, C4, C1), scale=s1, bias=b1, relu=True, y_is_h2=True, y_exp=e_r1)
        a = engine.h2_to_f32(y3, e_x).double().reshape(M, C4)
        b = engine.h2_to_f32(y1, e_r1).double().reshape(M, C1)
        assert float((a - x_out).abs().max() / x_ref.abs().max()) < 4e-6
        assert float((b - r1_out).abs().max() / r1_ref.abs().max()) < 4e-6


UNIT_CASES = [
    # N, H, W, CIN2, res_mode            (C = C1 = 64: block1)
    (2, 13, 21, 0, 1),        # ragged in both directions (4 x 16 tiles): halo zeros at every image edge
    (1, 4, 16, 0, 1),         # exactly one tile
    (2, 19, 33, 64, 0),       # unit_1: conv3 + shortcut conv K-concatenated (8 x 16 tiles)
    (32, 120, 160, 0, 1),     # the batch-32 640x480 shape of block1/unit_2
    (4, 120, 160, 64, 0),     # block1/unit_1 at full frame size
]


@pytest.mark.parametrize("case", UNIT_CASES)
def test_unit_kernel_matches_float64(lib_built, case):
    """conv2 (3x3 SAME) -> conv3 (+ shortcut) -> ReLU -> next conv1 in one launch vs float64 on the 22-bit values each stage consumes."""
    from deepgraphpose_amd import engine
    N, H, W, CIN2, res_mode = case
    C = C1 = 64
    C4 = 4 * C
    seed = abs(hash(case)) % (2 ** 31)
    g = torch.Generator(device="cuda").manual_seed(seed)
    rng = np.random.default_rng(seed)
    r1 = torch.relu(torch.randn((N, H, W, C), device="cuda", generator=g)) * 3.0
    src2 = torch.relu(torch.randn((N, H, W, CIN2 if res_mode == 0 else C4), device="cuda", generator=g)) * 2.0
    w2 = (rng.standard_normal((3, 3, C, C)) / np.sqrt(9 * C)).astype(np.float32)
    s2 = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32)
    b2 = (0.1 * rng.standard_normal(C)).astype(np.float32)
    w3 = (rng.standard_normal((C + CIN2, C4)) / np.sqrt(C + CIN2)).astype(np.float32)
    s3 = None if res_mode == 0 else (1 + 0.1 * rng.standard_normal(C4)).astype(np.float32)
    b3 = (0.1 * rng.standard_normal(C4)).astype(np.float32)
    w1 = (rng.standard_normal((C4, C1)) / np.sqrt(C4)).astype(np.float32)
    s1 = (1 + 0.1 * rng.standard_normal(C1)).astype(np.float32)
    b1 = (0.1 * rng.standard_normal(C1)).astype(np.float32)
    dd = lambda a: torch.from_numpy(np.asarray(a)).double().cuda()
    q22 = lambda t, e: engine.h2_to_f32(engine.f32_to_h2(t.float().contiguous(), e), e).double()
    e_r1 = engine.h2_exp_for(float(r1.max()))
    r1h = engine.f32_to_h2(r1, e_r1)
    r1q = engine.h2_to_f32(r1h, e_r1).double()
    xp = torch.zeros((N, H + 2, W + 2, C), dtype=torch.float64, device="cuda")
    xp[:, 1:H + 1, 1:W + 1] = r1q
    cols = torch.stack([xp[:, a:a + H, b:b + W] for a in range(3) for b in range(3)], 3).reshape(N * H * W, 9 * C)
    r2_ref = torch.relu((cols @ dd(w2.reshape(9 * C, C))) * dd(s2) + dd(b2))
    M = N * H * W
    if res_mode == 0:
        e_r2 = e_s2 = engine.h2_exp_for(max(float(r2_ref.max()), float(src2.max())))
    else:
        e_r2, e_s2 = engine.h2_exp_for(float(r2_ref.max())), engine.h2_exp_for(float(src2.max()))
    s2h = engine.f32_to_h2(src2, e_s2)
    s2q = engine.h2_to_f32(s2h, e_s2).double()
    r2q = q22(r2_ref.reshape(N, H, W, C), e_r2).reshape(M, C)
    if res_mode == 0:
        acc = torch.cat([r2q, s2q.reshape(M, CIN2)], 1) @ dd(w3) + dd(b3)
    else:
        acc = (r2q @ dd(w3)) * dd(s3) + dd(b3) + s2q.reshape(M, C4)
    x_ref = torch.relu(acc)
    e_x = engine.h2_exp_for(float(x_ref.max()))
    xq = q22(x_ref.reshape(N, H, W, C4), e_x).reshape(M, C4)
    r1o_ref = torch.relu((xq @ dd(w1)) * dd(s1) + dd(b1))
    e_o = engine.h2_exp_for(float(r1o_ref.max()))
    xo, r1o, r2rng, xrng, orng = engine.unit_h2(r1h, e_r1, s2h, e_s2, w2, s2, b2, e_r2, w3, s3, b3, w1, s1, b1, res_mode, e_x, e_o)
    x_out = engine.h2_to_f32(xo, e_x).double().reshape(M, C4)
    r1_out = engine.h2_to_f32(r1o, e_o).double().reshape(M, C1)
    ex = float((x_out - x_ref).abs().max() / x_ref.abs().max())
    er = float((r1_out - r1o_ref).abs().max() / r1o_ref.abs().max())
    assert ex < 2e-5 and er < 2e-5, (case, ex, er)
    for got, want in ((r2rng, r2_ref), (xrng, x_ref), (orng, r1o_ref)):
        assert abs(float(got.max()) - float(want.max())) <= 1e-4 * float(want.max())


def test_chain_rejects_shapes_without_a_kernel_instance(lib_built):
    from deepgraphpose_amd import _lib, engine
    r2 = torch.zeros((1, 4, 4, 32), device="cuda")
    x = torch.zeros((1, 4, 4, 128), device="cuda")
    w3, w1 = np.zeros((32, 128), np.float32), np.zeros((128, 32), np.float32)
    with pytest.raises(_lib.DgpError, match="no kernel instance"):
        engine.chain_h2(r2, 0, x, 0, w3, None, None, w1, None, None, 1, 0, 0)


def test_network_is_the_same_with_and_without_chains(lib_built):
    """Whole-net A/B in one process is not possible (DGP_CHAIN is read once), so compare with a child process."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, numpy as np, torch
from deepgraphpose_amd.engine import DGPNet
from deepgraphpose_amd.synthetic import make_frames, make_weights
wts = make_weights(50, 4, False, seed=5, head_std=0.05)
frames = torch.from_numpy(make_frames(3, 96, 128, 4, seed=6)).cuda()
net = DGPNet(50, 4, 96, 128, max_batch=3); net.load_weights(wts)
sc = net.forward(frames)[0]
mu, conf, idx = net.infer(frames)
np.savez(sys.argv[1], sc=sc.cpu().numpy(), mu=mu.cpu().numpy(), idx=idx.cpu().numpy())
'''
    import tempfile
    outs = []
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        for flag in ("1", "0"):
            env = dict(os.environ, DGP_CHAIN=flag, PYTHONPATH=root)
            path = os.path.join(td, "o%s.npz" % flag)
            subprocess.check_call([sys.executable, "-c", code, path], env=env, cwd=root)
            outs.append(dict(np.load(path)))
    a, b = outs
    assert np.array_equal(a["idx"], b["idx"])
    assert np.abs(a["mu"] - b["mu"]).max() * 8.0 < 2e-4            # px; both within 1e-3 of the oracle (test_parity_gpu.py)
    assert np.abs(a["sc"] - b["sc"]).max() <= 2e-5 * np.abs(b["sc"]).max()


def test_other_workgroup_shapes_of_the_fused_kernels_in_child_processes(lib_built, tuning_build):
    """The unit / chain kernels have alternative workgroup shapes behind tuning knobs (DGP_UNIT_CFG: four weight-loader waves, ten-row tiles,
    the halo wave swapped between the two unit instances; DGP_CHAIN_CFG: row blocks per wave, loader waves, ring depth).  The arithmetic per
    pixel does not depend on the shape: the layer tests of this file must pass unchanged under each of them (tuning builds only)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for var, values in (("DGP_UNIT_CFG", ("1", "2", "3")), ("DGP_CHAIN_CFG", ("1", "4", "5"))):
        for v in values:
            env = dict(os.environ, PYTHONPATH=root)
            env[var] = v
            r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_chain_gpu.py"), "-q", "-m", "gpu", "-k",
                                "test_chain_matches_float64 or test_unit_kernel_matches_float64"], env=env, cwd=root,
                               capture_output=True, text=True, timeout=900)
            assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, (var, v, r.stdout[-1500:] + r.stderr[-500:])
