"""A PEAKY full-size fixture for BASELINE configs[4]'s per-GPU shape (ResNet-101, 1280 x 720, 20 keypoints, 16 frames): run in the build
container, commit the result.

  python tests/golden/make_fullsize_peaky_golden.py      -> tests/golden/fullsize_peaky_vectors.npz

Why a second fixture.  make_fullsize_golden.py uses seeded RANDOM heads: logits of standard deviation ~5-8 over a 90 x 160 map, a broad
multi-modal softmax -- the regime in which ANY two fp32 evaluations of the network differ by ~1e-3 px (the fp32 CPU oracle is 1.4e-3 px
from its own float64 evaluation), so that test has to gate against the float64 anchor.  A trained network is confident: one compact
peak per keypoint, margins of tens of logits to everything else.  This fixture builds such heads WITHOUT training: the part_pred head of
keypoint j is a matched filter -- the mean (centred) block4 feature vector the oracle computes at the cells where synthetic.make_frames
draws blob j -- scaled so that the peak logit is ~ +100 above the map's bulk (taps (0..1, 0..1) of the 3 x 3 transposed conv with weights
1, rho, rho, rho^2: the four output phases of a feature cell are NOT tied).  The backbone stays the seeded random ResNet-101.

What the fixture pins: soft-argmax coordinates, window indices and likelihoods of all 16 x 20 (frame, keypoint) pairs from the fp32 oracle
and from its float64 evaluation, and the mask of WELL-CONDITIONED pairs (oracle-to-float64 distance < 1e-4 px: no two cells near a tie).
On those the GPU test asserts the north-star gate literally (< 1e-3 px, index bit-exact); a near-tie between two cells is ill-conditioned
in fp32 whatever the network (the coordinate moves by 8 px x p (1 - p) x the error of the logit gap), those pairs stay under the anchored gate.
The prototype matrix P [2048, 20] (float32) is stored in the fixture; the test rebuilds the head weights from it.
The oracle is test infrastructure (oracle/dgp_oracle.py); nothing here is product code."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
DEPTH, NJ, T, H, W = 101, 20, 16, 720, 1280
SEED_W, SEED_F, BETA, RHO = 41, 42, 5.0, 0.75


def blob_cells(fh, fw):
    """feature cells (stride 16) of the blobs synthetic.make_frames(T, H, W, NJ, SEED_F) draws: [T, NJ, 2]"""
    rng = np.random.default_rng(SEED_F + 1000)
    cy = rng.uniform(0.2 * H, 0.8 * H, size=NJ)
    cx = rng.uniform(0.2 * W, 0.8 * W, size=NJ)
    out = np.zeros((T, NJ, 2), np.int64)
    for t in range(T):
        for j in range(NJ):
            py = cy[j] + 0.08 * H * np.sin(0.05 * t + j)
            px = cx[j] + 0.08 * W * np.cos(0.04 * t + 2 * j)
            out[t, j] = (min(int(py / 16), fh - 1), min(int(px / 16), fw - 1))
    return out


def head_from_prototypes(P, pm, beta=BETA, rho=RHO):
    """part_pred weights [3, 3, nj, C] / biases [nj] of the matched-filter head (see the module docstring); float32.
    pm [nj] = P^T (mean feature): the bias removes the bulk's offset, so a blob cell sits at ~ +beta and the map's bulk around 0"""
    C, nj = P.shape
    w = np.zeros((3, 3, nj, C), np.float32)
    for a in (0, 1):
        for b in (0, 1):
            w[a, b] = (np.float32(beta * rho ** (a + b)) * P).T
    return w, (-np.float32(beta) * pm).astype(np.float32)


def main():
    import torch
    from deepgraphpose_amd.synthetic import make_frames, make_weights
    from oracle import dgp_oracle as O
    torch.set_num_threads(8)
    wts = make_weights(DEPTH, NJ, True, seed=SEED_W, head_std=0.05)
    frames = make_frames(T, H, W, NJ, seed=SEED_F)
    cache = os.environ.get("DGP_PEAKY_FEATURE_CACHE", "")          # (re-runs while choosing BETA: the features take minutes)
    if cache and os.path.exists(cache):
        z = np.load(cache)
        f32, f64 = z["f32"], z["f64"]
    else:
        f32, f64 = [], []
        for i in range(0, T, 2):
            f32.append(O.resnet_features(frames[i:i + 2], wts, DEPTH))
            f64.append(O.resnet_features(frames[i:i + 2], wts, DEPTH, dtype=np.float64))
            print("features of frames", i, i + 1, flush=True)
        f32, f64 = np.concatenate(f32), np.concatenate(f64)
        if cache:
            np.savez(cache, f32=f32, f64=f64)
    fh, fw = f64.shape[1:3]
    m = f64.reshape(-1, f64.shape[-1]).mean(0)
    cells = blob_cells(fh, fw)
    P = np.stack([np.mean([f64[t, cells[t, j, 0], cells[t, j, 1]] - m for t in range(T)], 0) for j in range(NJ)], 1)
    P = (P / (P ** 2).sum(0, keepdims=True)).astype(np.float32)
    pm = (P.astype(np.float64) * m[:, None]).sum(0).astype(np.float32)
    w, b = head_from_prototypes(P, pm)
    wt = dict(wts)
    wt["pose/part_pred/block4/weights"], wt["pose/part_pred/block4/biases"] = w, b
    mu, idx, lik, mu64, idx64, lik64 = [], [], [], [], [], []
    for i in range(0, T, 2):
        s32, _ = O.pose_heads(f32[i:i + 2], wt, False)
        s64, _ = O.pose_heads(f64[i:i + 2], wt, False)
        m32, _ = O.argmax_2d_from_cm(s32, 1.0, 1)
        m64, _ = O.argmax_2d_from_cm(s64, 1.0, 1, dtype=np.float64)
        assert np.isfinite(s32).all() and float(s32.max()) < 80.0, float(s32.max())      # (the reference's exp(m) / (exp(m) + 1) must not overflow in fp32)
        for k in range(2):
            ix, lk = O.likelihood_window(s32[k], m32[k]); lik.append(lk); idx.append(ix)
            ix, lk = O.likelihood_window(s64[k], m64[k]); lik64.append(lk); idx64.append(ix)
        mu.append(m32); mu64.append(m64)
    mu, mu64 = np.concatenate(mu).astype(np.float32), np.concatenate(mu64).astype(np.float64)
    d = np.abs(mu.astype(np.float64) - mu64).max(-1) * 8.0
    well = d < 1e-4
    print("oracle fp32 vs float64: max %.3g px, median %.3g px; well-conditioned pairs %d of %d" % (d.max(), np.median(d), int(well.sum()), well.size))
    np.savez_compressed(os.path.join(HERE, "fullsize_peaky_vectors.npz"), P=P, pm=pm, beta=np.float32(BETA), rho=np.float32(RHO),
                        mu=mu, idx=np.stack(idx).astype(np.int32), lik=np.stack(lik).astype(np.float32), mu64=mu64,
                        idx64=np.stack(idx64).astype(np.int32), well=well, cells=cells.astype(np.int32))
    print("wrote", os.path.join(HERE, "fullsize_peaky_vectors.npz"))


if __name__ == "__main__":
    main()
