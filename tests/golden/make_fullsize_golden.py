"""Oracle outputs at BASELINE's full sizes, cached so that the GPU suite compares WHOLE batches with the oracle without paying for the CPU
oracle at run time (minutes of fp32 CPU convolutions): run in the build container, commit the result.

  python tests/golden/make_fullsize_golden.py            -> tests/golden/fullsize_vectors.npz (+ tests/golden/reaching_frames/*.png)

  r101_*   BASELINE configs[4]'s per-GPU shape: ResNet-101, 1280 x 720, 20 keypoints, both heads, a batch of 16 seeded frames
           (synthetic.make_weights(101, 20, True, seed 41), synthetic.make_frames(16, 720, 1280, 20, seed 42)): soft-argmax mu, window
           indices, likelihoods of all 16 frames, and the same coordinates / indices from the oracle's graph evaluated in float64 (the anchor every fp32
           evaluation order is measured against: with logits of standard deviation 5 on a 90 x 160 map the softmax is broad and two fp32
           evaluations of this network differ by ~1e-3 px from each other, EXPERIMENTS.md section 2d); the scoremap and the locref map sampled at 16 384 seeded positions each + their
           maxima (the maps themselves are 18 / 37 MB)
  reach_*  BASELINE configs[1] / [0]: the 55 labeled frames of the reference's Reaching demo project (832 x 747, 15 of them 640 x 470 crops that
           frames.LabeledDirSource resizes) through ResNet-50 with seeded weights (make_weights(50, 5, True, seed 43)): x, y, likelihood,
           window indices per labeled image.  The frames are the reference project's DATA files (copied beside this script; the GPU box has no
           /root/reference), the numbers are the oracle's.
The oracle is test infrastructure (oracle/dgp_oracle.py); nothing here is product code."""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_FRAMES = "/root/reference/data/Reaching-Mackenzie-2018-08-30/labeled-data/reachingvideo1"


def main():
    import torch
    from deepgraphpose_amd.frames import LabeledDirSource
    from deepgraphpose_amd.synthetic import make_frames, make_weights
    from oracle import dgp_oracle as O
    torch.set_num_threads(8)
    out = {}
    # ---- configs[4] per-GPU shape
    wts = make_weights(101, 20, True, seed=41, head_std=0.05)
    frames = make_frames(16, 720, 1280, 20, seed=42)
    mu, idx, lik, sc_s, lr_s, mu64, idx64 = [], [], [], [], [], [], []
    rng = np.random.default_rng(7)
    n_sc, n_lr = 90 * 160 * 20, 90 * 160 * 40
    pos_sc = rng.integers(0, n_sc, size=(16, 1024))
    pos_lr = rng.integers(0, n_lr, size=(16, 1024))
    sc_max, lr_max = 0.0, 0.0
    for i in range(0, 16, 2):
        r = O.infer(frames[i:i + 2], wts, 101, 8.0, 1.0, 1)
        sc, lr = O.pose_heads(r["features"], wts, True)
        assert np.array_equal(sc, r["scmap"])
        mu.append(r["mu"]); idx.append(r["idx"]); lik.append(r["likelihoods"])
        r64 = O.infer(frames[i:i + 2], wts, 101, 8.0, 1.0, 1, dtype=np.float64)        # the accuracy anchor (EXPERIMENTS.md section 2d)
        mu64.append(r64["mu"]); idx64.append(r64["idx"])
        for k in range(2):
            sc_s.append(sc[k].reshape(-1)[pos_sc[i + k]]); lr_s.append(lr[k].reshape(-1)[pos_lr[i + k]])
        sc_max, lr_max = max(sc_max, float(np.abs(sc).max())), max(lr_max, float(np.abs(lr).max()))
        print("r101 frames", i, i + 1, "done", flush=True)
    out.update(r101_mu=np.concatenate(mu).astype(np.float32), r101_idx=np.concatenate(idx).astype(np.int32),
               r101_lik=np.concatenate(lik).astype(np.float32), r101_pos_sc=pos_sc.astype(np.int32), r101_pos_lr=pos_lr.astype(np.int32),
               r101_sc=np.stack(sc_s).astype(np.float32), r101_lr=np.stack(lr_s).astype(np.float32),
               r101_sc_max=np.float32(sc_max), r101_lr_max=np.float32(lr_max),
               r101_mu64=np.concatenate(mu64).astype(np.float64), r101_idx64=np.concatenate(idx64).astype(np.int32))
    # ---- the Reaching project's labeled frames
    dst = os.path.join(HERE, "reaching_frames")
    os.makedirs(dst, exist_ok=True)
    src = LabeledDirSource(REF_FRAMES)
    for f in src.files:
        shutil.copyfile(f, os.path.join(dst, os.path.basename(f)))
    src = LabeledDirSource(dst)
    wts = make_weights(50, 5, True, seed=43, head_std=0.05)
    xs, ys, ls, ix, xs64, ys64, ix64 = [], [], [], [], [], [], []
    for k, f in enumerate(src.files):
        r = O.infer(src._read(f)[None], wts, 50, 8.0, 1.0, 1)
        xs.append(r["x"][0]); ys.append(r["y"][0]); ls.append(r["likelihoods"][0]); ix.append(r["idx"][0])
        r64 = O.infer(src._read(f)[None], wts, 50, 8.0, 1.0, 1, dtype=np.float64)        # the anchor: seeded weights on real frames give broad maps
        xs64.append(r64["x"][0]); ys64.append(r64["y"][0]); ix64.append(r64["idx"][0])
        if k % 10 == 0:
            print("reaching frame", k, flush=True)
    out.update(reach_numbers=src.numbers.astype(np.int32), reach_x=np.stack(xs), reach_y=np.stack(ys), reach_lik=np.stack(ls),
               reach_idx=np.stack(ix).astype(np.int32), reach_x64=np.stack(xs64), reach_y64=np.stack(ys64), reach_idx64=np.stack(ix64).astype(np.int32))
    np.savez_compressed(os.path.join(HERE, "fullsize_vectors.npz"), **out)
    print("wrote", os.path.join(HERE, "fullsize_vectors.npz"))


if __name__ == "__main__":
    main()
