#!/usr/bin/env python3
"""Counterpart of the reference's demo/run_dgp_demo.py (same flags, same project layout, same step order):

  step 0  fit_dlc                    -> snapshot-step0-final--0   (DLC baseline; needs resnet_v1_<d>.ckpt in $DGP_PRETRAINED_DIR, or pass --dlcsnapshot)
  step 1  fit_dgp_labeledonly        -> snapshot-step1-final--0
  step 2  fit_dgp (gm2=1, gm3=3)     -> snapshot-step2-final--0
  step 3  plot_dgp / estimate_pose   -> <proj>/videos_pred/<video>_labeled.{csv,h5[,mp4]}

    python demo/run_dgp_demo.py --dlcpath <project> --dlcsnapshot snapshot-step0-final--0 [--test]

Videos are taken from <proj>/videos_dgp/ (real videos need moviepy; directories of frames and .npy stacks work
without a decoder).  Snapshots are .npz files keyed by TF variable names.
"""
import argparse
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from deepgraphpose_amd.models.eval import plot_dgp                                   # noqa: E402
from deepgraphpose_amd.models.fitdgp import fit_dgp, fit_dgp_labeledonly, fit_dlc    # noqa: E402
from deepgraphpose_amd.models.fitdgp_util import get_snapshot_path                   # noqa: E402

if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--dlcpath", type=str, default=None, help="the path for the DLC project")
    parser.add_argument("--dlcsnapshot", type=str, default=None, help="use snapshot for dlc (skips step 0)")
    parser.add_argument("--shuffle", type=int, default=1, help="Project shuffle")
    parser.add_argument("--batch_size", type=int, default=10, help="size of the batch")
    parser.add_argument("--test", action="store_true", default=False)
    args = parser.parse_args()
    dlcpath, shuffle, batch_size, test = args.dlcpath, args.shuffle, args.batch_size, args.test
    if dlcpath is None:
        raise SystemExit("--dlcpath is required (the bundled Reaching demo project ships no videos or checkpoints)")

    # step 0
    if args.dlcsnapshot is None:
        snapshot = "resnet_v1_50.ckpt"
        print("\n\n" + "=" * 80 + "\n|   step 0: fit_dlc\n" + "=" * 80)
        fit_dlc(snapshot=snapshot, dlcpath=dlcpath, shuffle=shuffle, step=0, saveiters=1000, displayiters=100,
                maxiters=2 if test else 200000)
        snapshot = "snapshot-step0-final--0"
    else:
        snapshot = args.dlcsnapshot

    # step 1
    print("\n\n" + "=" * 80 + "\n|   step 1: DGP with labeled frames only\n" + "=" * 80)
    fit_dgp_labeledonly(snapshot=snapshot, dlcpath=dlcpath, shuffle=shuffle, step=1, saveiters=1000, displayiters=1 if test else 100,
                        maxiters=2 if test else 50000)
    snapshot = "snapshot-step1-final--0"

    # step 2
    print("\n\n" + "=" * 80 + "\n|   step 2: DGP\n" + "=" * 80)
    fit_dgp(snapshot=snapshot, dlcpath=dlcpath, batch_size=batch_size, shuffle=shuffle, step=2, saveiters=1000,
            displayiters=1 if test else 100, maxiters=5 if test else 200000, gm2=1, gm3=3)
    snapshot = "snapshot-step2-final--0"

    # step 3
    print("\n\n" + "=" * 80 + "\n|   step 3: predict\n" + "=" * 80)
    snapshot_path, cfg_yaml = get_snapshot_path(snapshot, dlcpath, shuffle=shuffle)
    video_path = os.path.join(dlcpath, "videos_dgp")
    if not os.path.exists(video_path):
        video_path = os.path.join(dlcpath, "videos")
    videos = [os.path.join(video_path, f) for f in sorted(os.listdir(video_path))
              if os.path.isdir(os.path.join(video_path, f)) or f.rsplit(".", 1)[-1] in ("avi", "mp4", "mov", "mkv", "npy")]
    out_dir = os.path.join(dlcpath, "videos_pred")
    for video in videos[:1] if test else videos:
        print("video file:", video)
        out = plot_dgp(video_file=str(video), output_dir=out_dir, proj_cfg_file=str(cfg_yaml), dgp_model_file=str(snapshot_path),
                       shuffle=shuffle)
        print("wrote", out)
