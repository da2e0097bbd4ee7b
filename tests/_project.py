"""Synthetic DLC project on disk (layout contract of the reference): config.yaml, pose_cfg.yaml, training .mat,
a frame stack as the 'video', and a seeded snapshot."""
import os

import numpy as np
import yaml


def make_project(root, nj=3, n_frames=40, hw=(64, 96), labeled=(3, 11, 19, 30), seed=0, nan_joint=True, depth=50):
    import scipy.io as sio
    from deepgraphpose_amd import weights_io
    from deepgraphpose_amd.synthetic import make_frames, make_weights
    parts = ["part%d" % i for i in range(nj)]
    proj = os.path.join(str(root), "Demo-me-2026-10-02")
    train = os.path.join(proj, "dlc-models", "iteration-0", "DemoOct2-trainset95shuffle1", "train")
    os.makedirs(train)
    os.makedirs(os.path.join(proj, "videos"))
    tsdir = os.path.join("training-datasets", "iteration-0", "UnaugmentedDataSet_DemoOct2")
    os.makedirs(os.path.join(proj, tsdir))
    frames = make_frames(n_frames, hw[0], hw[1], nj, seed=seed)
    np.save(os.path.join(proj, "videos", "clip.npy"), frames)
    cfg = dict(Task="Demo", date="Oct2", iteration=0, TrainingFraction=[0.95], bodyparts=parts,
               skeleton=[[parts[0], parts[1]], [parts[1], parts[2]]] if nj >= 3 else [], project_path=proj, scorer="me",
               pcutoff=0.4, video_sets={"videos/clip.npy": {"crop": "0, %d, 0, %d" % (hw[1], hw[0])}})
    with open(os.path.join(proj, "config.yaml"), "w") as f:
        yaml.safe_dump(cfg, f)
    mat_rel = os.path.join(tsdir, "Demo_me95shuffle1.mat")
    pose = dict(all_joints=[[i] for i in range(nj)], all_joints_names=parts, dataset=mat_rel, net_type="resnet_%d" % depth,
                num_joints=nj, pos_dist_thresh=17, location_refinement=True, locref_huber_loss=True,
                locref_loss_weight=0.05, locref_stdev=7.2801, project_path=proj, init_weights="resnet_v1_50.ckpt")
    with open(os.path.join(train, "pose_cfg.yaml"), "w") as f:
        yaml.safe_dump(pose, f)
    rng = np.random.default_rng(seed + 1)
    items = np.zeros((1, len(labeled)), dtype=[("image", "O"), ("size", "O"), ("joints", "O")])
    for k, fi in enumerate(labeled):
        joints = []
        for j in range(nj):
            if nan_joint and k == 1 and j == nj - 1:
                continue                                    # unlabeled joint -> NaN target
            joints.append([j, int(rng.integers(8, hw[1] - 8)), int(rng.integers(8, hw[0] - 8))])
        items[0, k]["image"] = np.array(["labeled-data/clip/img%03d.png" % fi])
        items[0, k]["size"] = np.array([[3, hw[0], hw[1]]])
        jj = np.empty((1, 1), dtype=object)
        jj[0, 0] = np.array(joints, dtype=np.int64)
        items[0, k]["joints"] = jj
    sio.savemat(os.path.join(proj, mat_rel), {"dataset": items})
    # human labels in DLC's CollectedData csv layout + train/test split pickle + the labeled frames as PNGs
    import pickle
    from PIL import Image
    os.makedirs(os.path.join(proj, "labeled-data", "clip"))
    rows = ["scorer," + ",".join(["me"] * (2 * nj)), "bodyparts," + ",".join(p for p in parts for _ in range(2)),
            "coords," + ",".join(["x", "y"] * nj)]
    for k, fi in enumerate(labeled):
        Image.fromarray(frames[fi]).save(os.path.join(proj, "labeled-data", "clip", "img%03d.png" % fi))
        xy = {int(r[0]): (r[1], r[2]) for r in items[0, k]["joints"][0, 0]}
        rows.append("labeled-data/clip/img%03d.png," % fi + ",".join(
            ("%d,%d" % xy[j]) if j in xy else "," for j in range(nj)))
    with open(os.path.join(proj, tsdir, "CollectedData_me.csv"), "w") as f:
        f.write("\n".join(rows) + "\n")
    with open(os.path.join(proj, tsdir, "Documentation_data-Demo_95shuffle1.pickle"), "wb") as f:
        pickle.dump([{}, np.arange(len(labeled) - 1), np.array([len(labeled) - 1]), 0.95], f)
    wts = make_weights(depth, nj, True, seed=seed, head_std=0.05)
    weights_io.save_weights(os.path.join(train, "snapshot-step0-final--0"), wts)
    return proj, frames, wts
