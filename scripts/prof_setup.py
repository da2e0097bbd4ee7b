"""Where estimate_pose's per-call set-up goes (cProfile of one call on 256 host frames, after a warm-up call)."""
import cProfile, pstats, os, sys, tempfile, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, yaml
from deepgraphpose_amd import weights_io
from deepgraphpose_amd.models import eval as E
from deepgraphpose_amd.synthetic import make_weights, make_frames
tmp = tempfile.mkdtemp()
proj = os.path.join(tmp, "proj"); train = os.path.join(proj, "dlc-models", "iteration-0", "DemoOct2-trainset95shuffle1", "train")
os.makedirs(train)
parts = ["a", "b", "c", "d"]
yaml.safe_dump(dict(Task="Demo", date="Oct2", iteration=0, TrainingFraction=[0.95], bodyparts=parts, skeleton=[], project_path=proj), open(os.path.join(proj, "config.yaml"), "w"))
yaml.safe_dump(dict(num_joints=4, all_joints_names=parts, net_type="resnet_50"), open(os.path.join(train, "pose_cfg.yaml"), "w"))
snap = weights_io.save_weights(os.path.join(train, "snapshot-step2-final--0"), make_weights(50, 4, False, seed=0, head_std=0.05))
frames = np.concatenate([make_frames(16, 480, 640, 4, seed=0)] * 16)
E.estimate_pose(os.path.join(proj, "config.yaml"), snap, frames[:64], os.path.join(tmp, "warm"), save_pose=False, batch_size=32)
pr = cProfile.Profile(); pr.enable()
E.estimate_pose(os.path.join(proj, "config.yaml"), snap, frames, os.path.join(tmp, "pred"), save_pose=False, batch_size=32)
pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(28); print(st.getvalue()[:6000])
