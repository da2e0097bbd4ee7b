"""GPU tests of the round-2 boundary pieces: pose_net drop-in, packed trajectory output, the demo script end to end, RCCL on
hardware (a fresh child process with RANK=0 WORLD_SIZE=1), frame-sharded estimate_pose, evaluate_dgp(loc_ref_calc='dgp')."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PX_TOL = 1e-3


def _child_env(**extra):
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update({k: str(v) for k, v in extra.items()})
    return env


def test_pose_net_drop_in_matches_engine_and_oracle(lib_built):
    """PoseNet(cfg).extract_features / prediction_layers / test and prediction_layer / dgp_prediction_layer
    (PET/nnet/pose_net.py:18-90, fitdgp_util.py:18-74) vs the fused engine and the CPU oracle."""
    from deepgraphpose_amd.config import AttrDict
    from deepgraphpose_amd.engine import DGPNet
    from deepgraphpose_amd.nnet.pose_net import PoseNet, dgp_prediction_layer, prediction_layer
    from deepgraphpose_amd.synthetic import make_frames, make_weights
    from oracle import dgp_oracle as O
    nj = 3
    wts = make_weights(50, nj, True, seed=12, head_std=0.05)
    frames = make_frames(3, 96, 128, nj, seed=12)
    cfg = AttrDict(net_type="resnet_50", num_joints=nj, location_refinement=True, intermediate_supervision=False,
                   mean_pixel=[123.68, 116.779, 103.939], weight_decay=0.0001)
    pn = PoseNet(cfg)
    assert cfg.output_stride == 16 and cfg.deconvolutionstride == 2          # pose_net.py:30-34 defaults
    with pytest.raises(ValueError):
        pn.extract_features(frames)                                           # no variables bound yet
    pn.restore(wts)
    net, end_points = pn.extract_features(frames.astype(np.float32))          # the reference feeds float frames
    ref = O.infer(frames, wts, 50)
    assert tuple(net.shape) == (3, 6, 8, 2048) and "resnet_v1_50/block4" in end_points
    assert np.abs(net.cpu().numpy() - ref["features"]).max() <= 2e-5 * np.abs(ref["features"]).max()
    heads = pn.prediction_layers(net, end_points)
    sc_ref, loc_ref = O.pose_heads(ref["features"], wts, True)
    assert np.abs(heads["part_pred"].cpu().numpy() - sc_ref).max() <= 5e-5 * np.abs(sc_ref).max()
    assert np.abs(heads["locref"].cpu().numpy() - loc_ref).max() <= 5e-5 * np.abs(loc_ref).max()
    # the fused engine gives the same maps
    eng = DGPNet(50, nj, 96, 128, max_batch=4, with_locref=True)
    eng.load_weights(wts)
    sc, loc = eng.forward(torch.from_numpy(frames).cuda(), want_locref=True)
    assert torch.allclose(sc, heads["part_pred"], atol=2e-5 * float(sc.abs().max()))
    assert torch.allclose(loc, heads["locref"], atol=2e-5 * float(loc.abs().max()))
    # free functions, explicit variables
    p2 = prediction_layer(cfg, net, "part_pred", nj, wts)
    assert torch.equal(p2, heads["part_pred"])
    p3 = dgp_prediction_layer(wts["pose/part_pred/block4/weights"], wts["pose/part_pred/block4/biases"], cfg, net, "part_pred", nj,
                              True, 2048, True)
    assert torch.equal(p3, heads["part_pred"])
    with pytest.raises(ValueError):
        dgp_prediction_layer(None, None, cfg, net, "part_pred", nj, False, 2048, True)
    t = pn.test(frames)
    assert set(t) == {"part_prob", "locref"} and float(t["part_prob"].min()) >= 0 and float(t["part_prob"].max()) <= 1


def test_infer_packed_equals_infer(lib_built):
    from deepgraphpose_amd import dist as ddist
    from deepgraphpose_amd.engine import DGPNet
    from deepgraphpose_amd.synthetic import make_frames, make_weights
    nj = 4
    net = DGPNet(50, nj, 64, 96, max_batch=8)
    net.load_weights(make_weights(50, nj, False, seed=2, head_std=0.05))
    ft = torch.from_numpy(make_frames(5, 64, 96, nj, seed=2)).cuda()
    mu, conf, idx = [t.clone() for t in net.infer(ft)]
    traj = torch.full((9, nj, 5), -7.0, dtype=torch.float32, device="cuda")
    net.infer_packed(ft, traj[2:7])                          # a slice of a longer trajectory, written in place
    m2, c2, i2 = ddist.unpack_keypoints(traj[2:7])
    assert torch.equal(m2, mu) and torch.equal(c2, conf) and torch.equal(i2, idx)
    assert float(traj[:2].min()) == -7.0 and float(traj[7:].max()) == -7.0      # nothing outside the slice touched
    with pytest.raises(Exception):
        net.infer_packed(ft, traj[:4])


def test_rccl_all_gather_in_a_fresh_child_process(lib_built, tmp_path):
    """init_process_group('nccl') (= RCCL) + all_gather_into_tensor + all_reduce on the hardware: a child process launched with
    RANK=0 WORLD_SIZE=1 before anything touches the GPU there, running the product's own dist.gather_trajectory and
    average_gradients on device tensors."""
    code = r'''
import json, os, sys, torch
from deepgraphpose_amd import dist as ddist
rank, local, world = ddist.init_from_env("nccl")
import torch.distributed as dist
assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
dev = torch.device("cuda", local)
g = torch.Generator(device="cuda").manual_seed(3)
local_traj = torch.randn((37, 4, 5), device=dev, generator=g)
full = ddist.gather_trajectory(local_traj, 37)
grads = torch.randn(1 << 20, device=dev, generator=g)
ref = grads.clone()
ddist.average_gradients(grads)
t = torch.ones(8, device=dev); dist.all_reduce(t)
torch.cuda.synchronize()
print(json.dumps({"gather_equal": bool(torch.equal(full, local_traj)), "avg_equal": bool(torch.equal(grads, ref)),
                  "allreduce": float(t.sum().item()), "backend": dist.get_backend()}))
dist.destroy_process_group()
'''
    env = _child_env(RANK=0, WORLD_SIZE=1, LOCAL_RANK=0, MASTER_ADDR="127.0.0.1", MASTER_PORT=29617)
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out == {"gather_equal": True, "avg_equal": True, "allreduce": 8.0, "backend": "nccl"}


def _tiny_project(tmp_path, nj=3, T=11, hw=(96, 128)):
    import yaml
    from deepgraphpose_amd import weights_io
    from deepgraphpose_amd.synthetic import make_frames, make_weights
    parts = ["a", "b", "c"][:nj]
    proj = tmp_path / "proj"
    train = proj / "dlc-models" / "iteration-0" / "DemoOct2-trainset95shuffle1" / "train"
    train.mkdir(parents=True)
    (proj / "config.yaml").write_text(yaml.safe_dump(dict(Task="Demo", date="Oct2", iteration=0, TrainingFraction=[0.95],
                                                          bodyparts=parts, skeleton=[], project_path=str(proj))))
    (train / "pose_cfg.yaml").write_text(yaml.safe_dump(dict(num_joints=nj, all_joints_names=parts, net_type="resnet_50")))
    wts = make_weights(50, nj, False, seed=9, head_std=0.05)
    snap = weights_io.save_weights(str(train / "snapshot-step2-final--0"), wts)          # (a TF V2 bundle: returns the prefix)
    assert snap.endswith("snapshot-step2-final--0")
    frames = make_frames(T, hw[0], hw[1], nj, seed=5)
    np.save(tmp_path / "clip.npy", frames)
    return proj, snap, frames, wts


def test_estimate_pose_under_torchrun_env_shards_and_gathers(lib_built, tmp_path):
    """The product entry point under the launcher's environment (one rank): it joins the RCCL group itself, takes
    shard_range(T, 0, 1), gathers with all_gather_into_tensor and writes the csv -- and matches the single-process call."""
    proj, snap, frames, wts = _tiny_project(tmp_path)
    code = r'''
import json, sys, numpy as np
from deepgraphpose_amd.models import eval as E
out = E.estimate_pose(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], shuffle=1, batch_size=4)
import torch.distributed as dist
assert dist.is_initialized() and dist.get_backend() == "nccl"
np.savez(sys.argv[5], **out)
dist.destroy_process_group()
'''
    env = _child_env(RANK=0, WORLD_SIZE=1, LOCAL_RANK=0, MASTER_ADDR="127.0.0.1", MASTER_PORT=29618)
    # WORLD_SIZE=1 alone does not trigger the init inside estimate_pose; the launcher contract (torchrun) is RANK + WORLD_SIZE
    # with the group created by init_from_env -> force it the way bench.py does
    code = "from deepgraphpose_amd import dist as d; d.init_from_env('nccl')\n" + code
    r = subprocess.run([sys.executable, "-c", code, str(proj / "config.yaml"), snap, str(tmp_path / "clip.npy"),
                        str(tmp_path / "pred_dist"), str(tmp_path / "out.npz")], env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(tmp_path / "out.npz")
    from deepgraphpose_amd.models import eval as E
    ref = E.estimate_pose(str(proj / "config.yaml"), snap, str(tmp_path / "clip.npy"), str(tmp_path / "pred_single"), shuffle=1,
                          batch_size=4)
    for k in ("x", "y", "likelihoods"):
        assert np.array_equal(got[k], ref[k]), k
    assert os.path.isfile(tmp_path / "pred_dist" / "clip_labeled.csv")


@pytest.mark.parametrize("world", [2, 8])
def test_estimate_pose_two_ranks_on_one_gpu_equal_single_process(lib_built, tmp_path, world):
    """The sharded product path with W = 2 and W = 8 for real: W processes (all on cuda:0, control plane on gloo via DGP_DIST_BACKEND
    because RCCL refuses two ranks on one device) run estimate_pose on the same video.  Every shard but the first starts in the middle
    (frame_at seek), all calibrate on the video's first batch, the gather reassembles [T, nj], rank 0 alone exports -- and the result
    equals the single-process run bit for bit.  T = 23: with eight ranks the shards are 3, 3, ..., 2 frames (T % 8 != 0, batch_size 4)."""
    proj, snap, frames, wts = _tiny_project(tmp_path, T=23)
    code = r'''
import json, os, sys, numpy as np
from deepgraphpose_amd.models import eval as E
out = E.estimate_pose(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], shuffle=1, batch_size=4)
import torch.distributed as dist
assert dist.is_initialized() and dist.get_world_size() == int(os.environ["WORLD_SIZE"]) and dist.get_backend() == "gloo"
np.savez(sys.argv[5] + os.environ["RANK"] + ".npz", **out)
dist.barrier()
dist.destroy_process_group()
'''
    procs = []
    for rank in range(world):
        env = _child_env(RANK=rank, WORLD_SIZE=world, LOCAL_RANK=0, MASTER_ADDR="127.0.0.1", MASTER_PORT=29631 + world, DGP_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, "-c", code, str(proj / "config.yaml"), snap, str(tmp_path / "clip.npy"),
                                       str(tmp_path / "pred_w2"), str(tmp_path / "out_rank")], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for pr in procs:
        try:
            outs.append(pr.communicate(timeout=900))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for pr, (so, se) in zip(procs, outs):
        assert pr.returncode == 0, se[-2000:]
    from deepgraphpose_amd.models import eval as E
    ref = E.estimate_pose(str(proj / "config.yaml"), snap, str(tmp_path / "clip.npy"), str(tmp_path / "pred_w1"), shuffle=1,
                          batch_size=4)
    for rank in range(world):
        got = np.load(str(tmp_path / "out_rank") + "%d.npz" % rank)
        for k in ("x", "y", "likelihoods"):
            assert got[k].shape == (23, 3) and np.array_equal(got[k], ref[k]), (rank, k)
    a = open(tmp_path / "pred_w2" / "clip_labeled.csv").read()
    b = open(tmp_path / "pred_w1" / "clip_labeled.csv").read()
    assert a == b


def test_bench_two_ranks_on_one_gpu(lib_built):
    """bench.py's N > 1 path (shard_range of one seeded stream, gather inside the timed region, max-over-ranks time, shard check
    against rank 0's ring) with two processes on cuda:0 and the control plane on gloo (DGP_DIST_BACKEND): one JSON line from rank 0
    only, n_gpus 2, every batch of both shards identical to the single-rank result."""
    procs = []
    for rank in (0, 1):
        env = _child_env(RANK=rank, WORLD_SIZE=2, LOCAL_RANK=0, MASTER_ADDR="127.0.0.1", MASTER_PORT=29633, DGP_DIST_BACKEND="gloo",
                         DGP_BENCH_VISIBLE_GPUS=1)          # (two ranks on ONE device, knowingly)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1",
                                       "--batch", "8", "--no-cpu-baseline", "--sustain-seconds", "0.3", "--prewarm-seconds", "0.2"],
                                      env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for pr in procs:
        try:
            outs.append(pr.communicate(timeout=900))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for pr, (so, se) in zip(procs, outs):
        assert pr.returncode == 0, se[-2000:]
    lines0 = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    lines1 = [l for l in outs[1][0].splitlines() if l.startswith("{")]
    assert len(lines0) == 1 and not lines1
    d = json.loads(lines0[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["scaling"] == "weak" and d["value"] > 0
    assert d["shard_check"]["batches_compared"] == 6 and d["shard_check"]["bit_identical_to_rank0"]
    assert abs(d["value"] - 2 * 5 * 8 / (d["ms_per_step"] * 5 / 1e3)) <= 1e-3 * d["value"]


def _run_bench(args, **env_extra):
    env = _child_env(**env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert cp.returncode == 0, (cp.stdout[-1500:], cp.stderr[-1500:])
    lines = [l for l in cp.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_through_its_own_spawner(lib_built):
    """`python bench.py --gpus N` with no launcher around it spawns its N workers itself (the parent never touches the GPU) and relays
    rank 0's line.  Here N = 1 through the spawner (DGP_BENCH_FORCE_SPAWN=1): RCCL group of one rank, the strict-fp32 child run included."""
    d = _run_bench(["--gpus", "1", "--steps", "6", "--warmup", "1", "--batch", "8", "--no-cpu-baseline", "--sustain-seconds", "0.3",
                    "--prewarm-seconds", "0.2", "--train-steps", "3", "--no-r101", "--host-frames", "128"], DGP_BENCH_FORCE_SPAWN=1)
    hp = d["host_pipeline"]                                 # the PCIe-inclusive estimate_pose leg, also a fresh child
    assert "error" not in hp and hp["frames"] == 128 and hp["frames_per_s"] > 0 and hp["host_seconds"]["drain_s"] >= 0
    ts = d["train_step"]                                    # BASELINE configs[3], timed by a fresh child of the same run
    assert "error" not in ts and 0 < ts["ms_per_step"] < 200 and ts["frames_per_step"] == 11 and 0 < ts["frac"] < 1 and np.isfinite(ts["loss"]["total_loss"])
    assert "spawned 1 worker" in d["launcher"] and d["n_gpus"] == 1 and d["steps"] == 6 and d["value"] > 0
    assert d["roofline"]["frac"] > 0 and d["shard_check"]["indices_identical"]
    sf = d["strict_f32"]
    assert "error" not in sf and sf["frames_per_s"] > 0 and 0 < sf["frac_of_fp32_mfma_peak"] < 1
    assert sf["frames_per_s"] < d["value"]                  # the IEEE-fp32 tier is the slower one
    kern = d["roofline"]["kernels"]
    assert any(k.startswith("chain_") or k.startswith("unit_") for k in kern)       # the fused bottleneck launches are in the table
    assert all(v["bound"] in ("hbm", "mfma") for v in kern.values())


def test_bench_under_torch_distributed_run_exactly_as_the_driver_launches_it(lib_built):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 --steps K
    --warmup W` -- the driver's command line for N > 1 -- on the one GPU of this box: the launcher numbers LOCAL_RANK 0 and 1, the explicit
    opt-in DGP_BENCH_VISIBLE_GPUS=1 folds both onto cuda:0 (control plane on gloo: RCCL refuses two ranks on one device).  ONE JSON line,
    from rank 0, with the collective object a reader verifies an N-rank line by."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = _child_env(DGP_DIST_BACKEND="gloo", DGP_BENCH_VISIBLE_GPUS=1)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--batch", "8", "--no-cpu-baseline",
           "--sustain-seconds", "0.2", "--prewarm-seconds", "0.2"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    lines = [q for q in r.stdout.splitlines() if q.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["shard_check"]["indices_identical"] and d["shard_check"]["bit_identical_to_rank0"]
    c = d["collective"]
    assert c["backend"] == "gloo" and c["world_size"] == 2 and c["initialized"] and len(c["ranks"]) == 2
    assert sorted(q["rank"] for q in c["ranks"]) == [0, 1] and c["distinct_devices"] == 1
    assert "launcher" not in d and "strict_f32" not in d        # (a launcher started the workers; the child legs run at N = 1 only)


def test_bench_strong_scaling_two_ranks_through_the_spawner(lib_built):
    """--scaling strong: ONE fixed stream (--total-batches) split over the ranks, through bench.py's own spawner; two ranks on the one
    GPU of the box (gloo control plane), result identical batch by batch to rank 0's ring."""
    d = _run_bench(["--gpus", "2", "--scaling", "strong", "--total-batches", "8", "--warmup", "1", "--batch", "8", "--no-cpu-baseline",
                    "--sustain-seconds", "0.2", "--prewarm-seconds", "0.2"], DGP_DIST_BACKEND="gloo", DGP_BENCH_VISIBLE_GPUS=1)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 4 and d["config"]["total_frames"] == 64
    assert d["shard_check"]["batches_compared"] == 4 and d["shard_check"]["indices_identical"]
    assert "spawned 2 worker" in d["launcher"] and "strict_f32" not in d


def test_bench_strong_scaling_eight_ranks_through_the_spawner(lib_built):
    """The command the README quotes for an 8-GPU node -- `python bench.py --gpus 8 --scaling strong` -- through bench.py's own spawner with all
    eight ranks on the one GPU of this box (gloo control plane, DGP_BENCH_VISIBLE_GPUS=1): eight shards of one fixed stream, one gather,
    every batch identical to rank 0's ring, per-rank elapsed times reported."""
    d = _run_bench(["--gpus", "8", "--scaling", "strong", "--total-batches", "8", "--warmup", "1", "--batch", "4", "--no-cpu-baseline",
                    "--sustain-seconds", "0.1", "--prewarm-seconds", "0.1", "--streams", "1"], DGP_DIST_BACKEND="gloo", DGP_BENCH_VISIBLE_GPUS=1)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["steps"] == 1 and d["config"]["total_frames"] == 32
    assert "spawned 8 worker" in d["launcher"]
    assert d["shard_check"]["batches_compared"] == 4 and d["shard_check"]["indices_identical"]       # ranks 4-7 against ranks 0-3
    assert d["frames_per_s"] == d["value"] > 0
    re = d["rank_elapsed_s"]
    assert re["max"] >= re["min"] > 0 and re["imbalance"] >= 0
    # the line verifies itself: what torch.distributed saw, every rank's device and rate, the one collective
    c = d["collective"]
    assert c["backend"] == "gloo" and c["world_size"] == 8 and c["initialized"] and c["all_gather_bytes"] == 8 * 4 * 4 * 5 * 4
    assert c["all_gather_ms"]["max"] >= c["all_gather_ms"]["rank0"] > 0
    assert [q["rank"] for q in c["ranks"]] == list(range(8))
    assert all(q["frames_per_s"] > 0 and q["elapsed_s"] > 0 and q["index"] == 0 and len(q["pci"]) == 12 and "numa_node" in q for q in c["ranks"])
    assert c["distinct_devices"] == 1                          # eight ranks on ONE device here: a real 8-GPU line reads 8


def test_bench_refuses_more_ranks_than_devices_unless_told(lib_built):
    """`--gpus 2` on a one-GPU box is an error (several ranks on one device by accident would print a line that is not a 2-GPU line); the
    explicit opt-in is DGP_BENCH_VISIBLE_GPUS."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than two devices")
    env = _child_env()
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "DGP_BENCH_VISIBLE_GPUS"):
        env.pop(k, None)
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"], env=env, cwd=ROOT, capture_output=True, text=True,
                        timeout=300)
    assert cp.returncode != 0 and "device(s) visible" in (cp.stderr + cp.stdout)


def test_shard_ranges_reassemble_bit_exactly(lib_built):
    """What N ranks would compute: the frames of each shard_range(T, r, W) inferred separately (own batches) and concatenated
    equal the one-process trajectory on the integer indices bit for bit and on the coordinates within 1e-3 px."""
    from deepgraphpose_amd import dist as ddist
    from deepgraphpose_amd.engine import DGPNet
    from deepgraphpose_amd.synthetic import make_frames, make_weights
    nj, T = 4, 21
    net = DGPNet(50, nj, 64, 96, max_batch=8)
    net.load_weights(make_weights(50, nj, False, seed=3, head_std=0.05))
    ft = torch.from_numpy(make_frames(T, 64, 96, nj, seed=8)).cuda()

    def run(lo, hi):
        traj = torch.zeros((hi - lo, nj, 5), dtype=torch.float32, device="cuda")
        for s in range(lo, hi, 8):
            e = min(s + 8, hi)
            net.infer_packed(ft[s:e].contiguous(), traj[s - lo:e - lo])
        return traj
    one = run(0, T)
    for W in (2, 4, 8):
        parts = [run(*ddist.shard_range(T, r, W)) for r in range(W)]
        cat = torch.cat(parts, 0)
        m1, c1, i1 = ddist.unpack_keypoints(one)
        m2, c2, i2 = ddist.unpack_keypoints(cat)
        assert torch.equal(i1, i2)
        assert float((m1 - m2).abs().max()) * 8.0 < PX_TOL


def test_run_dgp_demo_test_mode_end_to_end(lib_built, tmp_path):
    """`demo/run_dgp_demo.py --dlcpath <synthetic project> --dlcsnapshot snapshot-step0-final--0 --test` as a subprocess: steps
    1-3 run on the GPU, the step snapshots and videos_pred/<video>_labeled.csv (DLC 3-row header, [T, 3 nj]) appear."""
    from _project import make_project
    proj, frames, wts = make_project(tmp_path)
    env = _child_env()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "demo", "run_dgp_demo.py"), "--dlcpath", proj, "--dlcsnapshot",
                        "snapshot-step0-final--0", "--batch_size", "4", "--test"], env=env, cwd=str(tmp_path), capture_output=True,
                       text=True, timeout=1800)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    train = os.path.join(proj, "dlc-models", "iteration-0", "DemoOct2-trainset95shuffle1", "train")
    for step in (1, 2):
        assert os.path.isfile(os.path.join(train, "snapshot-step%d-final--0.index" % step))          # TF V2 bundles, like the reference's Saver
        assert os.path.isfile(os.path.join(train, "snapshot-step%d-final--0.data-00000-of-00001" % step))
    csv = os.path.join(proj, "videos_pred", "clip_labeled.csv")
    assert os.path.isfile(csv)
    rows = open(csv).read().strip().split("\n")
    assert rows[0].startswith("scorer") and rows[1].startswith("bodyparts") and rows[2].startswith("coords")
    assert len(rows) == 3 + frames.shape[0] and len(rows[3].split(",")) == 1 + 3 * 3
    assert "Running DGP with labeled frames only" in r.stdout and "Predict with DGP" in r.stdout


def test_run_dgp_demo_at_the_reaching_frame_size(lib_built, tmp_path):
    """BASELINE configs[0] at its real shape: a synthetic project with the Reaching demo's geometry -- 832 x 747 frames, 5 bodyparts
    (config.yaml:6-11) -- through `run_dgp_demo.py --test`: the fit steps train at 747 x 832 (scoremaps 94 x 104), estimate_pose
    writes [T, 3 x 5] labels."""
    from _project import make_project
    proj, frames, wts = make_project(tmp_path, nj=5, n_frames=24, hw=(747, 832), labeled=(2, 9, 15, 21))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "demo", "run_dgp_demo.py"), "--dlcpath", proj, "--dlcsnapshot",
                        "snapshot-step0-final--0", "--batch_size", "4", "--test"], env=_child_env(), cwd=str(tmp_path), capture_output=True,
                       text=True, timeout=2400)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    csv = os.path.join(proj, "videos_pred", "clip_labeled.csv")
    rows = open(csv).read().strip().split("\n")
    assert len(rows) == 3 + frames.shape[0] and len(rows[3].split(",")) == 1 + 3 * 5
    vals = np.array([[float(v) for v in row.split(",")[1:]] for row in rows[3:]])
    assert np.isfinite(vals).all()
    assert (vals[:, 0::3] >= 0).all() and (vals[:, 0::3] <= 832).all() and (vals[:, 1::3] >= 0).all() and (vals[:, 1::3] <= 747).all()


def test_strict_two_ranks_stay_bit_identical_after_a_late_overflow(lib_built, tmp_path):
    """DGP_EVAL_STRICT=1 in a sharded run: the first chunk of EACH shard is flat (the scales calibrated on the video's first batch are far too
    small, and nothing overflows in round 0), the real frames begin in every rank's second chunk, so both ranks overflow LATE.  The ranks
    decide together, widen together, re-run that chunk, and -- strict -- compute the whole video again on the final scales: the gathered
    result equals, bit for bit, a single-process run whose one chunk holds the whole video (everything on the wide scales)."""
    from deepgraphpose_amd.models import eval as E
    from deepgraphpose_amd.synthetic import make_frames
    proj, snap, _, wts = _tiny_project(tmp_path)
    frames = make_frames(40, 96, 128, 3, seed=77)                                      # T = 40: shards of 20 frames, chunks of 8
    for lo in (0, 20):                                                                 # the FIRST chunk of each shard is flat
        frames[lo:lo + 8, ..., 0], frames[lo:lo + 8, ..., 1], frames[lo:lo + 8, ..., 2] = 124, 117, 104
    np.save(tmp_path / "mixed.npy", frames)
    code = r'''
import os, sys, numpy as np
from deepgraphpose_amd.models import eval as E
out = E.estimate_pose(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], shuffle=1, batch_size=4)
np.savez(sys.argv[5] + os.environ["RANK"] + ".npz", strict_passes=E.RUN_STATS["strict_passes"], **out)
import torch.distributed as dist
dist.barrier(); dist.destroy_process_group()
'''
    procs = []
    for rank in range(2):
        env = _child_env(RANK=rank, WORLD_SIZE=2, LOCAL_RANK=0, MASTER_ADDR="127.0.0.1", MASTER_PORT=29671, DGP_DIST_BACKEND="gloo",
                         DGP_EVAL_STRICT=1, DGP_EVAL_CHUNK_BATCHES=2)
        procs.append(subprocess.Popen([sys.executable, "-c", code, str(proj / "config.yaml"), snap, str(tmp_path / "mixed.npy"),
                                       str(tmp_path / "pred_strict2"), str(tmp_path / "strict_rank")], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for pr in procs:
        so, se = pr.communicate(timeout=900)
        assert pr.returncode == 0, se[-2000:]
    os.environ["DGP_EVAL_CHUNK_BATCHES"] = "64"
    try:
        whole = E.estimate_pose(str(proj / "config.yaml"), snap, str(tmp_path / "mixed.npy"), str(tmp_path / "pred_whole2"), shuffle=1, batch_size=4)
    finally:
        del os.environ["DGP_EVAL_CHUNK_BATCHES"]
    for rank in range(2):
        got = np.load(str(tmp_path / "strict_rank") + "%d.npz" % rank)
        assert int(got["strict_passes"]) == 1
        for k in ("x", "y", "likelihoods"):
            assert got[k].shape == (40, 3) and np.array_equal(got[k], whole[k]), (rank, k)


def test_evaluate_dgp_soft_argmax_locref_readout(lib_built, tmp_path):
    """evaluate_dgp(loc_ref=True, loc_ref_calc='dgp') (eval.py:752-786): runs on the synthetic project and its per-frame read-out
    equals the restated numpy of the reference on the oracle's maps."""
    from _project import make_project
    from deepgraphpose_amd.models.eval import evaluate_dgp, soft_argmax_locref_pose
    from deepgraphpose_amd.models.fitdgp_util import get_snapshot_path
    from oracle import dgp_oracle as O
    proj, frames, wts = make_project(tmp_path)
    snap, cfg_path = get_snapshot_path("snapshot-step0-final--0", proj, shuffle=1)
    rmse = evaluate_dgp(str(cfg_path), snap, shuffle=1, loc_ref=True, loc_ref_calc="dgp")
    assert rmse.shape == (4, 3) and np.isfinite(rmse.values[~np.isnan(rmse.values)]).all()
    # closed form on a one-hot softmax map: position of the hot cell + its locref offset, (x, y) order
    H, W, nj = 6, 7, 2
    st = np.zeros((H, W, nj)); st[2, 5, 0] = 1.0; st[4, 1, 1] = 1.0
    lr = np.zeros((H, W, 2 * nj)); lr[2, 5, 0:2] = [0.5, -0.25]; lr[4, 1, 2:4] = [1.0, 2.0]
    pose = soft_argmax_locref_pose(lr, st, 8.0, 7.2801)
    np.testing.assert_allclose(pose[0], [5 * 8 + 4 - 0.25 * 7.2801, 2 * 8 + 4 + 0.5 * 7.2801, 1.0])
    np.testing.assert_allclose(pose[1], [1 * 8 + 4 + 2.0 * 7.2801, 4 * 8 + 4 + 1.0 * 7.2801, 1.0])


def test_estimate_pose_reruns_only_the_chunk_that_overflowed(lib_built, tmp_path, monkeypatch):
    """A video whose first frames are flat (tiny activations: the H2 scales calibrated on the first batch are far too small) followed by
    real frames.  The range check of the chunk the real frames land in reports the overflow; estimate_pose re-calibrates and
    re-runs THAT chunk from the frames still resident in HBM -- one chunk re-run, not the video -- and every frame matches the oracle."""
    from oracle import dgp_oracle as O
    from deepgraphpose_amd.models import eval as E
    from deepgraphpose_amd.synthetic import make_frames
    proj, snap, _, wts = _tiny_project(tmp_path)
    T_flat, T_real = 16, 24
    flat = np.zeros((T_flat, 64, 96, 3), np.uint8)
    flat[..., 0], flat[..., 1], flat[..., 2] = 124, 117, 104
    frames = np.concatenate([flat, make_frames(T_real, 64, 96, 3, seed=77)], 0)
    np.save(tmp_path / "mixed.npy", frames)
    monkeypatch.setenv("DGP_EVAL_CHUNK_BATCHES", "2")                 # chunks of 2 batches of 4 frames: the real frames start in chunk 2
    out = E.estimate_pose(str(proj / "config.yaml"), snap, str(tmp_path / "mixed.npy"), str(tmp_path / "pred_mixed"), shuffle=1,
                          batch_size=4)
    assert E.RUN_STATS["chunks"] == 5 and E.RUN_STATS["chunk_reruns"] == 1 and E.RUN_STATS["strict_passes"] == 0, E.RUN_STATS
    ref = O.infer(frames, wts, 50, 8.0, 1.0, 1)
    assert np.abs(out["x"] - ref["x"]).max() < 1e-3 and np.abs(out["y"] - ref["y"]).max() < 1e-3
    # DGP_EVAL_STRICT=1: chunks 0-1 were computed on the narrower scales, so the video is computed again on the final ones -- and is
    # then bit-identical to a run whose ONE chunk holds the whole video (overflow in chunk 0: everything re-run on the wide scales)
    monkeypatch.setenv("DGP_EVAL_STRICT", "1")
    strict = E.estimate_pose(str(proj / "config.yaml"), snap, str(tmp_path / "mixed.npy"), str(tmp_path / "pred_strict"), shuffle=1,
                             batch_size=4)
    assert E.RUN_STATS["strict_passes"] == 1 and E.RUN_STATS["chunk_reruns"] == 1, E.RUN_STATS
    monkeypatch.setenv("DGP_EVAL_CHUNK_BATCHES", "64")
    whole = E.estimate_pose(str(proj / "config.yaml"), snap, str(tmp_path / "mixed.npy"), str(tmp_path / "pred_whole"), shuffle=1,
                            batch_size=4)
    assert E.RUN_STATS["chunks"] == 1 and E.RUN_STATS["chunk_reruns"] == 1 and E.RUN_STATS["strict_passes"] == 0, E.RUN_STATS
    for k in ("x", "y", "likelihoods"):
        assert np.array_equal(strict[k], whole[k]), k
    assert np.abs(strict["x"] - ref["x"]).max() < 1e-3 and np.abs(strict["y"] - ref["y"]).max() < 1e-3


def test_estimate_pose_edge_cases_vs_oracle(lib_built, tmp_path):
    """estimate_pose options and ragged inputs (eval.py:217-360): a single frame with a larger batch size; odd frame sizes; T not a
    multiple of the batch; `new_size` (PIL resize, coordinates scaled back to the original frame) and `crop_size` -- each against
    the CPU oracle run on the frames as the reference would have prepared them."""
    from PIL import Image
    from oracle import dgp_oracle as O
    from deepgraphpose_amd.models import eval as E
    from deepgraphpose_amd.synthetic import make_frames
    proj, snap, _, wts = _tiny_project(tmp_path)
    cfg = str(proj / "config.yaml")

    def run(frames, tag, **kw):
        np.save(tmp_path / (tag + ".npy"), frames)
        return E.estimate_pose(cfg, snap, str(tmp_path / (tag + ".npy")), str(tmp_path / ("pred_" + tag)), shuffle=1, **kw)

    def check(out, ref, sx=1.0, sy=1.0):
        assert out["x"].shape == ref["x"].shape
        assert np.abs(out["x"] - ref["x"] * sx).max() < PX_TOL * max(sx, 1.0)
        assert np.abs(out["y"] - ref["y"] * sy).max() < PX_TOL * max(sy, 1.0)
        assert np.abs(out["likelihoods"] - ref["likelihoods"]).max() < 1e-4

    one = make_frames(1, 96, 128, 3, seed=41)
    check(run(one, "one", batch_size=8), O.infer(one, wts, 50, 8.0, 1.0, 1))
    odd = make_frames(7, 75, 101, 3, seed=42)                      # odd sizes, T = 7 over batches of 3 (3 + 3 + 1)
    check(run(odd, "odd", batch_size=3), O.infer(odd, wts, 50, 8.0, 1.0, 1))
    big = make_frames(3, 120, 160, 3, seed=43)
    small = np.stack([np.asarray(Image.fromarray(f).resize(size=(96, 72))) for f in big])       # new_size = (rows 72, cols 96)
    out = run(big, "resized", batch_size=2, new_size=(72, 96))
    check(out, O.infer(small, wts, 50, 8.0, 1.0, 1), sx=160 / 96, sy=120 / 72)
    crop = (8, 16, 136, 112)                                       # PIL box (left, upper, right, lower) -> 96 x 128 frames
    cropped = np.stack([np.asarray(Image.fromarray(f).crop(crop)) for f in big])
    check(run(big, "cropped", batch_size=2, crop_size=crop), O.infer(cropped, wts, 50, 8.0, 1.0, 1))


def test_pipeline_two_engines_equal_single_engine(lib_built):
    """engine.DGPPipeline (two engines on two HIP streams, batches dealt in turn): every batch's packed records equal what ONE engine
    calibrated on the same first batch computes, bit for bit, whichever engine they landed on; ragged last batch; a re-calibration
    (widen) keeps the engines on identical scales."""
    from deepgraphpose_amd.engine import DGPNet, DGPPipeline
    from deepgraphpose_amd.synthetic import make_frames, make_weights
    nj, T, B = 4, 45, 8
    wts = make_weights(50, nj, False, seed=13, head_std=0.05)
    ft = torch.from_numpy(make_frames(T, 64, 96, nj, seed=14)).cuda()
    one = DGPNet(50, nj, 64, 96, max_batch=B)
    one.load_weights(wts)
    ref = torch.zeros((T, nj, 5), dtype=torch.float32, device="cuda")
    for s in range(0, T, B):
        one.infer_packed(ft[s:s + B].contiguous(), ref[s:s + B])
    assert one.range_status() == (False, 1)
    pipe = DGPPipeline(50, nj, 64, 96, max_batch=B, n_streams=2)
    pipe.load_weights(wts)
    got = torch.zeros_like(ref)
    events = [pipe.submit(ft[s:s + B].contiguous(), got[s:s + B]) for s in range(0, T, B)]      # 6 batches, the last of 5 frames
    pipe.join()
    torch.cuda.synchronize()
    assert all(e.query() for e in events)
    assert pipe.range_status() == (False, 1) and [n.range_status()[1] for n in pipe.nets] == [1, 1]
    assert torch.equal(got, ref)
    pipe.widen()                                            # e.g. another rank overflowed: all engines re-calibrate together
    got2 = torch.zeros_like(ref)
    for s in range(0, T, B):
        pipe.submit(ft[s:s + B].contiguous(), got2[s:s + B])
    pipe.join()
    torch.cuda.synchronize()
    assert [n.range_status() for n in pipe.nets] == [(False, 2), (False, 2)]
    m1, c1, i1 = [t.cpu().numpy() for t in __import__("deepgraphpose_amd.dist", fromlist=["x"]).unpack_keypoints(got2)]
    m0, c0, i0 = [t.cpu().numpy() for t in __import__("deepgraphpose_amd.dist", fromlist=["x"]).unpack_keypoints(ref)]
    assert np.array_equal(i1, i0) and np.abs(m1 - m0).max() * 8.0 < PX_TOL


def test_pipeline_resyncs_engines_after_a_private_recalibration(lib_built):
    """EvalSession shares engine 0 with the pipeline: a synchronous forward on it that overflows re-calibrates THAT engine alone
    (new scales, +3 bits).  The next submit notices (scale_epoch), gives every engine the widest headroom and re-calibrates all of them
    on one batch -- which engine a batch lands on again does not change a bit of its result."""
    from deepgraphpose_amd.engine import DGPPipeline
    from deepgraphpose_amd.synthetic import make_frames, make_weights
    nj, B = 3, 4
    wts = make_weights(50, nj, False, seed=33, head_std=0.05)
    ft = torch.from_numpy(make_frames(2 * B, 64, 96, nj, seed=34)).cuda()
    pipe = DGPPipeline(50, nj, 64, 96, max_batch=B, n_streams=2)
    pipe.load_weights(wts)
    out = torch.zeros((4, B, nj, 5), device="cuda")
    pipe.submit(ft[:B].contiguous(), out[0]); pipe.submit(ft[B:].contiguous(), out[1])
    pipe.join(); torch.cuda.synchronize()
    assert pipe.range_status() == (False, 1)
    pipe.nets[0].widen()                                   # what DGPNet.infer(check_range=True) does to the shared engine on an overflow
    pipe.nets[0].infer(ft[B:].contiguous())                # ... and it re-calibrates on ITS batch: engine 0 now has other scales
    assert [n.widen_count for n in pipe.nets] == [1, 0]
    pipe.submit(ft[:B].contiguous(), out[2]); pipe.submit(ft[:B].contiguous(), out[3])       # the same batch on both engines
    pipe.join(); torch.cuda.synchronize()
    assert [n.widen_count for n in pipe.nets] == [1, 1]
    assert [n.range_status()[1] for n in pipe.nets] == [3, 2]      # engine 0: load + private + resync; engine 1: load + resync
    assert torch.equal(out[2], out[3])


def test_pipeline_overflow_widens_every_engine(lib_built):
    """DGPPipeline.range_status: scales calibrated on a near-empty batch overflow on real frames on whichever engine gets them; the
    status call reports it, gives EVERY engine the wider headroom, and the next submit re-calibrates all of them on its batch -- after
    which both engines again produce the single-engine bits."""
    from deepgraphpose_amd.engine import DGPNet, DGPPipeline
    from deepgraphpose_amd.synthetic import make_frames, make_weights
    nj, B = 3, 4
    wts = make_weights(50, nj, False, seed=31, head_std=0.05)
    flat = np.zeros((B, 64, 96, 3), dtype=np.uint8)
    flat[..., 0], flat[..., 1], flat[..., 2] = 124, 117, 104
    real = torch.from_numpy(make_frames(3 * B, 64, 96, nj, seed=32)).cuda()
    pipe = DGPPipeline(50, nj, 64, 96, max_batch=B, n_streams=2)
    pipe.load_weights(wts)
    out = torch.zeros((3 * B, nj, 5), device="cuda")
    pipe.submit(torch.from_numpy(flat).cuda(), out[:B])            # calibrates both engines on ~zero activations
    pipe.join(); torch.cuda.synchronize()
    assert pipe.range_status() == (False, 1)
    pipe.submit(real[:B].contiguous(), out[:B])                    # engine 1 overflows
    pipe.join(); torch.cuda.synchronize()
    ov, _ = pipe.range_status()
    assert ov and not pipe._calibrated
    for s in range(0, 3 * B, B):                                   # re-run: the first submit re-calibrates BOTH engines on its batch
        pipe.submit(real[s:s + B].contiguous(), out[s:s + B])
    pipe.join(); torch.cuda.synchronize()
    assert pipe.range_status()[0] is False and [n.range_status()[1] for n in pipe.nets] == [2, 2]
    one = DGPNet(50, nj, 64, 96, max_batch=B)
    one.load_weights(wts)
    one.widen()                                                    # same headroom as the widened engines, calibrated on the same batch
    ref = torch.zeros_like(out)
    for s in range(0, 3 * B, B):
        one.infer_packed(real[s:s + B].contiguous(), ref[s:s + B])
    torch.cuda.synchronize()
    assert torch.equal(out, ref)


def test_frame_uploader_matches_plain_upload(lib_built):
    """fitdgp._FrameUploader (pinned staging + own stream, used from the fit drivers' prefetch thread): the device frames equal a plain
    upload, float images are rounded like _frames_to_device does, buffers are reused across calls of different sizes."""
    from deepgraphpose_amd.models.fitdgp import _FrameUploader
    up = _FrameUploader(0, slots=2)
    rng = np.random.RandomState(0)
    for shape in [(3, 64, 96, 3), (5, 64, 96, 3), (2, 32, 48, 3), (5, 64, 96, 3)]:
        img = rng.randint(0, 256, shape).astype(np.uint8)
        dev = up(img)
        torch.cuda.current_stream().wait_event(dev._dgp_ready)
        assert dev.dtype == torch.uint8 and tuple(dev.shape) == shape and np.array_equal(dev.cpu().numpy(), img)
    f = rng.uniform(0, 255, (2, 16, 16, 3))
    dev = up(f)
    torch.cuda.current_stream().wait_event(dev._dgp_ready)
    assert np.array_equal(dev.cpu().numpy(), np.clip(np.rint(f), 0, 255).astype(np.uint8))


def test_estimate_pose_keeps_its_engines_between_videos_without_changing_a_bit(lib_built, tmp_path, monkeypatch):
    """Round 6: setup_dgp_eval_graph keeps ONE session for the next call on the same snapshot (a project's videos are labelled one after the
    other with one model; the reference restored the graph for every video).  What must not change is the result: video B through the
    kept engines -- after video A whose flat first frames forced a re-calibration with WIDENED headroom -- equals, bit for bit, video B
    through a fresh session (DGP_EVAL_SESSION_CACHE=0): dgp_net_reset_scales puts a kept engine back to its post-load state.  A changed
    snapshot file (other size / mtime) or another tier is another session."""
    from deepgraphpose_amd.models import eval as E
    from deepgraphpose_amd import weights_io
    from deepgraphpose_amd.synthetic import make_frames, make_weights
    proj, snap, frames_b, wts = _tiny_project(tmp_path, T=21, hw=(64, 96))
    flat = np.zeros((8, 64, 96, 3), np.uint8)
    flat[..., 0], flat[..., 1], flat[..., 2] = 124, 117, 104
    frames_a = np.concatenate([flat, make_frames(12, 64, 96, 3, seed=77)], 0)
    cfgp = str(proj / "config.yaml")
    E.clear_session_cache()
    monkeypatch.setenv("DGP_EVAL_SESSION_CACHE", "0")
    fresh = E.estimate_pose(cfgp, snap, frames_b, str(tmp_path / "o0"), save_pose=False, batch_size=4)
    assert not E._SESSION_CACHE
    monkeypatch.setenv("DGP_EVAL_SESSION_CACHE", "1")
    E.estimate_pose(cfgp, snap, frames_a, str(tmp_path / "o1"), save_pose=False, batch_size=4)         # (overflows: the engines widen)
    assert E.RUN_STATS["chunk_reruns"] >= 1
    sess_a = E._SESSION_CACHE["sess"]
    kept = E.estimate_pose(cfgp, snap, frames_b, str(tmp_path / "o2"), save_pose=False, batch_size=4)
    assert E._SESSION_CACHE["sess"] is sess_a and E.RUN_STATS["chunk_reruns"] == 0
    for k in ("x", "y", "likelihoods"):
        assert np.array_equal(kept[k], fresh[k]), k
    f16 = E.estimate_pose(cfgp, snap, frames_b, str(tmp_path / "o3"), save_pose=False, batch_size=4, tier="f16")
    assert E._SESSION_CACHE["sess"] is not sess_a and not np.array_equal(f16["x"], fresh["x"])
    sess_f = E._SESSION_CACHE["sess"]
    w2 = dict(wts)
    w2["pose/part_pred/block4/biases"] = wts["pose/part_pred/block4/biases"] + 0.25
    weights_io.save_weights(snap, w2)                                                                   # same path, new contents
    other = E.estimate_pose(cfgp, snap, frames_b, str(tmp_path / "o4"), save_pose=False, batch_size=4, tier="f16")
    assert E._SESSION_CACHE["sess"] is not sess_f and not np.array_equal(other["likelihoods"], f16["likelihoods"])
    E.clear_session_cache()
    assert not E._SESSION_CACHE
