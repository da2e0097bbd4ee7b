"""Differential fuzz of the host helpers against the REFERENCE's own functions, imported from /root/reference with the stub modules of
tests/golden/make_golden.py (only possible in the build container: nothing here travels).  Hundreds of random inputs per helper instead of the
handful of pinned golden vectors: marker-index helpers, hidden-frame selection, gen_batch (same seeds), locref targets, the DLC hard arg-max.
Usage: python scripts/diff_reference_host.py [n] [seed]"""
import os, random, sys
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
if not os.path.isdir("/root/reference/src"):
    print("no /root/reference here: nothing to compare with"); sys.exit(0)
import make_golden as G
G.install_stubs()
REF = G.REF
fu = G.load(os.path.join(REF, "deepgraphpose/models/fitdgp_util.py"), "ref_fitdgp_util")
ds = G.load(os.path.join(REF, "deepgraphpose/dataset.py"), "ref_dataset")
pdd = G.load(os.path.join(REF, "DeepLabCut/deeplabcut/pose_estimation_tensorflow/dataset/pose_defaultdataset.py"), "ref_pose_defaultdataset")
predict = G.load(os.path.join(REF, "DeepLabCut/deeplabcut/pose_estimation_tensorflow/nnet/predict.py"), "ref_predict")
from deepgraphpose_amd import dataset as D
from deepgraphpose_amd import config as K
from deepgraphpose_amd.models import fitdgp_util as F
from deepgraphpose_amd.models.fitdgp import _limb_statistics
from oracle import dgp_oracle as O
import make_reaching_golden as GR
FIT = os.path.join(REF, "deepgraphpose/models/fitdgp.py")
S0_LINES = GR.cut(FIT, "bodyparts = cfg['bodyparts']", "S0[s, skj] = -1")                                   # fitdgp.py:607-617
EXPORT_LINES = GR.cut(os.path.join(REF, "deepgraphpose/models/eval.py"), "def export_pose_like_dlc(labels, scorer, joints_names, save_file):", "def evaluate_dgp(", include_last=False)
LIMB_LINES = GR.cut(FIT, "joint_locs = [d.labels for d in data_batcher.datasets]", "limb_full) + 1e-20) * dgp_cfg.ws")   # :875-892

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = {}
def check(name, ok, detail=""):
    if not ok:
        bad[name] = bad.get(name, 0) + 1
        if bad[name] <= 3:
            print("DIFF %s %s" % (name, detail), flush=True)
def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)

for it in range(n):
    # ---- marker-index helpers
    nv, nh, nj = int(rng.integers(0, 7)), int(rng.integers(0, 9)), int(rng.integers(1, 8))
    frames = rng.permutation(30)[: nv + nh]
    vis, hid = np.sort(frames[:nv]), np.sort(frames[nv:])
    jl = rng.uniform(0, 50, size=(nv, nj, 2)); jl[rng.random((nv, nj)) < rng.uniform(0, 0.6)] = np.nan
    try:
        r = ds.gen_idx_chunk(vis, hid, jl); o = D.gen_idx_chunk(vis, hid, jl)
        check("gen_idx_chunk", all(same(np.asarray(a, dtype=np.int64), b) for a, b in zip(r, o)), (nv, nh, nj))
        nan_r = fu.find_nan_ind(vis, jl); nan_o = F.find_nan_ind(vis, jl)
        check("find_nan_ind", same(np.asarray(nan_r, dtype=np.int64), np.asarray(nan_o, dtype=np.int64)))
        check("find_hidden_markers", same(np.asarray(fu.find_hidden_markers(hid, nj, nan_r), dtype=np.int64), F.find_hidden_markers(hid, nj, nan_o)))
        if nv > 0 or len(nan_r) == 0:
            a0, a1 = fu.find_visible_markers(vis, nj, nan_r); b0, b1 = F.find_visible_markers(vis, nj, nan_o)
            check("find_visible_markers", same(np.asarray(a0, dtype=np.int64), b0) and same(np.asarray(a1, dtype=np.int64), b1))
        if nv > 0 and nh > 0:
            a = ds.find_marker_index(vis, hid, jl); b = D.find_marker_index(vis, hid, jl)
            check("find_marker_index", all(same(np.asarray(x, dtype=np.int64), y) for x, y in zip(a, b)))
    except Exception as e:      # noqa: BLE001 -- an exception on one side only is a difference too
        check("marker helpers raised", False, repr(e)[:120])
    # ---- windows / hidden-frame selection
    n_frames = int(rng.integers(30, 400))
    pv = np.sort(rng.choice(n_frames, size=int(rng.integers(1, min(12, n_frames))), replace=False))
    me = rng.random(n_frames); pvh = np.argsort(-me)
    ns, nmax = int(rng.integers(1, 9)), int(rng.integers(5, 200))
    nsj = [None, 0, 1, 2, 5][int(rng.integers(0, 5))]
    try:
        check("get_neighboring_window", same(ds.get_neighboring_window(pv, ns, n_frames), D.get_neighboring_window(pv, ns, n_frames)), (len(pv), ns, n_frames))
        er = eo = None
        try: a = np.asarray(ds.select_hidden_frames(ns, pv, pvh, n_frames, nmax, nsj), dtype=np.int64)
        except Exception as e: er = type(e).__name__     # noqa: BLE001,E701
        try: b = D.select_hidden_frames(ns, pv, pvh, n_frames, nmax, nsj)
        except Exception as e: eo = type(e).__name__     # noqa: BLE001,E701
        check("select_hidden_frames", (er is None) == (eo is None) and (er is not None or same(a, b)), (ns, len(pv), n_frames, nmax, nsj, er, eo))
    except Exception as e:      # noqa: BLE001
        check("window helpers raised", False, repr(e)[:120])
    # ---- gen_batch with the same seeds
    if it % 4 == 0:
        nd = int(rng.integers(1, 4))
        vis_tot = [np.sort(rng.choice(90, int(rng.integers(1, 9)), replace=False)) for _ in range(nd)]
        hid_tot = [np.sort(rng.choice(90, int(rng.integers(0, 12)), replace=False)) for _ in range(nd)]
        all_tot = [np.arange(int(rng.integers(0, 10)), int(rng.integers(12, 60))) for _ in range(nd)]
        cfg = SimpleNamespace(batch_size=int(rng.integers(1, 16)), n_times_all_frames=int(rng.integers(1, 120)))
        maxit = int(rng.integers(1, 300))
        outs = []
        for fn in (fu.gen_batch, lambda *a: F.gen_batch(*a, verbose=False)):
            np.random.seed(1000 + it); random.seed(2000 + it)
            try: outs.append([np.asarray(b, dtype=np.int64) for b in fn(vis_tot, hid_tot, all_tot, cfg, maxit)])
            except Exception as e: outs.append(type(e).__name__)     # noqa: BLE001,E701
        if isinstance(outs[0], str) or isinstance(outs[1], str):
            check("gen_batch", outs[0] == outs[1] if isinstance(outs[0], str) and isinstance(outs[1], str) else False, (outs[0] if isinstance(outs[0], str) else "ok", outs[1] if isinstance(outs[1], str) else "ok"))
        else:
            check("gen_batch", len(outs[0]) == len(outs[1]) and all(same(a, b) for a, b in zip(*outs)), (cfg.batch_size, cfg.n_times_all_frames, maxit))
    # ---- locref targets
    thr, njt = int(rng.choice([4, 8, 17, 30])), int(rng.integers(1, 9))
    size = (int(rng.integers(3, 70)), int(rng.integers(3, 90)))
    pdata = object.__new__(pdd.PoseDataset)
    pdata.cfg = SimpleNamespace(pos_dist_thresh=thr, num_joints=njt, weigh_only_present_joints=False)
    pdata.stride, pdata.half_stride, pdata.locref_scale = 8.0, 4.0, 1.0 / 7.2801
    present = np.sort(rng.choice(njt, size=int(rng.integers(1, njt + 1)), replace=False))
    coords = np.stack([rng.uniform(-20, size[1] * 8 + 20, len(present)), rng.uniform(-20, size[0] * 8 + 20, len(present))], 1)     # also outside the frame
    try:
        sc_r, _, lmap_r, lmask_r = pdata.compute_target_part_scoremap([present], [coords], 0, np.array(size), 1)
        sc_o, lmap_o, lmask_o = D.compute_target_part_scoremap([present], [coords], size, njt, thr)
        check("compute_target_part_scoremap", same(sc_r.astype(np.uint8), sc_o.astype(np.uint8)) and same(lmask_r.astype(np.uint8), lmask_o.astype(np.uint8))
              and same(lmap_r.astype(np.float64), lmap_o), (thr, njt, size))
        jl2 = np.stack([rng.uniform(-2, size[0] + 1, (int(rng.integers(1, 4)), njt)), rng.uniform(-2, size[1] + 1, (1, njt)).repeat(1, 0)], -1) if False else \
            np.stack([rng.uniform(-2, size[0] + 1, (2, njt)), rng.uniform(-2, size[1] + 1, (2, njt))], -1)
        jl2[rng.random((2, njt)) < 0.25] = np.nan
        lt_o, lm_o = D.coord2map(jl2, size[0], size[1], njt, thr)
        if njt >= 2:        # (with ONE bodypart the reference raises: dataset.py:253 flips axis 1 of a squeezed 1-D array; ours handles it)
            lt_r, lm_r = ds.coord2map(pdata, jl2, size[0], size[1], njt)
            check("coord2map", same(lt_r.astype(np.float64), lt_o) and same(lm_r.astype(np.uint8), lm_o.astype(np.uint8)), (thr, njt, size))
    except Exception as e:      # noqa: BLE001
        check("locref targets raised", False, repr(e)[:160])
    # ---- skeleton matrix and limb statistics: the reference's own lines (fitdgp.py:607-617, :875-892), exec'd as the golden generator does
    njs = int(rng.integers(2, 9))
    parts = ["part%d" % i for i in range(njs)]
    pairs = [tuple(rng.choice(njs, 2, replace=False)) for _ in range(int(rng.integers(1, 8)))]
    proj = dict(bodyparts=parts, skeleton=[[parts[a], parts[b]] for a, b in pairs])
    try:
        ns_ = dict(np=np, cfg=proj); exec(S0_LINES, ns_)
        S0r = np.asarray(ns_["S0"], dtype=np.float64)
        S0o = np.asarray(K.skeleton_matrix(proj), dtype=np.float64)
        check("skeleton_matrix", same(S0r, S0o), (njs, pairs))
        labels = []
        for dset in range(int(rng.integers(1, 4))):
            lab = rng.uniform(0, 60, size=(int(rng.integers(1, 30)), njs, 2))
            lab[rng.random(lab.shape[:2]) < rng.uniform(0, 0.5)] = np.nan
            labels.append(lab)
        ws_c, wsmax_c, stride_c = float(rng.choice([1.0, 1000.0, 3.5])), float(rng.choice([1.2, 2.0])), float(rng.choice([8.0, 4.0]))
        AD = GR.AttrDict
        ns_ = dict(np=np, data_batcher=AD(datasets=[AD(labels=l) for l in labels]), nj=njs, S0=S0r, dgp_cfg=AD(stride=stride_c, ws=ws_c, ws_max=wsmax_c))
        with np.errstate(all="ignore"):
            exec(LIMB_LINES, ns_)
            ws_o, wsmax_o = _limb_statistics(labels, S0r, stride_c, ws_c, wsmax_c)
        close = lambda a, b: np.allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), rtol=1e-12, atol=0, equal_nan=True)
        check("limb statistics", close(ns_["ws"], ws_o) and close(ns_["ws_max"], wsmax_o), (njs, len(labels), np.asarray(ns_["ws"])[:3], np.asarray(ws_o)[:3]))
    except Exception as e:      # noqa: BLE001
        check("limb statistics raised", False, repr(e)[:160])
    # ---- A5: the DLC export (eval.py:621-653), the reference's function text exec'd; to_hdf needs pytables, which neither side has here
    if it % 5 == 0:
        import pandas as pd, tempfile
        from deepgraphpose_amd.models import eval as E
        nfr, nlab = int(rng.integers(0, 40)), int(rng.integers(1, 9))
        lab = {k: rng.standard_normal((nfr, nlab)) * 10.0 ** rng.integers(-3, 5) for k in ("x", "y", "likelihoods")}
        if nfr:
            lab["x"][rng.integers(0, nfr), 0] = np.nan
        names = ["part %d/%s" % (i, "ab_c"[i % 4:]) for i in range(nlab)]      # (no commas: the reference's genfromtxt loader cannot read quoted fields)
        td = tempfile.mkdtemp()
        ns_ = dict(np=np, os=os); exec(EXPORT_LINES, ns_)
        hdf = pd.DataFrame.to_hdf
        try:
            pd.DataFrame.to_hdf = lambda self, *a, **k: None
            ns_["export_pose_like_dlc"](lab, "snapshot-5", names, os.path.join(td, "ref"))
        finally:
            pd.DataFrame.to_hdf = hdf
        E.export_pose_like_dlc(lab, "snapshot-5", names, os.path.join(td, "ours"))
        a, b = open(os.path.join(td, "ref.csv")).read(), open(os.path.join(td, "ours.csv")).read()
        check("export_pose_like_dlc csv", a == b, (nfr, nlab))
        if nfr:             # (a NaN is an empty csv field, which the reference's loader cannot convert: both sides must then fail alike)
            res = []
            for fn, path in ((ns_["load_pose_from_dlc_to_dict"], "ref.csv"), (E.load_pose_from_dlc_to_dict, "ours.csv")):
                try: res.append(fn(os.path.join(td, path)))
                except Exception as e: res.append(type(e).__name__)     # noqa: BLE001,E701
            if isinstance(res[0], str) or isinstance(res[1], str):
                check("load_pose_from_dlc_to_dict", res[0] == res[1], (res[0] if isinstance(res[0], str) else "ok", res[1] if isinstance(res[1], str) else "ok"))
            else:
                check("load_pose_from_dlc_to_dict", all(same(res[0][k], res[1][k]) for k in ("x", "y", "likelihoods")))
    # ---- DLC hard arg-max (the oracle's restatement against the reference's function)
    h, w, c = int(rng.integers(1, 40)), int(rng.integers(1, 50)), int(rng.integers(1, 8))
    logits = (rng.standard_normal((h, w, c)) * rng.uniform(0.5, 20)).astype(np.float32)
    scm = O.sigmoid_f32(logits)
    off = (rng.standard_normal((h, w, c, 2)).astype(np.float32) * np.float32(7.2801)) if rng.integers(0, 2) else None
    pose_r = np.asarray(predict.argmax_pose_predict(scm, off, 8.0), dtype=np.float64)
    pose_o, _ = O.argmax_pose_predict(scm, off, 8.0)
    check("argmax_pose_predict", same(pose_r, pose_o), (h, w, c))
print("compared %d random inputs per helper; differences: %s" % (n, bad if bad else "none"))
sys.exit(1 if bad else 0)
