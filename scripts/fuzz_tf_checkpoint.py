"""Random variable sets through the TF-1 checkpoint writer and reader (V2 bundles and V1 tensor-slice files; CPU only): 1-400 variables, ranks 0-5,
empty and one-element tensors, names that share long prefixes (the SSTable's prefix compression and block restarts), tables that span many data
blocks, checksum verification on, and load_checkpoint's format detection.  Usage: python scripts/fuzz_tf_checkpoint.py [n] [seed]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepgraphpose_amd import tf_checkpoint as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(n):
    nv = int(rng.choice([1, 2, 5, 40, 400])) if it % 3 else int(rng.integers(1, 60))
    tensors = {}
    stems = ["resnet_v1_50/block%d/unit_%d/bottleneck_v1/conv%d" % (a, b, c) for a in range(1, 5) for b in range(1, 7) for c in range(1, 4)]
    for v in range(nv):
        rank = int(rng.integers(0, 6))
        shape = tuple(int(rng.choice([0, 1, 2, 3, 5, 8, 17])) if rng.integers(0, 12) == 0 else int(rng.integers(1, 9)) for _ in range(rank))
        name = stems[int(rng.integers(0, len(stems)))] + "/" + "".join(rng.choice(list("abcdefghij_/"), size=int(rng.integers(1, 30)))) + str(v)
        kind = int(rng.integers(0, 4))
        a = rng.standard_normal(shape).astype(np.float32)
        if kind == 1:
            a = (a * 1e30).astype(np.float32)
        elif kind == 2 and a.size:
            a.flat[0] = np.float32(np.nan); a.flat[-1] = np.float32(-np.inf)
        tensors[name] = a
    tmp = tempfile.mkdtemp()
    ok, msg = True, ""
    for fmt in ("v2", "v1"):
        p = os.path.join(tmp, "model.ckpt-%d" % it) + ("" if fmt == "v2" else ".v1")
        try:
            (T.write_v2 if fmt == "v2" else T.write_v1)(p, tensors)
            got = T.read_v2(p, verify=True) if fmt == "v2" else T.read_v1(p)
            got2 = T.load_checkpoint(p)
            same = set(got) == set(tensors) and set(got2) == set(tensors) and T.is_tf_checkpoint(p)
            for k, a in tensors.items():
                same = same and got[k].shape == a.shape and got[k].dtype == np.float32 and np.array_equal(got[k].view(np.uint32), a.view(np.uint32)) \
                       and np.array_equal(got2[k].view(np.uint32), a.view(np.uint32))
            if not same:
                ok = False; msg += " %s: content differs" % fmt
        except Exception as e:      # noqa: BLE001
            ok = False; msg += " %s: %r" % (fmt, e)
    bad += not ok
    print("%s %3d variables, %7d floats%s" % ("ok  " if ok else "BAD ", nv, sum(a.size for a in tensors.values()), msg[:200]), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
