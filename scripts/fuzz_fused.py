"""Random shapes through the layer-level checks of the fused kernels (tests/test_chain_gpu.py: chain_h2 and unit_h2 against float64 on the
22-bit values each stage consumes, and against the layer-by-layer kernels): every kernel family, 1-4 frames, maps from 1 x 1 to 40 x 50 with ragged
tiles in both directions.  --h1: the kernels' instances on H1 tensors (tests/test_h1_gpu.py: dgp_chain_h1 / dgp_unit_h1).
Usage: python scripts/fuzz_fused.py [n] [seed] [--h1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_chain_gpu as T
import test_h1_gpu as T1
H1 = "--h1" in sys.argv
argv = [a for a in sys.argv if a != "--h1"]
n = int(argv[1]) if len(argv) > 1 else 60
rng = np.random.default_rng(int(argv[2]) if len(argv) > 2 else 0)
FAM = [(64, 64, 0, 1), (64, 64, 64, 0), (64, 128, 0, 2), (128, 128, 0, 1), (128, 256, 0, 2), (256, 256, 0, 1)]
bad = 0
for it in range(n):
    N = int(rng.integers(1, 5))
    H, W = (int(rng.integers(1, 5)), int(rng.integers(1, 20))) if rng.integers(0, 5) == 0 else (int(rng.integers(1, 41)), int(rng.integers(1, 51)))
    if rng.integers(0, 2):
        C, C1, CIN2, res = FAM[int(rng.integers(0, len(FAM)))]
        case, fn, name = (N, H, W, C, C1, CIN2, res), (T1.test_chain_on_h1_tensors if H1 else T.test_chain_matches_float64_and_the_layer_kernels), "chain"
    else:
        CIN2, res = [(0, 1), (64, 0)][int(rng.integers(0, 2))]
        case, fn, name = (N, H, W, CIN2, res), (T1.test_unit_kernel_on_h1_tensors if H1 else T.test_unit_kernel_matches_float64), "unit "
    try:
        fn(None, case)
        ok, msg = True, ""
    except AssertionError as e:
        ok, msg = False, str(e)[:160]
    except Exception as e:      # noqa: BLE001
        ok, msg = False, repr(e)[:160]
    bad += not ok
    print("%s %s %s  %s" % ("ok  " if ok else "BAD ", name, case, msg), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
