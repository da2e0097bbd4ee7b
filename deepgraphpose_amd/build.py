"""In-tree build of libdgp_hip.so (hipcc, gfx950 only).  `python -m deepgraphpose_amd.build`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdgp_hip.so")
SOURCES = ["dgp_kernels.hip", "dgp_loss.hip", "dgp_net.hip", "dgp_train.hip"]
HEADERS = ["dgp_internal.h", "dgp_engine.h", os.path.join("..", "..", "include", "dgp_hip.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print("[dgp build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
