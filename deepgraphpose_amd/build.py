"""In-tree build of libdgp_hip.so (hipcc, gfx950 only).  `python -m deepgraphpose_amd.build`.

Every source is compiled to its own object (in parallel) and the objects are linked into the shared library.  Whether a
step can be skipped is decided by CONTENT: the sha256 of the source, of every header and of the command line is stored
next to each object (`build/*.o.sha`) and next to the library (`libdgp_hip.so.sha`); file times play no role, so a
snapshot copied to another machine rebuilds exactly what changed there and nothing else.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libdgp_hip.so")
SOURCES = ["dgp_kernels.hip", "dgp_ops.hip", "dgp_chain.hip", "dgp_loss.hip", "dgp_net.hip", "dgp_train.hip"]
HEADERS = ["dgp_internal.h", "dgp_device.h", "dgp_engine.h", os.path.join("..", "..", "include", "dgp_hip.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def _sha(paths, extra: str = "") -> str:
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(hashlib.sha256(f.read()).digest())
    return h.hexdigest()


def _read(path: str) -> str:
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return ""


def _extra_flags() -> list:
    return os.environ.get("DGP_BUILD_FLAGS", "").split()


def source_hash() -> str:
    """sha256 over every source, every header and the compile flags: what libdgp_hip.so.sha must hold for the library to be current"""
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return _sha(deps, " ".join(FLAGS + _extra_flags()))


def needs_build() -> bool:
    return not os.path.exists(LIB) or _read(LIB + ".sha") != source_hash()


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)
    flags = FLAGS + _extra_flags()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]

    def compile_one(src: str) -> str:
        path = os.path.join(CSRC, src)
        obj = os.path.join(OBJDIR, src + ".o")
        want = _sha([path] + hdrs, " ".join(flags))
        if not force and os.path.exists(obj) and _read(obj + ".sha") == want:
            return obj
        cmd = [hipcc] + flags + ["-c", path, "-o", obj]
        if verbose:
            print("[dgp build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(obj + ".sha", "w") as f:
            f.write(want)
        return obj

    jobs = max(1, min(len(SOURCES), os.cpu_count() or 1))
    with ThreadPoolExecutor(jobs) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + objs
    if verbose:
        print("[dgp build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(LIB + ".sha", "w") as f:
        f.write(source_hash())
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
