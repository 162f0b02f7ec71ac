"""Build libuclstm.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

The shared object lands next to this file (``unet-convlstm_amd/libuclstm.so``) so that it
travels with the repository snapshot to the GPU box; nothing is installed or cached elsewhere.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libuclstm.so")
SOURCES = ["igemm_fwd.hip", "igemm_wgrad.hip", "pointwise.hip", "pack.hip", "loss_optim.hip"]
# the sources that touch 16-bit activations / panels are compiled a second time for IEEE binary16 (entry points *_f16)
F16_SOURCES = ["igemm_fwd.hip", "igemm_wgrad.hip", "pointwise.hip", "pack.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "uclstm.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every HIP source for gfx950 and link libuclstm.so. Returns the library path."""
    if not force and not _stale():
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(job) -> str:
        src, f16 = job
        obj = os.path.join(objdir, src.replace(".hip", "_f16.o" if f16 else ".o"))
        cmd = [hipcc, *FLAGS, *(["-DUCLSTM_ACT_F16", "-Wno-unused-function"] if f16 else []), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    jobs = [(s, False) for s in SOURCES] + [(s, True) for s in F16_SOURCES]
    with ThreadPoolExecutor(max_workers=5) as ex:
        objs = list(ex.map(compile_one, jobs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
