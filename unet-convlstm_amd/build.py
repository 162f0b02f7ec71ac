"""Build libuclstm.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

The shared object lands next to this file (``unet-convlstm_amd/libuclstm.so``) so that it
travels with the repository snapshot to the GPU box; nothing is installed or cached elsewhere.

What is shipped is what is tracked: the library carries the SHA-256 of the sources it was built from
(``uclstm_source_hash()``: csrc/*, include/uclstm.h and the compiler flags, compiled into a generated
translation unit at link time), every object file has a sidecar with the hash of its own inputs, and both
``build()`` and the loader (``_lib._load``) compare against the tree -- a stale library is rebuilt here and a
loud error there, never a silent reuse.  File modification times are not consulted.
A/B builds of kernel variants belong under ``unet-convlstm_amd/ab/`` (git- and gpurun-ignored), loaded with
``UCLSTM_LIB=``.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HEADER = os.path.join(HERE, "..", "include", "uclstm.h")
LIB = os.path.join(HERE, "libuclstm.so")
SOURCES = ["igemm_fwd.hip", "igemm_wgrad.hip", "pointwise.hip", "pack.hip", "loss_optim.hip"]
# the sources that touch 16-bit activations / panels are compiled a second time for IEEE binary16 (entry points *_f16)
F16_SOURCES = ["igemm_fwd.hip", "igemm_wgrad.hip", "pointwise.hip", "pack.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"]
F16_FLAGS = ["-DUCLSTM_ACT_F16", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _sha(parts) -> str:
    h = hashlib.sha256()
    for p in parts:
        if isinstance(p, str):
            p = p.encode()
        h.update(len(p).to_bytes(8, "little"))
        h.update(p)
    return h.hexdigest()


def _read(path: str) -> bytes:
    with open(path, "rb") as f:
        return f.read()


def source_hash(csrc: str = CSRC, header: str = HEADER, sources=None, f16_sources=None) -> str:
    """SHA-256 over every file of ``csrc`` (name + bytes, sorted), the public header, the source lists and the flags."""
    sources = SOURCES if sources is None else sources
    f16_sources = F16_SOURCES if f16_sources is None else f16_sources
    parts = [" ".join(FLAGS), " ".join(F16_FLAGS), " ".join(sources), " ".join(f16_sources), _read(header)]
    for name in sorted(os.listdir(csrc)):
        path = os.path.join(csrc, name)
        if os.path.isfile(path):
            parts += [name, _read(path)]
    return _sha(parts)


def _object_hash(csrc: str, header: str, src: str, f16: bool) -> str:
    """Inputs of one object: its source, every header of csrc/, the public header, the flags of its pass."""
    parts = [" ".join(FLAGS), " ".join(F16_FLAGS) if f16 else "", src, _read(os.path.join(csrc, src)), _read(header)]
    for name in sorted(os.listdir(csrc)):
        if name.endswith(".h"):
            parts += [name, _read(os.path.join(csrc, name))]
    return _sha(parts)


def library_hash(lib: str) -> str | None:
    """The source hash a built library reports (None: no such file / symbol).  Read through a throw-away ctypes handle on
    the host side of the library only -- no HIP call is made."""
    if not os.path.exists(lib):
        return None
    side = lib + ".srchash"
    # The sidecar is written with the library; reading it avoids dlopen()ing a HIP library in a process that only builds.
    if os.path.exists(side):
        lines = _read(side).decode().split()
        if len(lines) == 2 and lines[1] == hashlib.sha256(_read(lib)).hexdigest():
            return lines[0]
    return None


def needs_rebuild(lib: str = LIB, csrc: str = CSRC, header: str = HEADER, sources=None, f16_sources=None) -> bool:
    return library_hash(lib) != source_hash(csrc, header, sources, f16_sources)


def build(force: bool = False, verbose: bool = True, *, csrc: str = CSRC, header: str = HEADER, lib: str = LIB,
          objdir: str | None = None, sources=None, f16_sources=None) -> str:
    """Compile every HIP source for gfx950 and link the library. Returns its path.  Objects whose inputs are unchanged
    (content hash, not mtime) are reused; the link always embeds the hash of the whole source set."""
    sources = SOURCES if sources is None else sources
    f16_sources = F16_SOURCES if f16_sources is None else f16_sources
    want = source_hash(csrc, header, sources, f16_sources)
    if not force and library_hash(lib) == want:
        return lib
    hipcc = _hipcc()
    objdir = objdir or os.path.join(os.path.dirname(lib), "build")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(job) -> str:
        src, f16 = job
        obj = os.path.join(objdir, src.replace(".hip", "_f16.o" if f16 else ".o"))
        oh = _object_hash(csrc, header, src, f16)
        if not force and os.path.exists(obj) and os.path.exists(obj + ".hash") and _read(obj + ".hash").decode().strip() == oh:
            return obj
        cmd = [hipcc, *FLAGS, *(F16_FLAGS if f16 else []), f"-I{os.path.dirname(os.path.abspath(header))}", "-c", os.path.join(csrc, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        if os.path.exists(obj + ".hash"):
            os.remove(obj + ".hash")
        subprocess.run(cmd, check=True)
        with open(obj + ".hash", "w") as f:
            f.write(oh + "\n")
        return obj

    jobs = [(s, False) for s in sources] + [(s, True) for s in f16_sources]
    with ThreadPoolExecutor(max_workers=5) as ex:
        objs = list(ex.map(compile_one, jobs))
    # the generated translation unit that makes the library say what it was built from
    stamp = os.path.join(objdir, "srchash.cpp")
    with open(stamp, "w") as f:
        f.write(f'extern "C" const char* uclstm_source_hash(void) {{ return "{want}"; }}\n')
    stamp_o = os.path.join(objdir, "srchash.o")
    subprocess.run([hipcc, "-O1", "-fPIC", "-x", "c++", "-c", stamp, "-o", stamp_o], check=True)
    for stale in (lib, lib + ".srchash"):
        if os.path.exists(stale):
            os.remove(stale)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, stamp_o]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(lib + ".srchash", "w") as f:
        f.write(f"{want}\n{hashlib.sha256(_read(lib)).hexdigest()}\n")
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
