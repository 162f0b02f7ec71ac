#!/usr/bin/env python3
"""Build libuclstm's HOST code with AddressSanitizer + UBSan and run the planner sweep (tools/host_asan/driver.cpp).

    python tools/host_asan/run.py        # ~2 min of hipcc; prints the driver's summary line, exit code 0 = clean

GPU AddressSanitizer is not available on this pool, so only the host side is instrumented (-Xarch_host); the device code is
compiled as usual because the host objects embed it.  Nothing is launched: no GPU needed.  Output goes to
unet-convlstm_amd/build_asan/ (git-ignored).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "unet-convlstm_amd")
sys.path.insert(0, PKG)
import build as B   # noqa: E402

OUT = os.path.join(PKG, "build_asan")
SAN = ["-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined", "-Xarch_host", "-fno-omit-frame-pointer", "-g"]


def main() -> int:
    os.makedirs(OUT, exist_ok=True)
    hipcc = B._hipcc()

    def one(src):
        obj = os.path.join(OUT, src.replace(".hip", ".o"))
        subprocess.run([hipcc, *B.FLAGS, *SAN, "-c", os.path.join(B.CSRC, src), "-o", obj], check=True)
        return obj

    with ThreadPoolExecutor(max_workers=5) as ex:
        objs = list(ex.map(one, B.SOURCES))
    lib = os.path.join(OUT, "libuclstm_asan.so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address", "-fsanitize=undefined", "-o", lib, *objs], check=True)
    exe = os.path.join(OUT, "driver")
    subprocess.run([hipcc, "-x", "c++", "--cuda-host-only", "-std=c++17", "-O1", "-g", "-fsanitize=address", "-fsanitize=undefined",
                    "-fno-omit-frame-pointer", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "host_asan", "driver.cpp"),
                    "-o", exe, "-L", OUT, "-luclstm_asan", f"-Wl,-rpath,{OUT}"], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    return subprocess.run([exe], env=env).returncode


if __name__ == "__main__":
    raise SystemExit(main())
