"""Build pycwt_amd/libcwt_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m pycwt_amd._build [--force]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libcwt_hip.so")
SOURCES = [os.path.join(CSRC, "cwt_abi.hip")]
DEPS = SOURCES + [os.path.join(CSRC, f) for f in ("fft_engine.hpp", "cwt_kernels.hpp")] + [
    os.path.join(ROOT, "include", "cwt_hip.h")]
ARCH = "gfx950"


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build the HIP extension (there is no CPU fallback)")


def up_to_date() -> bool:
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and up_to_date():
        return OUT
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-unused-result", "-I", os.path.join(ROOT, "include"), "-I", CSRC] + SOURCES + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return OUT


def ensure(local_rank: int = 0, timeout_s: float = 600.0) -> str:
    """Make sure the library exists when several ranks start at once on a box that did not receive
    a prebuilt one: local rank 0 compiles to a temporary name and renames it into place, the other
    ranks wait for the file.  An existing library is used as is (no mtime check: a snapshot copy
    does not preserve mtimes)."""
    import time
    if os.path.exists(OUT):
        return OUT
    if local_rank == 0:
        tmp = OUT + f".tmp{os.getpid()}"
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-shared", "-fPIC",
               "-Wno-unused-result", "-I", os.path.join(ROOT, "include"), "-I", CSRC] + SOURCES + ["-o", tmp]
        subprocess.run(cmd, check=True)
        os.replace(tmp, OUT)
        return OUT
    t0 = time.time()
    while not os.path.exists(OUT):
        if time.time() - t0 > timeout_s:
            raise RuntimeError(f"{OUT} was not built by local rank 0 within {timeout_s:.0f} s")
        time.sleep(0.5)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
