"""TEST INFRASTRUCTURE ONLY: compile the unmodified kernel sources against the CPU stand-in for
the HIP runtime (tests/emu/hip/hip_runtime.h) into tests/emu/_build/libcwt_emu.so.

The product never loads this library; pycwt_amd._hip.load() only opens pycwt_amd/libcwt_hip.so.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "_build", "libcwt_emu.so")
SRCS = [os.path.join(ROOT, "pycwt_amd", "csrc", "cwt_abi.hip"), os.path.join(HERE, "hipemu.cpp")]
DEPS = SRCS + [os.path.join(ROOT, "pycwt_amd", "csrc", f) for f in ("fft_engine.hpp", "cwt_kernels.hpp")] + [
    os.path.join(ROOT, "include", "cwt_hip.h"), os.path.join(HERE, "hip", "hip_runtime.h")]


def build(force=False):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) > os.path.getmtime(d) for d in DEPS):
        return OUT
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", "-DCWT_BACKEND_NAME=\"cpu-emulation\"",
           "-I", HERE, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "pycwt_amd", "csrc"),
           "-x", "c++", SRCS[0], SRCS[1], "-o", OUT]
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
