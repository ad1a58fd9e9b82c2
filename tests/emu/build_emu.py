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
CSRC = os.path.join(ROOT, "pycwt_amd", "csrc")
UNITS = ["plan_host.cpp", "launch_f64.hip", "launch_f32.hip", "abi.hip"]       # the product's translation units, unmodified
SRCS = [os.path.join(CSRC, u) for u in UNITS] + [os.path.join(HERE, "hipemu.cpp")]
DEPS = SRCS + [os.path.join(CSRC, f) for f in ("plan.hpp", "cwt_types.hpp", "launch_impl.hpp", "fft_engine.hpp", "cwt_kernels.hpp",
                                               "cwt_kernels_rows.hpp", "cwt_kernels_callers.hpp")] + [
    os.path.join(ROOT, "include", "cwt_hip.h"), os.path.join(HERE, "hip", "hip_runtime.h")]


def build(force=False):
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(HERE, "_build")
    os.makedirs(objdir, exist_ok=True)
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) > os.path.getmtime(d) for d in DEPS):
        return OUT
    flags = ["-O2", "-std=c++17", "-fPIC", "-pthread", "-DCWT_BACKEND_NAME=\"cpu-emulation\"",
             "-I", HERE, "-I", os.path.join(ROOT, "include"), "-I", CSRC]

    def compile_one(src):          # (object names carry the pid: pytest-xdist workers may build side by side)
        obj = os.path.join(objdir, os.path.splitext(os.path.basename(src))[0] + f".{os.getpid()}.emu.o")
        subprocess.run(["g++"] + flags + ["-x", "c++", "-c", src, "-o", obj], check=True)
        return obj
    with ThreadPoolExecutor(max_workers=len(SRCS)) as pool:
        objs = list(pool.map(compile_one, SRCS))
    tmp = OUT + f".tmp{os.getpid()}"
    subprocess.run(["g++", "-shared", "-pthread"] + objs + ["-o", tmp], check=True)
    os.replace(tmp, OUT)
    for o in objs:
        os.remove(o)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
