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
OUT_ASAN = os.path.join(HERE, "_build", "libcwt_emu_asan.so")
CSRC = os.path.join(ROOT, "pycwt_amd", "csrc")
UNITS = ["plan_host.cpp", "launch_f64.hip", "launch_f32.hip", "abi.hip"]       # the product's translation units, unmodified
SRCS = [os.path.join(CSRC, u) for u in UNITS] + [os.path.join(HERE, "hipemu.cpp")]
DEPS = SRCS + [os.path.join(CSRC, f) for f in ("plan.hpp", "cwt_types.hpp", "launch_impl.hpp", "fft_engine.hpp", "cwt_kernels.hpp",
                                               "cwt_kernels_rows.hpp", "cwt_kernels_callers.hpp")] + [
    os.path.join(ROOT, "include", "cwt_hip.h"), os.path.join(HERE, "hip", "hip_runtime.h")]


SANITIZE = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g"]


def sanitizer_env(env=None):
    """Environment for a Python process that loads the --asan build: the sanitizer runtimes must come first (the interpreter is
    not instrumented).  Leak detection off (the interpreter itself never frees everything)."""
    env = dict(os.environ if env is None else env)
    libs = []
    for name in ("libasan.so", "libubsan.so"):
        path = subprocess.run(["g++", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
        if os.path.isabs(path):
            libs.append(os.path.realpath(path))
    env["LD_PRELOAD"] = ":".join(libs + [env["LD_PRELOAD"]] if env.get("LD_PRELOAD") else libs)
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=1:detect_stack_use_after_return=0:" + env.get("ASAN_OPTIONS", "")
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1:" + env.get("UBSAN_OPTIONS", "")
    env["CWT_EMU_LIBRARY"] = OUT_ASAN
    return env


def build(force=False, asan=False):
    """asan=True: the same sources under -fsanitize=address,undefined (the index arithmetic of plan_host.cpp that
    sizes every launch and scratch buffer, the kernels' LDS / global indexing on the emulator) -> libcwt_emu_asan.so; load it in
    a process started with `sanitizer_env()`."""
    from concurrent.futures import ThreadPoolExecutor
    global OUT
    out = OUT_ASAN if asan else OUT
    objdir = os.path.join(HERE, "_build")
    os.makedirs(objdir, exist_ok=True)
    if not force and os.path.exists(out) and all(os.path.getmtime(out) > os.path.getmtime(d) for d in DEPS):
        return out
    flags = ["-O1" if asan else "-O2", "-std=c++17", "-fPIC", "-pthread", "-DCWT_BACKEND_NAME=\"cpu-emulation\"",
             "-I", HERE, "-I", os.path.join(ROOT, "include"), "-I", CSRC] + (SANITIZE if asan else [])

    def compile_one(src):          # (object names carry the pid: pytest-xdist workers may build side by side)
        obj = os.path.join(objdir, os.path.splitext(os.path.basename(src))[0] + f".{os.getpid()}.{'asan' if asan else 'emu'}.o")
        subprocess.run(["g++"] + flags + ["-x", "c++", "-c", src, "-o", obj], check=True)
        return obj
    sys.path.insert(0, ROOT)
    from pycwt_amd import _build as product_build          # the generated translation unit with cwt_build_id()
    id_src = product_build.write_id_source(os.path.join(objdir, f"build_id.{os.getpid()}.cpp"), product_build.source_id(["emu"]))
    with ThreadPoolExecutor(max_workers=len(SRCS) + 1) as pool:
        objs = list(pool.map(compile_one, SRCS + [id_src]))
    os.remove(id_src)
    tmp = out + f".tmp{os.getpid()}"
    subprocess.run(["g++", "-shared", "-pthread"] + (SANITIZE if asan else []) + objs + ["-o", tmp], check=True)
    os.replace(tmp, out)
    for o in objs:
        os.remove(o)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, asan="--asan" in sys.argv))
