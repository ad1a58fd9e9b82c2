"""Build pycwt_amd/libcwt_hip.so for gfx950 (cross-compiles without a GPU).

    python -m pycwt_amd._build [--force]

Four translation units, compiled side by side and linked by hipcc:
  csrc/plan_host.cpp   row classification, row-table cache, scratch, host copies -- host C++ only (seconds)
  csrc/launch_f64.hip  every kernel launch for double  }  launch_impl.hpp instantiated once per precision: the two halves of
  csrc/launch_f32.hip  ... and for float               }  the device code (~45 s each) build in parallel
  csrc/abi.hip         the exported C functions + the few small kernels they launch directly
An edit of the host logic (plan_host.cpp) rebuilds in a few seconds: only that object and the link.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_build")
OUT = os.path.join(HERE, "libcwt_hip.so")
UNITS = ["plan_host.cpp", "launch_f64.hip", "launch_f32.hip", "abi.hip"]
SOURCES = [os.path.join(CSRC, u) for u in UNITS]
HOST_HEADERS = [os.path.join(CSRC, f) for f in ("plan.hpp", "cwt_types.hpp")] + [os.path.join(ROOT, "include", "cwt_hip.h")]
DEVICE_HEADERS = [os.path.join(CSRC, f) for f in ("launch_impl.hpp", "fft_engine.hpp", "cwt_kernels.hpp", "cwt_kernels_rows.hpp",
                                                  "cwt_kernels_callers.hpp")]
DEPS = SOURCES + HOST_HEADERS + DEVICE_HEADERS
ARCH = "gfx950"


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build the HIP extension (there is no CPU fallback)")


def up_to_date() -> bool:
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS)


def _deps_of(unit: str) -> list[str]:
    src = os.path.join(CSRC, unit)
    return [src] + HOST_HEADERS + ([] if unit == "plan_host.cpp" else DEVICE_HEADERS)


def _compile(unit: str, obj: str, extra: list[str], verbose: bool) -> None:
    if unit.endswith(".cpp"):                                    # host only: the C++ compiler against the HIP runtime API headers
        rocm = os.path.dirname(os.path.dirname(os.path.realpath(hipcc())))
        cmd = [shutil.which("g++") or "g++", "-O2", "-std=c++17", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(rocm, "include")]
    else:
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
    cmd += ["-I", os.path.join(ROOT, "include"), "-I", CSRC] + extra + ["-c", os.path.join(CSRC, unit), "-o", obj]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)


def build(force: bool = False, verbose: bool = False, out: str = OUT, extra: list[str] | None = None, objdir: str = OBJ) -> str:
    """`extra`: more compiler flags (-D variants of tools/), with their own `out` and `objdir`."""
    if not force and out == OUT and up_to_date():
        return OUT
    os.makedirs(objdir, exist_ok=True)
    todo, objs = [], []
    for unit in UNITS:
        obj = os.path.join(objdir, os.path.splitext(unit)[0] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or any(os.path.getmtime(obj) < os.path.getmtime(d) for d in _deps_of(unit)):
            todo.append((unit, obj))
    with ThreadPoolExecutor(max_workers=len(UNITS)) as pool:
        for f in [pool.submit(_compile, unit, obj, extra or [], verbose) for unit, obj in todo]:
            f.result()
    tmp = out + f".tmp{os.getpid()}"
    subprocess.run([hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC"] + objs + ["-o", tmp], check=True)
    os.replace(tmp, out)
    return out


def ensure(local_rank: int = 0, timeout_s: float = 600.0) -> str:
    """Make sure the library exists when several ranks start at once on a box that did not receive
    a prebuilt one: local rank 0 builds it (into a temporary name, renamed into place), the other
    ranks wait for the file.  An existing library is used as is (no mtime check: a snapshot copy
    does not preserve mtimes)."""
    import time
    if os.path.exists(OUT):
        return OUT
    if local_rank == 0:
        return build(force=True)
    t0 = time.time()
    while not os.path.exists(OUT):
        if time.time() - t0 > timeout_s:
            raise RuntimeError(f"{OUT} was not built by local rank 0 within {timeout_s:.0f} s")
        time.sleep(0.5)
    return OUT


if __name__ == "__main__":
    import time
    t = time.time()
    print(build(force="--force" in sys.argv, verbose=True), f"{time.time() - t:.1f} s")
