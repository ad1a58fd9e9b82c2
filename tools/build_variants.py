#!/usr/bin/env python3
"""Build tuning variants of libcwt_hip.so (extra -D flags) into tools/experiments/_variants/<name>.so.

    python tools/build_variants.py name1:-DX=1,-DY=2 name2:-DZ=3 ...

The variants travel to the GPU box with the snapshot; tools/gpu_variants.sh copies each one over
pycwt_amd/libcwt_hip.so of the scratch copy and runs bench.py.  Nothing here is part of the product.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pycwt_amd import _build  # noqa: E402

OUT = os.path.join(ROOT, "tools", "experiments", "_variants")


def one(spec):
    name, _, flags = spec.partition(":")
    out = os.path.join(OUT, name + ".so")
    cmd = [_build.hipcc(), f"--offload-arch={_build.ARCH}", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-unused-result", "-I", os.path.join(ROOT, "include"), "-I", _build.CSRC] + \
          [f for f in flags.split(",") if f] + _build.SOURCES + ["-o", out]
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    with ThreadPoolExecutor(max_workers=6) as ex:
        for o in ex.map(one, sys.argv[1:]):
            print("built", o)
