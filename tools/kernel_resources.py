#!/usr/bin/env python3
"""Compile the HIP library for gfx950 and print per-kernel register / scratch / occupancy."""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = ""
for unit in ("launch_f64.hip", "launch_f32.hip", "abi.hip"):          # the translation units that hold device code
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", "include", "-I", "pycwt_amd/csrc", "-c",
           "pycwt_amd/csrc/" + unit, "-o", "/tmp/_res.o", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
    out += subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", name).replace("void cwt::", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(\w[\w ]*?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
    if "error" in line:
        print(line)
print(f"{'kernel':44s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'occ':>4s} {'vspill':>6s}")
for k, v in rows.items():
    print(f"{k:44s} {v.get('VGPRs', -1):5d} {v.get('AGPRs', -1):5d} {v.get('TotalSGPRs', -1):5d} "
          f"{v.get('ScratchSize', -1):8d} {v.get('Occupancy', -1):4d} {v.get('VGPRs Spill', -1):6d}")
