#!/usr/bin/env python3
"""Where the hot kernels spill: compiles the product sources to gfx950 assembly (hipcc -S, no GPU needed) and lists,
per kernel, the scratch (spill) instructions with their position inside the kernel's code.

    python tools/spill_report.py [kernel name substring ...]      (default: the row kernels of the bench workloads)

A kernel whose `switch` covers many (K, terms) cases is ONE register allocation: the position tells which case a spill
belongs to (the cases are laid out in source order), and how many of the instructions sit on a row's path."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
want = sys.argv[1:] or ["k_narrow_ct_all", "k_narrow_ct_many", "k_narrow_ct_big", "k_ols_ct", "k_ols_fwd_r", "k_pass_a_ct_rows<",
                        "k_pass_b_ct"]
lines = []
with tempfile.TemporaryDirectory() as tmp:
    for unit in ("launch_f64.hip", "launch_f32.hip"):
        asm = os.path.join(tmp, unit + ".s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I",
                        os.path.join(ROOT, "pycwt_amd", "csrc"), "--cuda-device-only", "-S",
                        os.path.join(ROOT, "pycwt_amd", "csrc", unit), "-o", asm], check=True, capture_output=True)
        lines += open(asm).read().splitlines()
starts = [(i, m.group(1)) for i, l in enumerate(lines) if (m := re.match(r"^(_Z\w+):", l))]
for idx, (i, sym) in enumerate(starts):
    name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void cwt::", "")
    if not any(w in name for w in want):
        continue
    end = next((k for k in range(i, len(lines)) if lines[k].startswith(".Lfunc_end")), len(lines))
    body = lines[i:end]
    insts = [l for l in body if l.startswith("\t") and not l.lstrip().startswith((".", ";"))]
    pos = {id(l): n for n, l in enumerate(insts)}
    spills = [(pos[id(l)], l.strip()) for l in insts if "scratch_" in l]
    branches = sum(1 for l in insts if "s_cbranch" in l or "s_branch" in l)
    if not spills:
        print(f"{name}: {len(insts)} instructions, no scratch instructions")
        continue
    st = sum(1 for _, l in spills if "scratch_store" in l)
    print(f"{name}: {len(insts)} instructions, {st} scratch stores + {len(spills) - st} scratch loads "
          f"({branches} branch instructions)")
    # group by position in tenths of the kernel
    for p, l in spills:
        print(f"   at {p:6d} ({100.0 * p / len(insts):5.1f} %)  {l.split(';')[0].strip():48s} {l.split(';')[-1].strip() if ';' in l else ''}")
