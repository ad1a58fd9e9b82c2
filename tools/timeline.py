#!/usr/bin/env python3
"""Kernel timeline of the last `--steps` steps of a rocprofv3 --kernel-trace run of bench.py.

    python tools/timeline.py <dir with *kernel_trace.csv> [--steps 2] [--anchor k_pass_a_ct<]

A step starts with the forward FFT's pass A (`--anchor`).  Prints start / end / duration (us, from the first listed
kernel), the hardware queue, the kernel; then how much of the window had 0 / 1 / 2+ kernels in flight.
"""
import argparse
import csv
import glob
import os
import re

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--anchor", default="k_pass_a_ct<")
args = ap.parse_args()
files = glob.glob(os.path.join(args.dir, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"),
                     re.sub(r"^void cwt::", "", r["Kernel_Name"]).split("(")[0]))
rows.sort()
anchors = [i for i, r in enumerate(rows) if r[3].startswith(args.anchor)]
first = anchors[-args.steps - 1] if len(anchors) > args.steps else 0
last = anchors[-1] if anchors else len(rows)
sel = rows[first:last]
t0 = sel[0][0]
queues = {q: i + 1 for i, q in enumerate(sorted({r[2] for r in sel}))}
print(f"{'start':>9s} {'end':>9s} {'dur':>8s} queue kernel")
for s, e, q, k in sel:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {queues[q]:5d} {k}")
ev = sorted([(s, 1) for s, e, q, k in sel] + [(e, -1) for s, e, q, k in sel])
depth, prev, hist = 0, ev[0][0], {}
for t, d in ev:
    hist[min(depth, 2)] = hist.get(min(depth, 2), 0) + (t - prev)
    depth += d
    prev = t
tot = sum(hist.values())
print("# kernels in flight: " + ", ".join(f"{'2+' if k == 2 else k}: {v / 1e3:.1f} us ({100 * v / tot:.1f} %)" for k, v in sorted(hist.items())))
