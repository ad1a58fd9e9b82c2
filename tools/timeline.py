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
ap.add_argument("--steady", action="store_true", help="pick steps of the timed loop (concurrent kernels) near the median duration")
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
if args.steady and len(anchors) > args.steps + 2:
    # steps of the timed loop: kernels on three or more hardware queues (the per-class timing pass of bench.py runs every
    # kernel alone on one queue, the cold-grid steps rebuild the filter tables); take the run of `steps` whose first step
    # is closest to the median duration of such steps
    spans = []
    for a, b in zip(anchors[:-1], anchors[1:]):
        names = [r[3] for r in rows[a:b]]
        if len({r[2] for r in rows[a:b]}) >= 3 and not any("gtab" in n for n in names):
            spans.append((rows[b][0] - rows[a][0], a))
    if spans:
        med = sorted(d for d, _ in spans)[len(spans) // 2]
        ok = {a for _, a in spans}
        best = None
        for d, a in spans:
            i = anchors.index(a)
            if i + args.steps < len(anchors) and all(anchors[i + k] in ok for k in range(args.steps)):
                if best is None or abs(d - med) < best[0]:
                    best = (abs(d - med), i)
        if best:
            first, last = anchors[best[1]], anchors[best[1] + args.steps]
            print(f"# {len(spans)} concurrent steps in the trace, median {med / 1e3:.1f} us; shown: steps {best[1]} .. {best[1] + args.steps - 1}")
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
