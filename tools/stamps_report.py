#!/usr/bin/env python3
"""Phase-stamp timeline of the two-pass kernels (plan option "stamps"; GPU only).

    python tools/stamps_report.py [--config c2] [--opt key=value ...] [--out file.txt]

Runs one transform of the bench workload with stamping on and prints, per launch of pass A / pass B: wall time
first-start -> last-acknowledge, the distribution of the per-workgroup phases (input wait, FFT, store issue, store
acknowledge), how many workgroups were resident over time, and the gap to the next launch.  Clock: s_memrealtime
(100 MHz, 10 ns ticks), comparable across CUs.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (scale grid helpers)
from pycwt_amd import _hip  # noqa: E402


def pct(a, q):
    return float(np.percentile(a, q)) if len(a) else float("nan")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    kind, param, prec, label = bench.CONFIGS[args.config]
    N, dt, rows = 1 << 20, 1.0, 256
    sj = bench.scale_grid(N, dt, bench.flambda_of(kind, param), rows)
    opts = {k: int(v) for k, v in (o.split("=") for o in args.opt)}
    plan = _hip.Plan(N, prec, max_rows=rows, options=opts)
    es = 8 if prec == 64 else 4
    x = np.random.default_rng(1234).standard_normal(N).astype(plan.real)
    xd, xh = _hip.DeviceBuffer(N * es), _hip.DeviceBuffer(N * 2 * es)
    W = _hip.DeviceBuffer(rows * N * 2 * es)
    xd.upload(plan, x)

    def step():
        plan.forward_fft(xd.ptr, N, xh.ptr)
        plan.transform_rows(xh.ptr, kind, param, dt, sj, W.ptr, N, N)

    for _ in range(3):
        step()
    plan.sync()
    cap = 1 << 19
    plan.set_option("stamps", cap)
    step()
    plan.sync()
    n, rec = plan.read_stamps(cap)
    plan.set_option("stamps", 0)
    lines = [f"# {label}; options {opts}; {n} workgroup records; tick = 10 ns"]
    rec = rec.astype(np.int64)
    # split into launches: blockIdx (0,0) starts a launch
    bx, by = rec[:, 7] & 0xffffffff, rec[:, 7] >> 32
    starts = np.flatnonzero((bx == 0) & (by == 0))
    ends = list(starts[1:]) + [len(rec)]
    t0 = rec[:, 0].min()
    prev_end = None
    for li, (a, b) in enumerate(zip(starts, ends)):
        r = rec[a:b]
        gx, gy = int(bx[a:b].max()) + 1, int(by[a:b].max()) + 1
        s = (r[:, :5] - t0) / 100.0          # us
        first, last = s[:, 0].min(), s[:, 4].max()
        wait, fft, issue, ack, life = (s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 4] - s[:, 3],
                                       s[:, 4] - s[:, 0])
        full = r[:, 1] > 0                       # band classes of pass A record only slots 0, 3, 4
        kind_l = "passA" if li % 2 == 0 else "passB"      # launches alternate: A(chunk 0), B(chunk 0), A(chunk 1), ...
        xcc = (r[:, 6] >> 32) & 0xf
        cu = ((r[:, 6] >> 8) & 0xf) | (((r[:, 6] >> 13) & 0x7) << 4) | (xcc << 8)
        gap = "" if prev_end is None else f" gap_from_prev_end {first - prev_end:6.2f}us"
        lines.append(f"launch {li:2d} {kind_l} grid {gx}x{gy} ({len(r)} WGs on {len(np.unique(cu))} CUs): start {first:8.2f}us "
                     f"dur {last - first:7.2f}us{gap}")
        lines.append(f"    WG life p10/p50/p90/max {pct(life,10):6.2f} {pct(life,50):6.2f} {pct(life,90):6.2f} {life.max():6.2f} us;"
                     f" last WG start at +{s[:, 0].max() - first:6.2f}us; first WG end at +{s[:, 4].min() - first:6.2f}us")
        if full.any():
            lines.append(f"    phases (full-class WGs, n={int(full.sum())}): input wait p50/p90 {pct(wait[full],50):5.2f}/{pct(wait[full],90):5.2f}"
                         f"  fft {pct(fft[full],50):5.2f}/{pct(fft[full],90):5.2f}  store issue {pct(issue[full],50):5.2f}/{pct(issue[full],90):5.2f}"
                         f"  store ack {pct(ack[full],50):5.2f}/{pct(ack[full],90):5.2f} us")
        if (~full).any():
            body = s[~full, 3] - s[~full, 0]
            lines.append(f"    band-class WGs (n={int((~full).sum())}): start->stores issued p50/p90 {pct(body,50):5.2f}/{pct(body,90):5.2f}"
                         f"  store ack {pct(ack[~full],50):5.2f}/{pct(ack[~full],90):5.2f} us")
        # residency over time (10 samples)
        ts = np.linspace(first, last, 11)[:-1] + (last - first) / 20
        res = [(int(((s[:, 0] <= t) & (s[:, 4] > t)).sum())) for t in ts]
        lines.append("    resident WGs at 5%,15%..95% of the launch: " + " ".join(f"{v:4d}" for v in res))
        # phase census at the same instants (full-class only): waiting-for-input / fft / storing
        if full.any():
            sf = s[full]
            cen = []
            for t in ts:
                w = int(((sf[:, 0] <= t) & (sf[:, 1] > t)).sum())
                f = int(((sf[:, 1] <= t) & (sf[:, 2] > t)).sum())
                st = int(((sf[:, 2] <= t) & (sf[:, 4] > t)).sum())
                cen.append(f"{w}/{f}/{st}")
            lines.append("    census wait/fft/store: " + " ".join(cen))
        prev_end = last
    total = (rec[:, 4].max() - rec[:, 0].min()) / 100.0
    lines.append(f"# first start -> last acknowledge of all stamped launches: {total:.2f} us")
    txt = "\n".join(lines)
    print(txt)
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        open(args.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
