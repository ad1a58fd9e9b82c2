#!/usr/bin/env python3
"""profiles/rNN_per_class.txt from the default bench line:  python tools/per_class_table.py profiles/r03_bench_default.json
(the JSON line of `python bench.py`: config 2 at the top level, the two config-3 workloads under `extra`; the traffic
ratios are the ones bench.py read from profiles/traffic_<config>.json)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"""# Per row class of the bench workloads (python bench.py, default command; N = 2^20, 256 scales; HIP events in the
# profiling pass of bench.py, every kernel alone; `roofline.per_class` of {sys.argv[1]}).
# frac = rows x N x sizeof(complex) / time / 8 TB/s; traffic = PMC bytes (FETCH_SIZE x2 + WRITE_SIZE) / algorithmic bytes.  Config 2:
# measured inside the bench run (bench.py live_traffic: two rocprofv3 child passes) and calibrated on k_poly_rows (exact output
# bytes) and k_icwt (exact input bytes) -- both factors came out as 1.0000; config 3: profiles/traffic_<config>.json (the same
# passes run by tools/gpu_profile.sh in the same session).""")
blocks = [(d["config"]["workload"], d)] + [(v["workload"], v) for v in d.get("extra", {}).values() if "roofline" in v and "per_class" in v.get("roofline", {})]
for name, b in blocks:
    r = b["roofline"]
    fi = b.get("from_idle", {})
    print(f"\n== {name}: {b['value']:.1f} GSamples*scales/s, {b['ms_per_step']:.3f} ms per step at sustained clocks "
          f"(from idle: {fi.get('ms_per_step', float('nan')):.3f}), whole path {r['whole_path_frac']:.3f} of 8 TB/s over the timed "
          f"step, cold scale grid {b.get('cold_grid_ms', float('nan')):.2f} ms, worst row error {b['parity']['max_row_err']:.2e}")
    print(f"{'class':14s} {'rows':>5s} {'us/step':>9s} {'us/row':>8s} {'frac':>6s} {'traffic':>8s}  kernels")
    tot = 0.0
    for c, v in r["per_class"].items():
        tr = v.get("traffic_ratio")
        print(f"{c:14s} {v['rows']:5d} {v['ms_per_step'] * 1e3:9.1f} {v['us_per_row']:8.2f} {v['frac']:6.3f} "
              f"{(f'{tr:.2f}x' if tr else '-'):>8s}  {'+'.join(v['kernels'])} ({v['launches_per_step']:.0f} launches)")
        tot += v["ms_per_step"]
    for k, ms in r["shared_kernels_ms_per_step"].items():
        print(f"{k:14s} {'':5s} {ms * 1e3:9.1f}")
        tot += ms
    print(f"{'sum of kernels':14s} {'':5s} {tot * 1e3:9.1f}")

for k, v in d.get("extra", {}).items():
    if not ("roofline" in v and "per_class" in v.get("roofline", {})):
        print(f"\n== extra.{k}: " + ", ".join(f"{a} = {b:.4g}" if isinstance(b, float) else f"{a} = {b}" for a, b in v.items()
                                                if not isinstance(b, (dict, list)) and a not in ("workload", "includes", "parity_all_rows")))
        print("   " + str(v.get("workload", "")))
if "icwt" in d:
    print("\n== icwt (standalone pass over the device-resident W of config 2): " + ", ".join(f"{a} = {b:.4g}" if isinstance(b, float) else f"{a} = {b}" for a, b in d["icwt"].items()))
