#!/usr/bin/env python3
"""bench.py -- throughput of the CWT hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3_paul|c3_dog]

One "step" = one pass of the hot path over one synthetic signal: (broadcast of the signal when
N > 1) -> forward FFT -> all rows of W written device-resident.  Inputs are resident in HBM when the
timed region starts.  Workload at 1 GPU = BASELINE.json configs[1]: N = 2^20 fp64 samples, Morlet(6),
256 scales spanning s0 = 2dt/flambda .. N*dt.  With G GPUs (one process per GPU,
launched by torch.distributed.run) the scale grid is refined to 256*G rows over the same span and
row j goes to rank j mod G (weak scaling: 256 rows per GPU); rank 0 owns the signal and broadcasts
it over RCCL each step; there is no other collective.

Prints ONE JSON line on rank 0 with the driver's contract fields plus `roofline` (dominant kernel,
HIP-event timed inside this process) and `cpu_baseline` (the oracle timed on this box's host cores on
a bounded row sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
CONFIGS = {
    # name: (mother id, param, precision, label)
    "c2": (0, 6.0, 64, "N=2^20 fp64 Morlet(6) 256 scales"),
    "c3_paul": (1, 4.0, 32, "N=2^20 fp32 Paul(4) 256 scales"),
    "c3_dog": (2, 2.0, 32, "N=2^20 fp32 DOG(2) 256 scales"),
}


def scale_grid(N, dt, flambda, rows):
    s0 = 2 * dt / flambda
    dj = np.log2(N * dt / s0) / (rows - 1)
    return s0 * 2 ** (np.arange(rows) * dj)


def flambda_of(kind, p):
    if kind == 0:
        return 4 * np.pi / (p + np.sqrt(2 + p * p))
    if kind == 1:
        return 4 * np.pi / (2 * p + 1)
    return 2 * np.pi / np.sqrt(p + 0.5)


def cpu_baseline(x, dt, kind, param, sj_all, budget_s=25.0):
    """The oracle (NumPy restatement of wavelet.py:91-106 on pocketfft, 1 thread like the reference)
    timed on this box's host cores on the same workload: rows in groups of 16 (forward FFT + filter bank +
    batched inverse FFT per group, as wavelet.py does for the whole matrix) until all rows are done or
    the time budget is used up."""
    from oracle import cwt_oracle as orc
    m = orc.Mother(kind, int(param) if kind else param)
    orc.cwt_rows(x[:4096], dt, sj_all[:2], m)                 # warm-up (imports, pocketfft plan)
    order = np.random.default_rng(0).permutation(len(sj_all))  # unbiased sample if the budget cuts it short
    t0 = time.perf_counter()
    done = 0
    for g in range(0, len(order), 16):
        idx = np.sort(order[g:g + 16])
        with np.errstate(all="ignore"):
            orc.cwt_rows(x, dt, sj_all[idx], m)
        done += len(idx)
        if time.perf_counter() - t0 > budget_s:
            break
    el = time.perf_counter() - t0
    return {"value": done * x.size / el / 1e9, "unit": "GSamples*scales/s", "cores": 1, "kind": "port",
            "sample": f"{done} of {len(sj_all)} rows (random order, groups of 16) at N={x.size}: forward FFT + "
                      f"filter bank + batched inverse FFT per group, {el:.1f} s; box has {os.cpu_count()} cores, "
                      "1 used (the reference is single-threaded)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--logn", type=int, default=20)
    ap.add_argument("--rows", type=int, default=256, help="rows per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="plan option key=value (tuning)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the collectives even with one rank (smoke test of the RCCL path)")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON result.  Libraries that write to the C stdout stream (RCCL prints
    # a version banner there on its first collective) are sent to stderr for the duration of the run.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from pycwt_amd import _hip

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    from pycwt_amd import _build
    _build.ensure(local)          # prebuilt library travels with the tree; compile once if it did not
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    kind, param, prec, label = CONFIGS[args.config]
    N = 1 << args.logn
    dt = 1.0
    rows_local = args.rows
    rows_total = rows_local * world
    sj_all = scale_grid(N, dt, flambda_of(kind, param), rows_total)
    sj = np.ascontiguousarray(sj_all[rank::world])
    real_t = torch.float64 if prec == 64 else torch.float32
    cplx_t = torch.complex128 if prec == 64 else torch.complex64
    csize = 16 if prec == 64 else 8

    x_host = np.random.default_rng(1234).standard_normal(N)
    x = torch.empty(N, dtype=real_t, device=dev)
    if rank == 0:
        x.copy_(torch.from_numpy(x_host).to(real_t))
    xhat = torch.empty(N, dtype=cplx_t, device=dev)
    W = torch.empty((rows_local, N), dtype=cplx_t, device=dev)

    opts = {k: int(v) for k, v in (o.split("=") for o in args.opt)}
    plan = _hip.Plan(N, prec, max_rows=rows_local, device=local, options=opts)
    plan.set_stream(torch.cuda.current_stream().cuda_stream)

    # Two signal buffers: with more than one rank the broadcast of step i+1 is issued (async, on RCCL's own
    # stream) before the kernels of step i are queued, so it travels over xGMI while step i computes.  Every
    # timed step still owns exactly one broadcast: the pipeline is drained at the boundaries of the timed
    # region (no broadcast is issued ahead of its start, none is prefetched past its end).
    xbuf = [x, x.clone()]

    def compute(buf):
        plan.forward_fft(buf.data_ptr(), N, xhat.data_ptr())
        plan.transform_rows(xhat.data_ptr(), kind, param, dt, sj, W.data_ptr(), N, N)

    def run_steps(count):
        pending = dist.broadcast(xbuf[0], src=0, async_op=True) if use_dist else None
        for i in range(count):
            if use_dist:
                pending.wait()                                       # current stream waits for broadcast i
                pending = dist.broadcast(xbuf[(i + 1) & 1], src=0, async_op=True) if i + 1 < count else None
            compute(xbuf[i & 1])

    def step():
        run_steps(1)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    run_steps(args.warmup)
    fence()
    t0 = time.perf_counter()
    run_steps(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    units_per_step = float(N) * rows_total
    value = units_per_step / (elapsed / args.steps) / 1e9

    # ---- per-kernel HIP-event timing (separate passes so the events do not perturb `value`) ----
    plan.set_option("profile", 1)
    plan.set_option("overlap", 0)     # kernels one at a time, so that each duration is its own
    plan.set_option("overlap_narrow", 0)
    prof_steps = max(3, min(10, args.steps))
    step(); fence(); plan.timings()
    for _ in range(prof_steps):
        step()
    fence()
    tm = plan.timings()
    plan.set_option("profile", 0)
    plan.set_option("overlap", opts.get("overlap", 0))
    plan.set_option("overlap_narrow", opts.get("overlap_narrow", 0))
    split = plan.last_split()
    units_by_class = {"small": split["small"] * N, "narrow": (split["narrow"] - split["narrow_k2048"]) * N,
                      "narrow_big": split["narrow_k2048"] * N, "pass_a": split["two_pass"] * N,
                      "pass_b": split["two_pass"] * N}
    kern = {}
    for name, (ms, cnt) in tm.items():
        kern[name] = {"ms_per_step": ms / prof_steps, "launches_per_step": cnt / prof_steps}
    dom = max((k for k in kern if k in units_by_class), key=lambda k: kern[k]["ms_per_step"])
    dom_units = units_by_class[dom]
    dom_launches = kern[dom]["launches_per_step"]
    dom_avg_ms = kern[dom]["ms_per_step"] / dom_launches
    alg_bytes_per_launch = dom_units * csize / dom_launches      # 16 B (8 B) per sample*scale
    achieved = alg_bytes_per_launch / (dom_avg_ms * 1e-3) / 1e9
    gpu_ms = sum(v["ms_per_step"] for v in kern.values())
    alg_bytes_total = units_per_step / world * csize + N * (csize // 2)
    traffic = None                      # HBM bytes per launch of the dominant kernel, from committed PMC passes
    tpath = os.path.join(ROOT, "profiles", f"traffic_{args.config}.json")
    if os.path.exists(tpath) and not opts and args.logn == 20 and args.rows == 256:
        t = json.load(open(tpath))["per_kernel_class"].get(dom)
        if t:
            traffic = t["hbm_bytes_per_launch"]
    roofline = {
        "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
        "traffic_source": (f"profiles/traffic_{args.config}.json (rocprofv3 PMC passes of this command, "
                           "FETCH_SIZE x2 + WRITE_SIZE)") if traffic else None,
        "avg_launch_ms": dom_avg_ms, "launches_per_step": dom_launches,
        "algorithmic_bytes_per_launch": alg_bytes_per_launch,
        # bytes the memory system really moved per launch (PMC) / live launch time: how busy HBM + fabric are,
        # as opposed to `frac`, which prices only the algorithmic bytes
        "traffic_GBs": (traffic / (dom_avg_ms * 1e-3) / 1e9) if traffic else None,
        "traffic_frac": (traffic / (dom_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
        "whole_path": {"algorithmic_bytes_per_step_per_gpu": alg_bytes_total,
                       "kernel_ms_per_step": gpu_ms,
                       "achieved_GBs": alg_bytes_total / (gpu_ms * 1e-3) / 1e9,
                       "frac": alg_bytes_total / (gpu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "frac_of_measured_copy_ceiling_6290": alg_bytes_total / (gpu_ms * 1e-3) / 1e9 / 6290.0},
        "kernels": kern, "row_split": split,
    }

    out = {
        "metric": "CWT GSamples*scales/s at N=2^20, J=256", "value": value, "unit": "GSamples*scales/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64" if prec == 64 else "f32", "data": "synthetic",
        "config": {"workload": label + (f" x{world} GPUs ({rows_total} rows, row j -> rank j mod {world})" if world > 1 else ""),
                   "N": N, "rows_per_gpu": rows_local, "rows_total": rows_total, "mother": ["morlet", "paul", "dog"][kind],
                   "param": param, "signal": "default_rng(1234).standard_normal(N)", "dt": dt,
                   "parallelism": f"scale-sharded x{world}, 1 broadcast/step" if world > 1 else "single GPU",
                   "plan_options": opts},
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(x_host, dt, kind, param, sj_all)
    plan.close()
    if use_dist:
        dist.destroy_process_group()
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)           # drain what C libraries buffered while fd 1 pointed at stderr
    except OSError:
        pass
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
