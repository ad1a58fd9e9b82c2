#!/usr/bin/env python3
"""bench.py -- throughput of the CWT hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3_paul|c3_dog]

One "step" = one pass of the hot path over one synthetic signal: (broadcast of the signal when N > 1) -> forward
FFT -> all rows of W written device-resident.  Inputs are resident in HBM when the timed region starts.

Workload at 1 GPU = BASELINE.json configs[1]: N = 2^20 fp64 samples, Morlet(6), 256 scales spanning
s0 = 2dt/flambda .. N*dt.  With G GPUs (one process per GPU, launched by torch.distributed.run) the
SAME 256 rows are split over the ranks -- contiguous runs of scales of equal estimated cost (`--partition balanced`,
default) or row j -> rank j mod G (`--partition interleaved`) -- (STRONG scaling, the quantity north_star's ">= 6x at
8 GPUs" is about); rank 0 owns the signal and broadcasts it over RCCL each step, the broadcast of step i+1
travelling while step i computes; there is no other collective.  `--weak` (and the `weak_scaling` block of the
default multi-GPU line) refines the grid to 256*G rows instead, 256 per GPU.

Clocks: this GPU idles at sclk ~0.5 GHz and needs ~45 ms of work to reach its sustained clock (tools/clock_ramp.py), more
than the W warm-up steps of a short run last.  bench.py therefore runs an untimed priming phase (>= 80 ms of the same
steps) BEFORE the W warm-up steps; W and K are honoured exactly.  The same W + K steps started from the idle device are
measured first and reported as `from_idle`; `--no-prime` makes that the headline.

Rank 0 prints ONE JSON line with the driver's contract fields plus
  roofline      dominant kernel, HIP-event timed inside this process in a separate pass
  parity        (1 GPU) every row of W of the timed workload against the CPU oracle: max per-row error, worst
                row, worst row per kernel class
  cpu_baseline  (1 GPU) the oracle timed on this box's host cores on the same rows (kind "port") and the whole
                reference function incl. its NaN scan + copy on a 64-row subset (`reference_as_is`)
  extra         (1 GPU, default config) BASELINE config 3 -- fp32 Paul(4) and DOG(2) -- measured the same way

`--emulate --backend gloo` runs the whole script on CPU against the emulated kernel library (tests/emu) with a small
transform: a rehearsal of the launch / environment / stdout contract, not a measurement.
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
    "c2": (0, 6.0, 64, "fp64 Morlet(6)"),
    "c3_paul": (1, 4.0, 32, "fp32 Paul(4)"),
    "c3_dog": (2, 2.0, 32, "fp32 DOG(2)"),
    # not BASELINE configs: the other mothers in double precision (python bench.py --config paul64 / dog64)
    "paul64": (1, 4.0, 64, "fp64 Paul(4)"),
    "dog64": (2, 2.0, 64, "fp64 DOG(2)"),
}
PARITY_TOL = {64: 1e-8, 32: 1e-5}       # per-row max|dW|/max|W|: 1/100 of north_star's bars (1e-6 / 1e-3)
# Accuracy target of the timed plans (cwt_plan_set_tolerance).  The engine's own default is round-off (1e-16 / 1e-7: safe
# for any input); its truncations are relative to the filter, and the synthetic workload of the metric is white noise
# (spectral dynamic range max|xhat|/rms|xhat| ~ 4), for which these targets leave every row >= 100x inside north_star's
# bars (parity block: all 256 rows).  `extra.c2_roundoff` times the same workload at the engine's default.
BENCH_TOLERANCE = {64: 1e-9, 32: 3e-5}


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


class Runtime:
    """Device / process-group plumbing: a real GPU + RCCL, or (rehearsal) CPU tensors + gloo + emulated kernels."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        from pycwt_amd import _hip
        self.torch, self.dist = torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}: launch with torch.distributed.run")
        self.emulate = args.emulate
        if self.emulate:
            sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
            import build_emu
            self.lib = _hip.Library(build_emu.build())
            self.dev = torch.device("cpu")
            self.device_index = 0
        else:
            from pycwt_amd import _build
            _build.ensure(self.local)     # prebuilt library travels with the tree; compile once if it did not
            self.lib = _hip.Library(os.path.abspath(args.lib)) if args.lib else _hip.load()
            torch.cuda.set_device(self.local)
            self.dev = torch.device("cuda", self.local)
            self.device_index = self.local
        self.shard = (self.rank, self.world)              # (R, G): this process computes rows j = R mod G
        if args.shard:
            r, g = (int(v) for v in args.shard.split("/"))
            if self.world != 1 or not 0 <= r < g:
                raise SystemExit("--shard R/G is a single-process diagnostic with 0 <= R < G")
            self.shard = (r, g)
        self.backend = args.backend or ("gloo" if self.emulate else "nccl")
        self.use_dist = self.world > 1 or args.force_dist
        if self.use_dist:
            if self.world == 1:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29517")
                os.environ.setdefault("RANK", "0")
                os.environ.setdefault("WORLD_SIZE", "1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group(self.backend)

    def stream_handle(self):
        return 0 if self.emulate else self.torch.cuda.current_stream().cuda_stream

    def sync(self):
        if not self.emulate:
            self.torch.cuda.synchronize()

    def fence(self):
        self.sync()
        if self.use_dist:
            self.dist.barrier()
            self.sync()

    def max_over_ranks(self, seconds):
        if not self.use_dist:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.use_dist:
            self.dist.destroy_process_group()


class Workload:
    """One BASELINE configuration on this rank: the signal, this rank's rows (j = rank mod world) of a
    `rows_total`-row scale grid, the device buffers and the plan."""

    def __init__(self, rt, config, logn, rows_total, opts, partition="balanced", pipeline=1, signal="white", auto_target=None):
        from pycwt_amd import _hip
        torch = rt.torch
        self.rt, self.config = rt, config
        self.kind, self.param, self.prec, self.label = CONFIGS[config]
        self.N, self.dt, self.rows_total = 1 << logn, 1.0, rows_total
        self.sj_all = scale_grid(self.N, self.dt, flambda_of(self.kind, self.param), rows_total)
        self.opts = dict(opts)
        self.opts.setdefault("tolerance", BENCH_TOLERANCE[self.prec])
        self.plan = _hip.Plan(self.N, self.prec, max_rows=rows_total, device=rt.device_index, lib=rt.lib,
                              options=self.opts)
        if rt.shard[1] > 1 and partition == "balanced":
            # contiguous shards of equal estimated cost (pycwt_amd.parallel.balanced_shards): every rank classifies the
            # whole grid -- host arithmetic, identical on all ranks -- and takes its run of scales
            from pycwt_amd.parallel import balanced_shards
            labels = self.plan.classify(self.kind, self.param, self.dt, self.sj_all, self.N, True)
            self.mine = balanced_shards(labels, rt.shard[1], self.prec, self.N, lib=rt.lib)[rt.shard[0]]
        else:
            self.mine = np.arange(rt.shard[0], rows_total, rt.shard[1])
        self.sj = np.ascontiguousarray(self.sj_all[self.mine])
        real_t = torch.float64 if self.prec == 64 else torch.float32
        cplx_t = torch.complex128 if self.prec == 64 else torch.complex64
        self.csize = 16 if self.prec == 64 else 8
        self.x_host = np.random.default_rng(1234).standard_normal(self.N)
        self.signal = signal
        if signal == "red":
            # AR(1) noise, lag-1 autocorrelation 0.99 (the colour of geophysical series: the reference's own NINO3 / SOI samples
            # are red), unit variance: spectrum ~ 1 / (1 - 2 g cos w + g^2), 4e4 in power between the ends
            from scipy.signal import lfilter
            y = lfilter([1.0], [1.0, -0.99], self.x_host)
            self.x_host = y / y.std()
        if self.prec == 32:
            self.x_host = self.x_host.astype(np.float32)
        x = torch.empty(self.N, dtype=real_t, device=rt.dev)
        if rt.rank == 0:
            x.copy_(torch.from_numpy(self.x_host))
        # two signal buffers: the broadcast of step i+1 lands in the other one while step i computes
        self.xbuf = [x, x.clone()]
        self.xhat = torch.empty(self.N, dtype=cplx_t, device=rt.dev)
        self.W = torch.empty((max(len(self.sj), 1), self.N), dtype=cplx_t, device=rt.dev)
        self.plan.set_stream(rt.stream_handle())
        # --pipeline P > 1 (diagnostic): P signals in flight, step i on lane i mod P = its own plan, stream and W
        self.lanes = [(self.plan, None, self.W, self.xhat)]
        for _ in range(1, pipeline):
            pl = _hip.Plan(self.N, self.prec, max_rows=rows_total, device=rt.device_index, lib=rt.lib, options=self.opts)
            st = torch.cuda.Stream(device=rt.dev)
            pl.set_stream(st.cuda_stream)
            self.lanes.append((pl, st, torch.empty_like(self.W), torch.empty_like(self.xhat)))
        if auto_target:
            # what pycwt_amd.cwt runs by default: the tolerance that holds `auto_target` relative to every row's peak for THIS
            # signal's spectrum (cwt_plan_auto_tolerance: the dynamic range is measured on the device)
            self.plan.forward_fft(self.xbuf[0].data_ptr(), self.N, self.xhat.data_ptr())
            self.auto_tolerance = self.plan.auto_tolerance(self.xhat.data_ptr(), auto_target)
            self.plan.set_tolerance(self.auto_tolerance)
        self.tolerance = self.plan.tolerance()
        self.sharded = rt.shard[1] > 1

    def compute(self, buf, i=0):
        if len(self.sj):   # forward FFT + every row of W in one call (wavelet.py:91-106); a rank of a sharded transform
            # has no use for the spectrum itself (None: kept in plan scratch, skipped if none of its rows needs it)
            plan, _, W, xhat = self.lanes[i % len(self.lanes)]
            plan.transform(buf.data_ptr(), self.N, self.kind, self.param, self.dt, self.sj,
                           None if self.sharded else xhat.data_ptr(), W.data_ptr(), self.N, self.N)

    def run_steps(self, count):
        """`count` steps.  With more than one rank the broadcast of step i+1 is issued (async, on RCCL's own
        stream) before the kernels of step i are queued; `wait()` makes the compute stream -- not the host --
        wait for the broadcast, so host enqueue, xGMI transfer and kernels all overlap.  Every timed step owns
        exactly one broadcast: none is issued ahead of the region's start, none is prefetched past its end."""
        rt = self.rt
        pending = rt.dist.broadcast(self.xbuf[0], src=0, async_op=True) if rt.use_dist else None
        for i in range(count):
            if rt.use_dist:
                pending.wait()
                pending = rt.dist.broadcast(self.xbuf[(i + 1) & 1], src=0, async_op=True) if i + 1 < count else None
            self.compute(self.xbuf[i & 1], i)

    def timed(self, steps, warmup):
        rt = self.rt
        self.run_steps(warmup)
        rt.fence()
        t0 = time.perf_counter()
        self.run_steps(steps)
        self.host_enqueue_ms = (time.perf_counter() - t0) / steps * 1e3      # host time to QUEUE a step (diagnostic)
        rt.fence()
        elapsed = rt.max_over_ranks(time.perf_counter() - t0)
        ms = elapsed / steps * 1e3
        return {"ms_per_step": ms, "value": float(self.N) * self.rows_total / (elapsed / steps) / 1e9,
                "host_enqueue_ms_per_step": self.host_enqueue_ms}

    def api_timed(self, steps, warmup):
        """The same K steps through the PUBLIC multi-GPU entry point, `pycwt_amd.parallel.cwt_sharded` (shape known to every
        rank, `assume_finite=True`, the signal already on rank 0's device): one broadcast per call, the scale grid, the shards
        and the output tensor made inside the call -- what a caller of the API pays, beside the bench's own loop above (which
        double-buffers the broadcast of step i + 1 under step i)."""
        import pycwt_amd
        from pycwt_amd import parallel
        rt = self.rt
        mother = {0: pycwt_amd.Morlet, 1: pycwt_amd.Paul, 2: pycwt_amd.DOG}[self.kind](self.param)
        s0 = 2 * self.dt / mother.flambda()
        dj = np.log2(self.N * self.dt / s0) / (self.rows_total - 1)
        eng = parallel.HipEngine(self.N, self.prec, self.rows_total, rt.device_index, on_torch_stream=not rt.emulate,
                                 options=self.opts, lib=rt.lib)
        x = self.xbuf[0] if rt.rank == 0 else None
        group = None
        rows_seen = [0]

        def run(count):
            for _ in range(count):
                W, mine, *_ = parallel.cwt_sharded(x, self.dt, dj, s0, self.rows_total - 1, mother, precision=self.prec,
                                                   device=rt.dev, engine=eng, shape=(self.N,), assume_finite=True, group=group)
                rows_seen[0] = len(mine)
        run(max(1, warmup))
        rt.fence()
        t0 = time.perf_counter()
        run(steps)
        host = (time.perf_counter() - t0) / steps * 1e3
        rt.fence()
        elapsed = rt.max_over_ranks(time.perf_counter() - t0)
        eng.plan.close()
        return {"ms_per_step": elapsed / steps * 1e3, "value": float(self.N) * self.rows_total / (elapsed / steps) / 1e9,
                "host_ms_per_call": host, "rows_this_rank": rows_seen[0], "collectives_per_call": 1 if rt.use_dist else 0,
                "entry_point": "pycwt_amd.parallel.cwt_sharded(x_dev, ..., shape=(N,), assume_finite=True)"}

    def prime(self, min_ms):
        """Untimed steps until at least `min_ms` of wall time have passed: brings the device from its idle clock to its
        sustained clock (DPM ramps over the first ~45 ms of work; measured with tools/clock_ramp.py: 1.23 ms per step
        for the first 10 steps after idle, 1.09 for the next 10, 1.03 from the 40th on).  Returns the steps run."""
        rt, done = self.rt, 0
        rt.fence()
        t0 = time.perf_counter()
        while True:
            self.run_steps(8)
            rt.fence()
            done += 8
            spent = rt.max_over_ranks(time.perf_counter() - t0) * 1e3
            if spent >= min_ms or done >= 4096:
                return done

    def roofline(self, steps, traffic=None):
        """Per-kernel HIP-event timing in separate passes (option "profile": every kernel alone on the plan's
        stream, so that each duration is its own) -> achieved algorithmic bandwidth of every kernel class and of the
        whole path; `frac` etc. describe the class with the largest total time (the "dominant kernel")."""
        plan, rt, N = self.plan, self.rt, self.N
        plan.set_option("profile", 1)
        prof_steps = max(3, min(10, steps))
        self.run_steps(1); rt.fence(); plan.timings()
        for _ in range(prof_steps):
            self.run_steps(1)
        rt.fence()
        tm = plan.timings()
        plan.set_option("profile", 0)
        split = plan.last_split()
        labels = plan.row_classes()
        kern = {name: {"ms_per_step": ms / prof_steps, "launches_per_step": cnt / prof_steps}
                for name, (ms, cnt) in tm.items()}

        def terms(label):
            return int(label.rsplit("/t", 1)[1]) if "/t" in label else 1
        # row class -> (kernels that compute it, rows); the overlap-save rows also own the block spectra (ols_fwd)
        groups = {
            "single_wg": (["small", "direct"], [c for c in labels if c == "single_wg"]),
            "narrow": (["narrow"], [c for c in labels if c.startswith("narrow/") and terms(c) <= 4]),
            "narrow_many": (["narrow_many"], [c for c in labels if c.startswith("narrow/") and terms(c) > 4]),
            "narrow_big": (["narrow_big"], [c for c in labels if c.startswith("narrow_k2048")]),
            "two_pass": (["pass_a", "pass_b"], [c for c in labels if c.startswith("two_pass")]),
            "ols_small": (["ols_small"], [c for c in labels if c.startswith("ols") and c.endswith("/half")]),
            "ols": (["ols"], [c for c in labels if c.startswith("ols") and not c.endswith("/half")]),
            # rows clipped at Nyquist: overlap-save on the band-passed complex signal (aols_pre = that signal + its block spectra)
            "aols": (["aols"], [c for c in labels if c.startswith("aols")]),
            # band-limited rows in polynomial form (poly_coef = their interval coefficients, shared)
            "poly": (["poly"], [c for c in labels if c.startswith("poly")]),
        }
        traffic_file, traffic_tab = None, {}
        tpath = os.path.join(ROOT, "profiles", f"traffic_{self.config}.json")
        default_opts = set(self.opts) <= {"tolerance"} and self.opts.get("tolerance") == BENCH_TOLERANCE[self.prec]
        if traffic:
            traffic_file, traffic_tab = traffic["source"], traffic["per_kernel_class"]
        elif os.path.exists(tpath) and default_opts and N == 1 << 20 and self.rows_total == 256 and rt.shard == (0, 1):
            traffic_file = f"builder_profile: profiles/traffic_{self.config}.json (committed PMC passes of the same command)"
            traffic_tab = json.load(open(tpath))["per_kernel_class"]
        per_class = {}
        for name, (kernels, rows) in groups.items():
            ms = sum(kern[k]["ms_per_step"] for k in kernels if k in kern)
            if not rows or not ms:
                continue
            launches = sum(kern[k]["launches_per_step"] for k in kernels if k in kern)
            alg = float(len(rows)) * N * self.csize
            t = [traffic_tab[k]["hbm_bytes_per_launch"] * kern[k]["launches_per_step"] for k in kernels
                 if k in kern and k in traffic_tab]
            per_class[name] = {"rows": len(rows), "kernels": [k for k in kernels if k in kern], "ms_per_step": ms,
                               "launches_per_step": launches, "us_per_row": ms * 1e3 / len(rows),
                               "achieved_GBs": alg / (ms * 1e-3) / 1e9, "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "traffic_ratio": (sum(t) / alg) if len(t) == len([k for k in kernels if k in kern]) and t else None}
        shared = {k: kern[k]["ms_per_step"] for k in ("fwd_small", "fwd_pass_a", "fwd_pass_b", "ols_fwd", "aols_pre", "poly_coef") if k in kern}
        if not per_class:
            return {"bound": "hbm", "kernel": None, "kernels": kern, "row_split": split}
        dom = max(per_class, key=lambda k: per_class[k]["ms_per_step"])
        d = per_class[dom]
        dom_kernel = max(d["kernels"], key=lambda k: kern[k]["ms_per_step"])
        dom_launches = kern[dom_kernel]["launches_per_step"]
        dom_avg_ms = kern[dom_kernel]["ms_per_step"] / dom_launches
        alg_bytes_per_launch = d["rows"] * float(N) * self.csize / dom_launches      # 16 B (8 B) per unit
        achieved = alg_bytes_per_launch / (dom_avg_ms * 1e-3) / 1e9
        gpu_ms = sum(v["ms_per_step"] for v in kern.values())
        alg_bytes_total = float(N) * len(self.sj) * self.csize + N * (self.csize // 2)
        traffic_cal = traffic.get("calibration") if isinstance(traffic, dict) else None
        traffic = traffic_tab.get(dom_kernel, {}).get("hbm_bytes_per_launch")
        whole = {"algorithmic_bytes_per_step_per_gpu": alg_bytes_total, "kernel_ms_per_step": gpu_ms,
                 "achieved_GBs": alg_bytes_total / (gpu_ms * 1e-3) / 1e9,
                 "frac": alg_bytes_total / (gpu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "frac_of_measured_copy_ceiling_6290": alg_bytes_total / (gpu_ms * 1e-3) / 1e9 / 6290.0}
        return {
            "bound": "hbm", "kernel": dom_kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            # the honest headline: ALL algorithmic bytes of the step over the timed step (set by measure(); here: over the
            # sum of all kernel times of the profiling pass)
            "whole_path_frac": whole["frac"],
            "traffic": traffic,
            # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, separate passes): measured inside
            # this run by live_traffic() where rocprofv3 is available, else the builder's committed passes of the same command
            "traffic_source": traffic_file if traffic else None,
            "traffic_calibration": traffic_cal,
            "avg_launch_ms": dom_avg_ms, "launches_per_step": dom_launches,
            "algorithmic_bytes_per_launch": alg_bytes_per_launch,
            "traffic_GBs": (traffic / (dom_avg_ms * 1e-3) / 1e9) if traffic else None,
            "traffic_frac": (traffic / (dom_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "whole_path": whole,
            # every row class: rows, us per row, fraction of 8 TB/s on its algorithmic bytes, PMC traffic / algorithmic
            "per_class": per_class, "shared_kernels_ms_per_step": shared,
            "kernels": kern, "row_split": split,
        }

    def cold_grid(self, reps=3):
        """ms of a step whose scale grid the plan has not seen: classification of the rows on the host (incl. the halo
        class dynamic programme), upload of the row table, the filter tables of the overlap-save rows (k_ols_gtab) and the
        step itself -- what the timed loop's warm-up pays once and the cached row table saves afterwards."""
        rt, keep = self.rt, self.sj
        out = []
        for r in range(reps):
            self.sj = np.ascontiguousarray(keep * (1.0 + (r + 1) * 1e-13))
            rt.fence()
            t0 = time.perf_counter()
            self.compute(self.xbuf[0])
            rt.fence()
            out.append((time.perf_counter() - t0) * 1e3)
        self.sj = keep
        self.compute(self.xbuf[0])
        rt.fence()
        return float(np.median(out))

    def cpu_and_parity(self, budget_s=25.0, group=16):
        """The oracle (NumPy restatement of wavelet.py:91-106 on pocketfft, 1 thread like the reference) computes the
        rows of the SAME workload in groups of `group` (forward FFT + filter bank + batched inverse FFT per group,
        as wavelet.py does for the whole matrix).  Its time is the CPU baseline; its rows are the parity reference
        for the rows of W that the last GPU step left on the device (downloaded group by group, outside the oracle's
        clock).  Rows the reference itself turns into NaN (Paul, wavelet.py:111-115) are timed but not compared."""
        from oracle import cwt_oracle as orc
        m = orc.Mother(self.kind, int(self.param) if self.kind else self.param)
        x, dt, N = self.x_host, self.dt, self.N
        orc.cwt_rows(x[:4096], dt, self.sj_all[:2], m)                 # warm-up (imports, pocketfft plan)
        classes = self.plan.row_classes()
        order = np.random.default_rng(0).permutation(len(self.sj))  # unbiased sample if the budget cuts it short
        dropped = orc.dropped_rows(self.sj, dt, m)
        cpu_s, done, checked = 0.0, 0, 0
        worst = (0.0, -1)
        per_class = {}
        l2_num = l2_den = 0.0
        for g in range(0, len(order), group):
            idx = np.sort(order[g:g + group])
            t0 = time.perf_counter()
            with np.errstate(all="ignore"):
                ref = orc.cwt_rows(x, dt, self.sj[idx], m)
            cpu_s += time.perf_counter() - t0
            done += len(idx)
            got = self.W[self.rt.torch.from_numpy(idx)].cpu().numpy()
            for k, j in enumerate(idx):
                if dropped[j]:
                    continue
                den = np.abs(ref[k]).max()
                err = float(np.abs(got[k] - ref[k]).max() / (den if den > 0 else 1.0))
                checked += 1
                l2_num += float(np.sum(np.abs(got[k] - ref[k]) ** 2))
                l2_den += float(np.sum(np.abs(ref[k]) ** 2))
                c = per_class.setdefault(classes[j], {"rows": 0, "max_row_err": 0.0, "worst_row": -1})
                c["rows"] += 1
                if err >= c["max_row_err"]:
                    c["max_row_err"], c["worst_row"] = err, int(self.mine[j])
                if err >= worst[0]:
                    worst = (err, int(self.mine[j]))
            if cpu_s > budget_s:
                break
        tol = PARITY_TOL[self.prec]
        parity = {"rows_checked": checked, "rows_total": int(len(self.sj)), "max_row_err": worst[0],
                  "worst_row": worst[1], "rel_l2": float(np.sqrt(l2_num / l2_den)) if l2_den else None,
                  "tolerance": tol, "ok": bool(checked > 0 and worst[0] < tol),
                  "metric": "per-row max|W_gpu - W_oracle| / max|W_oracle| over all N columns",
                  "per_kernel_class": per_class}
        cpu = {"value": done * float(N) / cpu_s / 1e9, "unit": "GSamples*scales/s", "cores": 1, "kind": "port",
               "sample": f"{done} of {len(self.sj)} rows (random order, groups of {group}) at N={N}: forward FFT + "
                         f"filter bank + batched inverse FFT per group, {cpu_s:.1f} s; box has {os.cpu_count()} "
                         "cores, 1 used (the reference is single-threaded)"}
        return cpu, parity

    def reference_as_is(self, nrows=64):
        """Wall time of the reference's WHOLE function -- incl. the NaN scan and the fancy-index copy of
        wavelet.py:111-115 that the row-wise port skips -- on every (rows/nrows)-th row of the workload, via the
        `freqs=` argument (wavelet.py:86-88): the unmodified reference when /root/reference is importable (build
        container), else its restatement oracle.cwt."""
        from oracle import cwt_oracle as orc
        m = orc.Mother(self.kind, int(self.param) if self.kind else self.param)
        sel = self.sj_all[::max(1, len(self.sj_all) // nrows)][:nrows]
        freqs = 1.0 / (m.flambda() * sel)
        fn, kind = None, "port"
        ref_root = "/root/reference"
        if os.path.isdir(os.path.join(ref_root, "pycwt")):
            try:
                sys.dont_write_bytecode = True
                sys.path.insert(0, ref_root)
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    import pycwt as ref
                mother = {0: ref.Morlet, 1: ref.Paul, 2: ref.DOG}[self.kind](int(self.param) if self.kind else self.param)
                fn, kind = (lambda: ref.cwt(self.x_host, self.dt, wavelet=mother, freqs=freqs)), "reference"
            except Exception:
                fn = None
            finally:
                sys.path.remove(ref_root)
        if fn is None:
            fn = lambda: orc.cwt(self.x_host, self.dt, wavelet=m, freqs=freqs)          # noqa: E731
        t0 = time.perf_counter()
        with np.errstate(all="ignore"):
            out = fn()
        el = time.perf_counter() - t0
        return {"value": len(sel) * float(self.N) / el / 1e9, "unit": "GSamples*scales/s", "cores": 1, "kind": kind,
                "reference_mounted": kind == "reference", "rows_returned": int(out[0].shape[0]),
                "sample": f"the whole cwt() (FFT, filter bank, batched inverse FFT, NaN scan, row copy) on {len(sel)} of "
                          f"the {len(self.sj_all)} rows at N={self.N}, once: {el:.1f} s"}

    def icwt_pass(self, reps=10):
        """BASELINE.md section 3 "icwt: report": the eq.-11 reduction (wavelet.py:169-170) over the device-resident W the
        last step left, as a standalone pass (k_icwt reads every element of W once: 16 B / 8 B per sample*scale)."""
        torch, rt = self.rt.torch, self.rt
        out = torch.empty(self.N, dtype=torch.float64 if self.prec == 64 else torch.float32, device=rt.dev)
        self.plan.icwt_reduce(self.W.data_ptr(), self.N, self.N, self.sj, 1.0, out.data_ptr())
        rt.fence()
        t0 = time.perf_counter()
        for _ in range(reps):
            self.plan.icwt_reduce(self.W.data_ptr(), self.N, self.N, self.sj, 1.0, out.data_ptr())
        rt.fence()
        ms = (time.perf_counter() - t0) / reps * 1e3
        nbytes = float(len(self.sj)) * self.N * self.csize
        return {"ms": ms, "algorithmic_bytes": nbytes, "achieved_GBs": nbytes / (ms * 1e-3) / 1e9,
                "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "bound": "hbm (read of W)",
                "note": "standalone pass over W; fusing it into the row kernels was measured and rejected (EXPERIMENTS.md)"}

    def close(self):
        for lane in self.lanes:
            lane[0].close()


def config1_latency(calls=300):
    """BASELINE config 1: the reference's canonical call (sample/simple_sample.py:58-60: 504 points, dt = 0.25, dj = 1/12,
    s0 = 0.5, J = 84, Morlet) through the drop-in pycwt_amd.cwt -- NumPy in, NumPy out, PCIe and launch latency included;
    synthetic 504-point series (the NINO3 file is not on the GPU box; parity on the real data: tests/golden/nino3_*)."""
    import pycwt_amd
    from oracle import cwt_oracle as orc
    x = np.random.default_rng(1234).standard_normal(504)
    for _ in range(20):
        pycwt_amd.cwt(x, 0.25, 1 / 12, 0.5, 84, "morlet")
    ts = []
    for _ in range(calls):
        t0 = time.perf_counter()
        out = pycwt_amd.cwt(x, 0.25, 1 / 12, 0.5, 84, "morlet")
        ts.append(time.perf_counter() - t0)
    ref = orc.cwt(x, 0.25, 1 / 12, 0.5, 84, orc.Mother(orc.MORLET, 6))
    t0 = time.perf_counter()
    for _ in range(20):
        orc.cwt(x, 0.25, 1 / 12, 0.5, 84, orc.Mother(orc.MORLET, 6))
    cpu_ms = (time.perf_counter() - t0) / 20 * 1e3
    err = float((np.abs(out[0] - ref[0]).max(axis=1) / np.abs(ref[0]).max(axis=1)).max())
    return {"workload": "pycwt_amd.cwt(x[504], 0.25, 1/12, 0.5, 84, 'morlet') -> 85 x 504 (BASELINE config 1)", "ms_per_call_median": float(np.median(ts) * 1e3),
            "ms_per_call_min": float(np.min(ts) * 1e3), "oracle_ms_per_call_1_core": cpu_ms, "max_row_err": err, "calls": calls,
            "includes": "host scale grid, H2D of the signal, kernels, D2H of W and the spectrum"}


def config5_callers(logn=20, dj=0.25):
    """BASELINE config 5 without its Monte-Carlo loop, on ONE GPU: pycwt_amd.xwt and pycwt_amd.wct (sig=False) of two
    N = 2^20 series, NumPy in, NumPy out -- two transforms, the smoothing of wavelet.py:499-514 on the device, and the
    download of the result matrices, which is most of the time (PCIe).  Parity: tests/test_gpu_parity.py::
    test_config5_deterministic_part_at_full_size, test_callers_against_reference_fixture."""
    import pycwt_amd
    n = 1 << logn
    rng = np.random.default_rng(55)
    e = rng.standard_normal(n)
    y1 = e + np.sin(2 * np.pi * np.arange(n) / 500.0)
    y2 = 0.5 * np.roll(e, 3) + rng.standard_normal(n) + np.sin(2 * np.pi * np.arange(n) / 500.0 + 0.7)

    def best(f, reps=3):
        out = f()
        shape = out[0].shape
        del out
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = f()
            ts.append(time.perf_counter() - t0)
            del out
        return min(ts) * 1e3, shape
    def best_device(f, reps=3):
        ts = []
        for r in range(reps + 1):
            t0 = time.perf_counter()
            out = f()
            ts.append(time.perf_counter() - t0)
            (out[0] if isinstance(out, tuple) else out).close()
        return min(ts[1:]) * 1e3

    def mc_draw_ms(rng_name, draws=6):
        """ms per Monte-Carlo draw of wct_significance on the surrogate length of this grid (wavelet.py:609-630)."""
        m = pycwt_amd.Morlet(6)
        s0 = 2 * 1.0 / m.flambda()
        J = int(np.round(np.log2(n * 1.0 / s0) / dj))
        kw = dict(progress=False, cache=False, rng=rng_name)
        np.random.seed(3)
        def call(count):
            t0 = time.perf_counter()
            pycwt_amd.wct_significance(0.5, 0.4, 1.0, dj, s0, J, mc_count=count, **kw)
            return time.perf_counter() - t0
        call(2)                                                                             # plans, row tables
        # a call = a fixed part (tens of GB of scratch allocated at its first draw, freed at its end; the first draw's look at the
        # spectra) + draws: the difference of two calls is the draws alone (tests/perf/wct_bench.py prints both parts)
        t2, tm = call(2), call(draws + 2)
        return (tm - t2) / draws * 1e3
    try:
        x_ms, shape = best(lambda: pycwt_amd.xwt(y1, y2, 1.0, dj))
        w_ms, _ = best(lambda: pycwt_amd.wct(y1, y2, 1.0, dj, sig=False))
        xd_ms = best_device(lambda: pycwt_amd.xwt_device(y1, y2, 1.0, dj))
        wd_ms = best_device(lambda: pycwt_amd.wct_device(y1, y2, 1.0, dj))
        mc_np, mc_dev = mc_draw_ms("numpy"), mc_draw_ms("device")
    except Exception as exc:                        # (a box without the memory for the intermediates: report, do not fail the line)
        return {"skipped": f"{type(exc).__name__}: {exc}"[:200]}
    return {"workload": f"xwt and wct(sig=False) of two N=2^{logn} series, dj={dj}: {shape[0]} scales, NumPy in / out (BASELINE config 5 "
                        "without the Monte-Carlo loop, one GPU)", "xwt_ms": x_ms, "wct_ms": w_ms,
            "xwt_device_ms": xd_ms, "wct_device_ms": wd_ms,
            "mc_draw_ms_numpy_surrogates": mc_np, "mc_draw_ms": mc_dev,
            "xwt_host_GBs": shape[0] * n * 16 / (x_ms * 1e-3) / 1e9,
            "includes": "two transforms, smoothing, coherence on the device; xwt / wct: + download of the result matrices (PCIe-bound); "
                        "*_device: results left on the GPU; mc_draw_ms: one Monte-Carlo draw of wct_significance for this grid "
                        "(surrogates of 6 s_max / dt samples, statistical parity only) with surrogates made on the GPU, "
                        "mc_draw_ms_numpy_surrogates: by NumPy on one host thread (the seed-for-seed default)"}


def config4_batch(rt, steps=3):
    """BASELINE config 4 on ONE GPU at full size: 1024 signals x N = 2^16 x 128 Morlet scales through cwt_transform_batch,
    W (137 GB complex128) device resident.  Roofline: 16 B per sample*scale over the timed step."""
    from pycwt_amd import _hip
    torch = rt.torch
    nb, N, rows = 1024, 1 << 16, 128
    sj = scale_grid(N, 1.0, flambda_of(0, 6.0), rows)
    try:
        g = torch.Generator(device=rt.dev)
        g.manual_seed(1234)
        X = torch.randn(nb, N, dtype=torch.float64, device=rt.dev, generator=g)
        xh = torch.empty(nb, N, dtype=torch.complex128, device=rt.dev)
        W = torch.empty(nb, rows, N, dtype=torch.complex128, device=rt.dev)
    except RuntimeError as e:                       # not enough free memory on this device
        return {"skipped": str(e)[:200]}
    plan = _hip.Plan(N, 64, max_rows=nb * rows, device=rt.device_index, lib=rt.lib, options={"tolerance": BENCH_TOLERANCE[64]})
    plan.set_stream(rt.stream_handle())

    def step():
        plan.transform_batch(X.data_ptr(), nb, N, N, 0, 6.0, 1.0, sj, xh.data_ptr(), W.data_ptr(), N, N)
    step(); step(); rt.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    rt.sync()
    ms = (time.perf_counter() - t0) / steps * 1e3
    split = plan.last_split()
    # parity on a few (signal, scale) pairs against the oracle
    from oracle import cwt_oracle as orc
    m = orc.Mother(orc.MORLET, 6)
    worst = 0.0
    for b, j in ((0, 0), (511, 37), (1023, 127), (300, 90), (77, 5)):
        ref = orc.cwt_rows(X[b].cpu().numpy(), 1.0, sj[j:j + 1], m)[0]
        got = W[b, j].cpu().numpy()
        worst = max(worst, float(np.abs(got - ref).max() / np.abs(ref).max()))
    plan.close()
    del X, xh, W
    torch.cuda.empty_cache()
    units = float(nb) * N * rows
    return {"workload": "1024 signals x N=2^16 x 128 Morlet scales, one GPU, cwt_transform_batch (BASELINE config 4)",
            "value": units / (ms * 1e-3) / 1e9, "unit": "GSamples*scales/s", "ms_per_step": ms, "steps": steps,
            "roofline": {"bound": "hbm", "algorithmic_bytes_per_step": units * 16 + nb * N * 8.0,
                         "frac_of_timed_step": (units * 16 + nb * N * 8.0) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "row_split_per_call": split, "sampled_pairs_max_row_err": worst,
            "parity_all_rows": "tests/test_gpu_parity.py::test_config4_full_batch_* (sampled pairs + Parseval on all 131072 rows)"}



def kernel_class(name):
    """rocprofv3 kernel name -> the kernel class names of cwt_plan_timings (the mask pass of the aols rows runs the two-pass
    kernels and is counted with them)."""
    n = name.replace("void cwt::", "").replace("cwt::", "")
    table = (("k_poly_rows", "poly"), ("k_poly_coef", "poly_coef"), ("k_poly_band", "poly_coef"), ("k_aols_rows", "aols"),
             ("k_aols_fwd", "aols_pre"), ("k_ols_fwd", "ols_fwd"), ("k_narrow_ct_big", "narrow_big"),
             ("k_narrow_ct_many", "narrow_many"), ("k_narrow", "narrow"), ("k_icwt", "icwt"), ("k_small", "small"))
    for prefix, cls in table:
        if n.startswith(prefix):
            return cls
    if n.startswith("k_ols_ct"):
        return "ols_small" if n.split("(")[0].rstrip(">").endswith(", 12") else "ols"
    if n.startswith("k_pass_a_ct_rows") or n.startswith("k_pass_a<"):
        return "pass_a"
    if n.startswith("k_pass_a_ct"):
        return "fwd_pass_a"
    if n.startswith("k_pass_b"):
        args = n[n.find("<") + 1:n.find(">")].split(", ")
        return "fwd_pass_b" if args[-1] == "true" else "pass_b"
    return None


def live_traffic(config, logn, rows, n_poly, csize):
    """HBM traffic of every kernel class from PMC counters, measured INSIDE this run: two short child runs of this script
    under `rocprofv3 --kernel-trace --pmc <counter>` (one counter per pass, kernels serialized), as MI355X_MICROARCH.md
    prescribes: FETCH_SIZE and WRITE_SIZE in separate passes, KiB units, FETCH_SIZE doubled on gfx950.  Both counters
    are then CALIBRATED on kernels of this very run whose byte counts are known exactly and whose access pattern is the
    path's own: WRITE_SIZE on k_poly_rows (non-temporal 16-byte stores: n_poly x N x sizeof(complex) bytes written), FETCH_SIZE
    on k_icwt (reads every element of W once).  Returns None when rocprofv3 is not available or a pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    sums, counts = {}, {}
    tmp = tempfile.mkdtemp(prefix="cwt_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [prof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "cwt", "--",
                   sys.executable, os.path.abspath(__file__), "--config", config, "--logn", str(logn), "--rows", str(rows),
                   "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extra", "--no-prime", "--no-live-traffic",
                   "--opt", "overlap_narrow=0", "--opt", "ols_early=0", "--opt", "ols_side=0"]
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=120)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            for f in files:
                for row in csv.DictReader(open(f)):
                    cls = kernel_class(row["Kernel_Name"])
                    if cls and row["Counter_Name"] == counter:
                        sums[(cls, counter)] = sums.get((cls, counter), 0.0) + float(row["Counter_Value"])
                        counts[(cls, counter)] = counts.get((cls, counter), 0) + 1
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    per = {}
    for (cls, counter), tot in sums.items():
        d = per.setdefault(cls, {})
        raw = tot * 1024.0 / counts[(cls, counter)] * (2.0 if counter == "FETCH_SIZE" else 1.0)
        d["fetch_raw" if counter == "FETCH_SIZE" else "write_raw"] = raw
    N = float(1 << logn)
    cal = {"write": None, "fetch": None}
    if n_poly and "poly" in per and per["poly"].get("write_raw"):
        cal["write"] = n_poly * N * csize / per["poly"]["write_raw"]
    if "icwt" in per and per["icwt"].get("fetch_raw"):
        cal["fetch"] = rows * N * csize / per["icwt"]["fetch_raw"]
    fw, ff = cal["write"] or 1.0, cal["fetch"] or 1.0
    for cls, d in per.items():
        d["write_bytes_per_launch"] = d.get("write_raw", 0.0) * fw
        d["fetch_bytes_per_launch"] = d.get("fetch_raw", 0.0) * ff
        d["hbm_bytes_per_launch"] = d["write_bytes_per_launch"] + d["fetch_bytes_per_launch"]
    return {"per_kernel_class": per,
            "calibration": {"WRITE_SIZE_factor": cal["write"], "FETCH_SIZE_x2_factor": cal["fetch"],
                            "on": "k_poly_rows (exact output bytes, nt 16-B stores) / k_icwt (exact input bytes)"},
            "source": "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes, serialized "
                      "kernels, 3 steps each), KiB units, FETCH_SIZE x2 (gfx950), calibrated as stated"}

PRIME_MS = 80.0     # untimed device work before the W warm-up steps, see prime()


def compact_line(out, detail_path):
    """The driver's contract line: the contract fields, `roofline` and `cpu_baseline` as scalars, one `parity` summary and
    ONE number per extra workload -- no per-class tables, no prose.  Everything else is in `detail_path`."""
    def pick(d, keys):
        return {k: d[k] for k in keys if d and k in d}

    def r6(v):
        return float(f"{v:.6g}") if isinstance(v, float) else v
    line = pick(out, ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                      "vs_baseline", "dtype", "data"])
    line["config"] = pick(out["config"], ["workload", "N", "rows_total", "rows_per_gpu", "mother", "param", "tolerance",
                                          "parallelism", "plan_options", "shard_diagnostic", "signals_in_flight"])
    if not line["config"].get("plan_options"):
        line["config"].pop("plan_options", None)
    roof = out.get("roofline") or {}
    line["roofline"] = pick(roof, ["bound", "kernel", "achieved", "peak", "unit", "frac", "whole_path_frac", "traffic",
                                   "avg_launch_ms", "launches_per_step", "algorithmic_bytes_per_launch", "row_split"])
    if roof.get("traffic_source"):
        line["roofline"]["traffic_source"] = "live rocprofv3 PMC passes" if roof["traffic_source"].startswith("measured") else "committed PMC passes"
    if "cpu_baseline" in out:
        cb = out["cpu_baseline"]
        line["cpu_baseline"] = pick(cb, ["value", "unit", "cores", "kind", "reference_mounted"])
        line["cpu_baseline"]["sample"] = cb.get("sample", "")[:96]
        if "reference_as_is" in cb:
            line["cpu_baseline"]["whole_function_value"] = cb["reference_as_is"]["value"]
    if "parity" in out:
        line["parity"] = pick(out["parity"], ["rows_checked", "rows_total", "max_row_err", "worst_row", "tolerance", "ok"])
    if "from_idle" in out:
        line["from_idle"] = pick(out["from_idle"], ["ms_per_step"])
    for k in ("effective_warmup_steps", "cold_grid_ms"):
        if k in out:
            line[k] = out[k]
    if "icwt" in out:
        line["icwt_ms"] = out["icwt"]["ms"]
    if "weak_scaling" in out:
        line["weak_scaling"] = out["weak_scaling"]
    if "build" in out and "id" in out["build"]:
        line["build_id"] = out["build"]["id"]
        line["build_matches_tree"] = out["build"]["matches_tree"]
    if "api" in out:                       # the same steps through pycwt_amd.parallel.cwt_sharded itself
        line["api_ms_per_step"] = out["api"]["ms_per_step"]
        line["api_collectives_per_call"] = out["api"]["collectives_per_call"]
    ex = out.get("extra") or {}
    short = {}
    if "c2_roundoff" in ex:
        short["c2_roundoff_ms"] = ex["c2_roundoff"]["ms_per_step"]
    if "c2_red" in ex and "ms_per_step" in ex["c2_red"]:      # coloured input at the drop-in's automatic tolerance
        short["c2_red_ms"] = ex["c2_red"]["ms_per_step"]
        short["c2_red_tolerance"] = ex["c2_red"]["tolerance"]
        if ex["c2_red"].get("parity"):
            short["c2_red_max_row_err"] = ex["c2_red"]["parity"]["max_row_err"]
    for c in ("c3_paul", "c3_dog", "paul64", "dog64"):
        if c in ex and "value" in ex[c]:
            short[c + "_gs"] = ex[c]["value"]
            short[c + "_ms"] = ex[c]["ms_per_step"]
            if ex[c].get("parity"):
                short[c + "_max_row_err"] = ex[c]["parity"]["max_row_err"]
    if "ms_per_step" in ex.get("c4_batch", {}):
        short["c4_ms"] = ex["c4_batch"]["ms_per_step"]
        short["c4_gs"] = ex["c4_batch"]["value"]
    if "ms_per_call_median" in ex.get("c1_nino3_latency", {}):
        short["c1_ms_per_call"] = ex["c1_nino3_latency"]["ms_per_call_median"]
    for k in ("xwt_ms", "wct_ms", "wct_device_ms", "mc_draw_ms", "mc_draw_ms_numpy_surrogates"):
        if k in ex.get("c5_xwt_wct", {}):
            short["c5_" + k] = ex["c5_xwt_wct"][k]
    if short:
        line["extra"] = short
    line["detail"] = os.path.relpath(detail_path, ROOT) if detail_path else None

    def rnd(o):
        if isinstance(o, dict):
            return {k: rnd(v) for k, v in o.items()}
        if isinstance(o, list):
            return [rnd(v) for v in o]
        return r6(o)
    return rnd(line)


def measure(rt, config, args, rows_total, opts, want_cpu, traffic_passes=True, signal="white", auto_target=None):
    wl = Workload(rt, config, args.logn, rows_total, opts, args.partition, args.pipeline, signal=signal, auto_target=auto_target)
    if args.prime and not args.emulate:
        # the same W + K steps first from an idle device (reported as `from_idle`), then with the clocks up (the headline)
        idle = wl.timed(args.steps, args.warmup)
        primed_steps = wl.prime(PRIME_MS)
        out = wl.timed(args.steps, args.warmup)
        if args.shard:
            # the single-GPU diagnostic of a rank's share: the faster of two timed regions (about one region in fifteen comes
            # out 50 % long on these boxes for no reason visible in the kernel trace; the headline keeps its ONE region)
            again = wl.timed(args.steps, args.warmup)
            if again["ms_per_step"] < out["ms_per_step"]:
                out = again
        out["from_idle"] = {"ms_per_step": idle["ms_per_step"], "value": idle["value"],
                            "note": "the same W warm-up + K timed steps started on an idle device (clocks still ramping)"}
        out["effective_warmup_steps"] = args.warmup + primed_steps + args.warmup + args.steps    # everything that ran before
        # the K timed steps of the headline: the from-idle pass (W + K), the priming steps, the W warm-up steps
        out["clock_priming"] = (f"{primed_steps} untimed steps (>= {PRIME_MS:.0f} ms of work) before the W warm-up steps: this "
                                "GPU needs ~45 ms of work to go from its idle clock (sclk ~0.5 GHz) to its sustained clock "
                                "(tools/clock_ramp.py, profiles/r03_clock_ramp.txt); the K timed steps are unchanged")
    else:
        out = wl.timed(args.steps, args.warmup)
        out["effective_warmup_steps"] = args.warmup
    if rt.use_dist and not args.shard and wl.kind != 1:      # (Paul: the API drops the reference's NaN rows, another workload)
        out["api"] = wl.api_timed(args.steps, args.warmup)
    traffic = None
    if (args.live_traffic and traffic_passes and not args.emulate and not wl.sharded and not rt.use_dist and not args.shard and want_cpu
            and set(wl.opts) <= {"tolerance"} and wl.tolerance == BENCH_TOLERANCE[wl.prec]):
        # (the config-3 blocks of the default run too: round 4 read theirs from committed files)
        split = wl.plan.last_split()
        traffic = live_traffic(config, args.logn, rows_total, split.get("poly", 0), wl.csize)
    if not wl.sharded and not rt.use_dist and not args.emulate:
        wl.run_steps(1)
        rt.fence()
        out["icwt"] = wl.icwt_pass()
    out["roofline"] = wl.roofline(args.steps, traffic)
    wp = out["roofline"].get("whole_path")
    if wp:
        # the headline fraction: this GPU's algorithmic bytes of a step over the TIMED step (wall clock of the K steps);
        # `whole_path.frac` prices the same bytes over the sum of the kernels' own durations in the profiling pass (every
        # kernel alone between two HIP events, which costs a few microseconds per launch)
        wp["frac_of_timed_step"] = wp["algorithmic_bytes_per_step_per_gpu"] / (out["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        out["roofline"]["whole_path_frac"] = wp["frac_of_timed_step"]
    if not wl.sharded and not rt.use_dist:
        out["cold_grid_ms"] = wl.cold_grid()
    if want_cpu:
        wl.run_steps(1)
        rt.fence()
        out["cpu_baseline"], out["parity"] = wl.cpu_and_parity()
        if signal == "white":                 # (the coloured-input block only wants the parity)
            out["cpu_baseline"]["reference_as_is"] = wl.reference_as_is()
            out["cpu_baseline"]["reference_mounted"] = out["cpu_baseline"]["reference_as_is"]["reference_mounted"]
    out["dtype"] = "f64" if wl.prec == 64 else "f32"
    out["label"] = wl.label
    out["tolerance"] = wl.tolerance
    wl.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--logn", type=int, default=20)
    ap.add_argument("--rows", type=int, default=256, help="rows of the scale grid (per GPU with --weak)")
    ap.add_argument("--partition", default="balanced", choices=["balanced", "interleaved"],
                    help="rows of a rank with more than one GPU: contiguous scales of equal estimated cost (default) or j = rank mod G")
    ap.add_argument("--weak", action="store_true", help="weak scaling: --rows rows per GPU of a rows*G-row grid")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip cpu_baseline / parity / extra")
    ap.add_argument("--no-extra", action="store_true", help="skip the config-3 block of the default run")
    ap.add_argument("--opt", action="append", default=[], help="plan option key=value (tuning)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the collectives even with one rank (smoke test of the RCCL path)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default: nccl = RCCL; gloo with --emulate)")
    ap.add_argument("--shard", default=None, metavar="R/G",
                    help="diagnostic on ONE GPU: compute only the rows rank R of G would own (j = R mod G), with the "
                         "--force-dist broadcast if given; `value` is then what G such ranks would deliver together")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="diagnostic: this many signals in flight (step i on plan / stream / W buffer i mod P); the headline is 1")
    ap.add_argument("--dummy-streams", type=int, default=0,
                    help="diagnostic: create this many idle HIP streams before the plan (the runtime maps streams to a few hardware "
                         "queues; a plan whose own streams share one loses its overlap)")
    ap.add_argument("--no-prime", dest="prime", action="store_false",
                    help="do not bring the device to its sustained clocks before the W warm-up steps (then `value` is what "
                         "`from_idle` reports otherwise)")
    ap.add_argument("--no-live-traffic", dest="live_traffic", action="store_false",
                    help="do not measure the HBM traffic with rocprofv3 PMC passes inside this run (two short child runs)")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                    help="file that receives the full result dictionary (stdout carries only the compact contract line)")
    ap.add_argument("--lib", default=None, help="tuning: another build of libcwt_hip.so (a -D variant under tools/lab/)")
    ap.add_argument("--emulate", action="store_true",
                    help="CPU rehearsal of the launch/stdout contract on the emulated kernel library (tests/emu); not a measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.shard:
        # plain `python bench.py --gpus N`: become the launcher -- one rank per GPU through torch.distributed.run on the
        # loopback address, exactly what the driver's command line does; the ranks' one JSON line passes through
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    # stdout carries exactly ONE line, the JSON result.  Libraries that write to the C stdout stream (RCCL prints
    # a version banner there on its first collective) are sent to stderr for the duration of the run.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    rt = Runtime(args)
    world, rank = rt.world, rt.rank
    dummies = [rt.torch.cuda.Stream(device=rt.dev) for _ in range(args.dummy_streams)] if not args.emulate else []
    for st in dummies:                         # (a stream gets its hardware queue at first use)
        with rt.torch.cuda.stream(st):
            rt.torch.zeros(1, device=rt.dev)
    rt.sync()
    opts = {k: int(v) for k, v in (o.split("=") for o in args.opt)}
    kind, param, prec, label = CONFIGS[args.config]
    N = 1 << args.logn
    rows_total = args.rows * world if args.weak else args.rows
    single = world == 1 and not args.no_cpu_baseline and not args.shard

    head = measure(rt, args.config, args, rows_total, opts, want_cpu=single and rank == 0)
    workload = f"N=2^{args.logn} {label} {rows_total} scales"
    if world > 1:
        workload += (f" split over {world} GPUs (contiguous cost-balanced shards)" if args.partition == "balanced"
                     else f" split over {world} GPUs (row j -> rank j mod {world})")
    out = {
        "metric": "CWT GSamples*scales/s at N=2^20, J=256", "value": head["value"], "unit": "GSamples*scales/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
        "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
        "dtype": head["dtype"], "data": "synthetic" + (" (CPU emulation rehearsal, not a measurement)" if args.emulate else ""),
        "config": {"workload": workload, "N": N, "rows_total": rows_total,
                   "rows_per_gpu": (rows_total + world - 1) // world,
                   "mother": ["morlet", "paul", "dog"][kind], "param": param,
                   "signal": "default_rng(1234).standard_normal(N)", "dt": 1.0,
                   "tolerance": head["tolerance"],
                   "parallelism": (f"scale-sharded x{world}, 1 broadcast/step, backend {rt.backend}" if world > 1
                                   else "single GPU"),
                   "plan_options": opts, **({"shard_diagnostic": args.shard} if args.shard else {}),
                   **({"signals_in_flight": args.pipeline} if args.pipeline > 1 else {})},
        "roofline": head["roofline"],
    }
    try:       # which binary was measured: the id embedded in the loaded library against the id of this tree's sources
        from pycwt_amd import _build
        tree = _build.source_id() if all(os.path.exists(d) for d in _build.DEPS) else None
        lib_id = rt.lib.build_id()
        out["build"] = {"library": os.path.relpath(rt.lib.path, ROOT), "id": lib_id, "tree_id": tree,
                        "matches_tree": (lib_id == tree) if (tree and not args.lib and not args.emulate) else None}
    except Exception as e:                                   # (never let bookkeeping take the measurement down)
        out["build"] = {"error": str(e)[:120]}
    if "cold_grid_ms" in head:
        # one step with a scale grid the plan has not seen (host classification + table upload + filter tables + step);
        # ms_per_step is the steady state with the row table cached
        out["cold_grid_ms"] = head["cold_grid_ms"]
    for k in ("parity", "cpu_baseline", "from_idle"):
        if k in head:
            out[k] = head[k]
    if "clock_priming" in head:
        out["config"]["clock_priming"] = head["clock_priming"]
    if world > 1 and not args.weak:
        # same run, weak-scaling variant (per-GPU work fixed): --rows rows per GPU of a rows*G-row grid
        weak = measure(rt, args.config, args, args.rows * world, opts, want_cpu=False)
        out["weak_scaling"] = {"value": weak["value"], "ms_per_step": weak["ms_per_step"], "rows_total": args.rows * world,
                               "rows_per_gpu": args.rows}
    for k in ("effective_warmup_steps", "icwt", "host_enqueue_ms_per_step", "api"):
        if k in head:
            out[k] = head[k]
    if single and args.config == "c2" and not args.no_extra and not opts and not args.emulate:
        out["extra"] = {}
        # the same workload at the engine's own default accuracy (round-off: every truncation below fp64 rounding)
        r = measure(rt, "c2", args, rows_total, {"tolerance": 1e-16}, want_cpu=False)
        out["extra"]["c2_roundoff"] = {"workload": workload + ", tolerance 1e-16 (the engine's default)", "value": r["value"],
                                       "unit": "GSamples*scales/s", "ms_per_step": r["ms_per_step"], "tolerance": r["tolerance"],
                                       "whole_path_frac": r["roofline"].get("whole_path_frac"),
                                       "row_split": r["roofline"].get("row_split"), "from_idle": r.get("from_idle")}
        # the same grid on COLOURED input at the automatic tolerance of the drop-in (the white-noise headline is the
        # best-case input class): AR(1) g = 0.99, target 1e-9 relative to every row's peak, every row against the oracle
        try:
            opts_red = dict(opts)
            opts_red.pop("tolerance", None)
            r = measure(rt, "c2", args, rows_total, opts_red, want_cpu=True, traffic_passes=False, signal="red", auto_target=BENCH_TOLERANCE[64])
            out["extra"]["c2_red"] = {"workload": workload + ", AR(1) g=0.99 input, cwt_plan_auto_tolerance(1e-9)", "value": r["value"],
                                      "unit": "GSamples*scales/s", "ms_per_step": r["ms_per_step"], "tolerance": r["tolerance"],
                                      "whole_path_frac": r["roofline"].get("whole_path_frac"),
                                      "row_split": r["roofline"].get("row_split"), "from_idle": r.get("from_idle"),
                                      "parity": r.get("parity")}
        except Exception as e:                 # (an extra block must never cost the contract line)
            out["extra"]["c2_red"] = {"error": repr(e)[:200]}
        out["extra"]["c1_nino3_latency"] = config1_latency()
        out["extra"]["c4_batch"] = config4_batch(rt)
        out["extra"]["c5_xwt_wct"] = config5_callers()
        import pycwt_amd
        pycwt_amd.release_scratch()                # (the shim keeps the work matrices of its last calls: tens of GB after config 5)
        for c in ("c3_paul", "c3_dog", "paul64"):
            # (paul64: not a BASELINE config -- config 3 is quoted in fp32 -- but the reference's own arithmetic for Paul,
            # mothers.py:118-122, and what pycwt_amd.cwt(..., 'paul') runs by default; parity on the rows the reference keeps)
            r = measure(rt, c, args, rows_total, {}, want_cpu=True, traffic_passes=c != "paul64")
            out["extra"][c] = {"workload": f"N=2^{args.logn} {r['label']} {rows_total} scales" + (" (BASELINE config 3)" if c != "paul64" else ""),
                               "value": r["value"], "unit": "GSamples*scales/s", "ms_per_step": r["ms_per_step"],
                               "dtype": r["dtype"], "steps": args.steps, "warmup": args.warmup, "tolerance": r["tolerance"],
                               "roofline": r["roofline"], "parity": r["parity"], "cpu_baseline": r["cpu_baseline"],
                               "cold_grid_ms": r.get("cold_grid_ms"), "from_idle": r.get("from_idle")}
    rt.close()
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)           # drain what C libraries buffered while fd 1 pointed at stderr
    except OSError:
        pass
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if rank == 0:
        # details (per-class tables, every kernel, the extra workloads in full) go to a FILE and to stderr; stdout's one and
        # LAST line is the compact contract line (< 4 KB) the driver parses
        detail = json.dumps(out)
        for path in (args.detail, os.path.join(ROOT, "gpurun_out", "bench_detail.json") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None):
            if path:
                try:
                    with open(path, "w") as f:
                        f.write(detail + "\n")
                except OSError:
                    pass
        print(detail, file=sys.stderr, flush=True)
        print(json.dumps(compact_line(out, args.detail)), flush=True)


if __name__ == "__main__":
    main()
