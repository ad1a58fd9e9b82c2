"""Multi-process path of pycwt_amd.parallel under the gloo backend (world_size 2, CPU).

The collective logic (one broadcast of the signal, interleaved row shards, one reduce for icwt) is
exercised for real.  The per-rank compute engine is the product's HipEngine -> C ABI with the library swapped for the
CPU emulation of the real kernels (tests/emu), and, in one test, a stand-in built on the oracle injected through the
`engine=` hook to check the call pattern."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class OracleEngine:
    """CPU engine with the interface of parallel.HipEngine (test infrastructure)."""

    def __init__(self):
        from oracle import cwt_oracle as orc
        self.orc = orc
        self.calls = []

    def forward(self, x, n0, xhat):
        self.calls.append("forward")
        xhat.copy_(torch.from_numpy(np.fft.fft(x.numpy(), n=xhat.numel())))
        self.x = x.numpy().copy()

    def rows(self, xhat, kind, param, dt, sj, W, ncols):
        self.calls.append(("rows", len(sj)))
        m = self.orc.Mother(kind, int(param) if kind else param)
        W.copy_(torch.from_numpy(self.orc.cwt_rows(self.x, dt, sj, m)[:, :ncols]))

    def icwt_partial(self, W, sj, out):
        out.copy_(torch.from_numpy((W.numpy().real / np.sqrt(sj)[:, None]).sum(axis=0)))


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pycwt_amd import parallel
    x = np.random.default_rng(4).standard_normal(777) if rank == 0 else None    # only rank 0 has the signal
    eng = OracleEngine()
    W, mine, sj, freqs, coi = parallel.cwt_sharded(x, 0.3, 1 / 6, -1, -1, "paul", engine=eng,
                                                   device=torch.device("cpu"))
    assert eng.calls[0] == "forward" and eng.calls[1] == ("rows", len(mine))
    iw = parallel.icwt_sharded(W, sj[mine], 0.3, 1 / 6, "paul", engine=eng)
    np.savez(os.path.join(tmpdir, f"rank{rank}.npz"), W=W.numpy(), mine=mine, sj=sj, freqs=freqs, coi=coi,
             iw=np.zeros(0) if iw is None else iw)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_broadcast_and_reduce(tmp_path):
    from oracle import cwt_oracle as orc
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    x = np.random.default_rng(4).standard_normal(777)
    W, sj, freqs, coi, _, _ = orc.cwt(x, 0.3, 1 / 6, -1, -1, "paul")
    assert list(r0["mine"]) == list(range(0, len(sj), 2)) and list(r1["mine"]) == list(range(1, len(sj), 2))
    full = np.empty_like(W)
    full[r0["mine"]] = r0["W"]
    full[r1["mine"]] = r1["W"]
    np.testing.assert_allclose(full, W, rtol=1e-12, atol=1e-13)        # rank 1 got the signal by broadcast
    for r in (r0, r1):
        np.testing.assert_allclose(r["sj"], sj)
        np.testing.assert_allclose(r["coi"], coi)
    np.testing.assert_allclose(r0["iw"], orc.icwt(W, sj, 0.3, 1 / 6, "paul"), rtol=1e-11, atol=1e-13)
    assert r1["iw"].size == 0


def _kernel_worker(rank, world, port, tmpdir):
    """cwt_sharded / icwt_sharded with the DEFAULT engine (HipEngine -> C ABI), the library being the CPU emulation of
    the real kernels: C-ABI plumbing, broadcast and sharding run together on 2 ranks, for one signal and a batch."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import build_emu
    from pycwt_amd import _hip, parallel
    _hip._default = _hip.Library(build_emu.build())      # test infrastructure: kernels on the CPU emulation
    cpu = torch.device("cpu")
    x = np.random.default_rng(6).standard_normal(5000) if rank == 0 else None
    W, mine, sj, freqs, coi = parallel.cwt_sharded(x, 0.5, 1 / 4, -1, -1, "morlet", device=cpu)
    iw = parallel.icwt_sharded(W, sj[mine], 0.5, 1 / 4, "morlet")
    # a caller-supplied engine sized for ITS OWN SHARE cannot classify the whole grid: interleaved rows, no error
    small = parallel.HipEngine(8192, 64, -(-len(sj) // world), 0, on_torch_stream=False)
    W2, mine2, *_ = parallel.cwt_sharded(x, 0.5, 1 / 4, -1, -1, "morlet", device=cpu, engine=small)
    assert list(mine2) == list(range(rank, len(sj), world))
    W3, mine3, *_ = parallel.cwt_sharded(x, 0.5, 1 / 4, -1, -1, "morlet", device=cpu)     # second call: cached engine
    assert list(mine3) == list(mine) and len(parallel._engines) == 1
    # the call is as cheap as its kernels: with the shape known to every rank EXACTLY ONE collective (the broadcast of the
    # signal) and no other traffic; without it one more broadcast (three integers), never an object broadcast
    import collections
    counts = collections.Counter()
    names = ("broadcast", "broadcast_object_list", "all_reduce", "reduce", "all_gather", "all_gather_object", "gather", "scatter",
             "barrier", "send", "recv", "isend", "irecv", "all_to_all", "reduce_scatter")
    keep = {n: getattr(dist, n) for n in names}

    def counted(n):
        def call(*a, **k):
            counts[n] += 1
            return keep[n](*a, **k)
        return call
    for n in names:
        setattr(dist, n, counted(n))
    try:
        W4, mine4, *_ = parallel.cwt_sharded(x, 0.5, 1 / 4, -1, -1, "morlet", device=cpu, shape=(5000,), assume_finite=True)
        assert dict(counts) == {"broadcast": 1}, dict(counts)
        counts.clear()
        xt = torch.from_numpy(x) if rank == 0 else None                  # a tensor that is already on the device: used as it is
        W5, mine5, *_ = parallel.cwt_sharded(xt, 0.5, 1 / 4, -1, -1, "morlet", device=cpu, assume_finite=True)
        assert dict(counts) == {"broadcast": 2}, dict(counts)
    finally:
        for n in names:
            setattr(dist, n, keep[n])
    assert list(mine4) == list(mine) and list(mine5) == list(mine)
    assert np.array_equal(W4.numpy(), W.numpy()) and np.array_equal(W5.numpy(), W.numpy())
    X = np.random.default_rng(7).standard_normal((3, 700)) if rank == 0 else None
    Wb, mineb, sjb, _, _ = parallel.cwt_sharded(X, 1.0, 1 / 2, -1, -1, "dog", device=cpu, precision=32)
    np.savez(os.path.join(tmpdir, f"k{rank}.npz"), W=W.numpy(), mine=mine, sj=sj, iw=np.zeros(0) if iw is None else iw,
             Wb=Wb.numpy(), mineb=mineb, sjb=sjb, W2=W2.numpy(), mine2=mine2, W3=W3.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_run_the_real_kernels_through_the_c_abi(tmp_path):
    from oracle import cwt_oracle as orc
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    build_emu.build()                                     # build once, before the workers race for it
    port = _free_port()
    mp.spawn(_kernel_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [np.load(tmp_path / f"k{i}.npz") for i in range(2)]
    x = np.random.default_rng(6).standard_normal(5000)
    W, sj, *_ = orc.cwt(x, 0.5, 1 / 4, -1, -1, "morlet")
    full = np.empty_like(W)
    for q in r:
        full[q["mine"]] = q["W"]
    assert np.abs(full - W).max() < 1e-12 * np.abs(W).max()
    for q in r:
        assert np.abs(q["W2"] - W[q["mine2"]]).max() < 1e-12 * np.abs(W).max()
        assert np.array_equal(q["W3"], q["W"])
    np.testing.assert_allclose(r[0]["iw"], orc.icwt(W, sj, 0.5, 1 / 4, "morlet"), rtol=1e-10, atol=1e-12)
    X = np.random.default_rng(7).standard_normal((3, 700))
    for b in range(3):
        Wr, sjr, *_ = orc.cwt(X[b].astype(np.float32), 1.0, 1 / 2, -1, -1, "dog")
        got = np.empty_like(Wr)
        for q in r:
            got[q["mineb"]] = q["Wb"][b]
        assert np.abs(got - Wr).max() < 3e-5 * np.abs(Wr).max()


def _mc_worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import build_emu
    from pycwt_amd import _hip, parallel
    _hip._default = _hip.Library(build_emu.build())      # test infrastructure: kernels on the CPU emulation
    sig = parallel.wct_significance_sharded(0.3, 0.5, dt=1.0, dj=0.5, s0=2.0, J=6, mc_count=10, seed=100)
    np.save(os.path.join(tmpdir, f"mc{rank}.npy"), sig)
    # surrogates made on the device: the ranks take consecutive blocks of ONE Philox sequence
    sig = parallel.wct_significance_sharded(0.3, 0.5, dt=1.0, dj=0.5, s0=2.0, J=6, mc_count=9, seed=77, rng="device")
    np.save(os.path.join(tmpdir, f"mcdev{rank}.npy"), sig)
    dist.barrier()
    dist.destroy_process_group()


def test_monte_carlo_draws_sharded_with_one_allreduce(tmp_path):
    """Config 5's Monte-Carlo part: draws split over 2 ranks, histograms all-reduced; both ranks must end
    with the same percentiles, and they must equal a single-rank run over the same 10 surrogate pairs."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    build_emu.build()                                     # build once, before the workers race for it
    port = _free_port()
    mp.spawn(_mc_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "mc0.npy"), np.load(tmp_path / "mc1.npy")
    np.testing.assert_allclose(a, b, equal_nan=True)
    ok = np.isfinite(a)
    assert ok.any() and (a[ok] > 0.2).all() and (a[ok] <= 1).all()
    # same surrogates in one process: rank r used seed 100 + r for its 5 draws
    from pycwt_amd import _hip, wavelet
    _hip._default = _hip.Library(build_emu.build())
    try:
        m = wavelet._check_parameter_wavelet("morlet")
        N, sj, outside, rows, maxscale = wavelet._mc_setup(m, 1.0, 0.5, 2.0, 6)
        hist = 0
        for r in range(2):
            np.random.seed(100 + r)
            hist = hist + wavelet._mc_histogram(5, 0.3, 0.5, 1.0, 0.5, sj, N, outside, maxscale, m, 64, 0)
        np.testing.assert_allclose(wavelet._mc_percentiles(hist, rows, maxscale, 0.95), a, equal_nan=True)
        # device surrogates: 5 + 4 draws on two ranks == the 9 draws of the same sequence on one
        da, db = np.load(tmp_path / "mcdev0.npy"), np.load(tmp_path / "mcdev1.npy")
        np.testing.assert_array_equal(da, db)
        one = wavelet._mc_histogram(9, 0.3, 0.5, 1.0, 0.5, sj, N, outside, maxscale, m, 64, 0, rng="device", seed=77)
        np.testing.assert_array_equal(wavelet._mc_percentiles(one, rows, maxscale, 0.95), da)
    finally:
        _hip._default = None
        for p in wavelet._plans.values():
            p.close()
        wavelet._plans.clear()


def test_shard_rows_partition():
    from pycwt_amd.parallel import shard_rows
    for n in (1, 7, 256):
        for world in (1, 2, 4, 8):
            parts = [shard_rows(n, world, r) for r in range(world)]
            assert sorted(np.concatenate(parts)) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_balanced_shards_cover_the_grid_and_even_out_the_cost(emu_library):
    """Contiguous cost-balanced shards (parallel.balanced_shards on Plan.classify labels): every row exactly once, in
    scale order, estimated per-rank cost within a few rows of each other, and never worse than interleaving by the
    same cost model."""
    from oracle import cwt_oracle as orc
    from pycwt_amd import _hip, parallel
    N = 1 << 16
    m = orc.Mother(orc.MORLET, 6)
    s0 = 2 / m.flambda()
    sj = s0 * 2 ** (np.arange(96) * np.log2(N / s0) / 95)
    plan = _hip.Plan(N, 64, max_rows=len(sj), lib=emu_library, options={"ols_min_logn": 15})
    labels = plan.classify(orc.MORLET, 6.0, 1.0, sj, N, True)
    assert len(labels) == len(sj) and any(l.startswith("ols") for l in labels)
    assert not any(l.startswith("ols") for l in plan.classify(orc.MORLET, 6.0, 1.0, sj, N, False))
    assert _hip.Plan._labels(_hip.Plan.codes_of(labels)) == labels           # label <-> code round trip
    by_plan = {w: plan.balanced_shards(orc.MORLET, 6.0, 1.0, sj, N, w) for w in (2, 8)}   # the C entry point a non-Python host uses
    plan.close()
    for w, shards in by_plan.items():
        assert all(np.array_equal(a, b) for a, b in zip(shards, parallel.balanced_shards(labels, w, 64, N, lib=emu_library)))
    for world in (1, 2, 3, 4, 8):
        shards = parallel.balanced_shards(labels, world, 64, N, lib=emu_library)
        assert len(shards) == world
        assert np.array_equal(np.concatenate(shards), np.arange(len(sj)))
        cost = [parallel.shard_cost([labels[i] for i in s], 64, N, lib=emu_library) for s in shards]
        inter = [parallel.shard_cost([labels[i] for i in range(r, len(sj), world)], 64, N, lib=emu_library)
                 for r in range(world)]
        assert max(cost) <= max(inter) + 1e-9, (world, cost, inter)
        if world > 1:
            assert max(cost) - min(c for c in cost if c > 0) < 40.0, (world, cost)
