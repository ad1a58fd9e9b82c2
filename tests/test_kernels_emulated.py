"""Runs the UNMODIFIED kernel sources on the CPU emulation of the HIP runtime (tests/emu) and
compares with the oracle.  This checks index algebra, LDS staging, launch geometry and the host
orchestration without a GPU; the tests marked gpu repeat the comparison on the real device.

Tolerances: fp64 1e-12 per-row max|dW|/max|Wref| (the north_star bar is 1e-6), fp32 2e-5 (bar 1e-3).
"""
import numpy as np
import pytest

from conftest import row_errors
from oracle import cwt_oracle as orc
from pycwt_amd import _hip

TOL = {64: 1e-12, 32: 2e-5}


def grid(n0, dt, mother, rows):
    s0 = 2 * dt / mother.flambda()
    sj = s0 * 2 ** (np.arange(rows) * np.log2(n0 * dt / s0) / max(rows - 1, 1))
    return sj[~orc.dropped_rows(sj, dt, mother)]


def run_case(lib, N, n0, kind, param, rows, prec=64, opts=None, dt=1.0, seed=5):
    x = np.random.default_rng(seed).standard_normal(n0)
    m = orc.Mother(kind, param)
    sj = grid(n0, dt, m, rows)
    plan = _hip.Plan(N, prec, max_rows=len(sj), lib=lib, options=opts)
    W, xhat = plan.execute_host(x, kind, param, dt, sj)
    split = plan.last_split()
    plan.close()
    ref = orc.cwt_rows(x, dt, sj, m, N=N)[:, :n0]
    xref = np.fft.fft(x, n=N)
    assert np.abs(xhat - xref).max() / np.abs(xref).max() < TOL[prec]
    per_row, l2 = row_errors(W, ref)
    assert per_row.max() < TOL[prec], (per_row.argmax(), per_row.max(), split)
    return split


@pytest.mark.parametrize("N,n0", [(2, 2), (4, 3), (8, 8), (16, 16), (32, 30), (64, 64), (128, 100),
                                  (256, 256), (512, 504), (1024, 1000), (2048, 2048), (4096, 4000)])
def test_single_workgroup_lengths(emu_library, N, n0):
    if N == 2:
        pytest.skip("reference itself yields NaN for N = 2 (sqrt of a negative ftfreqs[1])")
    split = run_case(emu_library, N, n0, orc.MORLET, 6, 7)
    assert split["small"] > 0 and split["two_pass"] == 0


@pytest.mark.parametrize("kind,param", [(orc.MORLET, 6), (orc.MORLET, 4.5), (orc.PAUL, 4), (orc.PAUL, 2),
                                        (orc.DOG, 2), (orc.DOG, 6), (orc.DOG, 1), (orc.DOG, 3)])
def test_mothers_small(emu_library, kind, param):
    run_case(emu_library, 512, 500, kind, param, 12, dt=0.25)


@pytest.mark.parametrize("opts", [
    {"lmax": 64, "narrow": 0, "wg_points": 1024},
    {"lmax": 64, "wg_points": 1024},
    {"lmax": 64, "wg_points": 512, "chunk_rows": 1},
    {"lmax": 64, "wg_points": 256, "chunk_rows": 3},
    {"lmax": 16, "wg_points": 256, "narrow_max_k": 16},
    {"lmax": 128, "wg_points": 2048, "narrow_max_k": 256},
])
@pytest.mark.parametrize("kind,param", [(orc.MORLET, 6), (orc.PAUL, 4), (orc.DOG, 2)])
def test_two_pass_and_band_limited_paths_small_geometry(emu_library, opts, kind, param):
    """Option overrides shrink the geometry so that every multi-pass code path runs at N = 4096/256."""
    N = 256 if opts["lmax"] == 16 else 4096
    split = run_case(emu_library, N, N - 5, kind, param, 12, opts=opts)
    assert split["small"] == 0
    if not opts.get("narrow", 1):
        assert split["narrow"] == 0 and split["two_pass"] > 0


def test_default_geometry_two_pass_8192(emu_library):
    split = run_case(emu_library, 8192, 8000, orc.MORLET, 6, 16, opts={"narrow_terms": 1})
    assert split["narrow"] > 0 and split["two_pass"] > 0


@pytest.mark.parametrize("terms", [2, 3, 4])
@pytest.mark.parametrize("kind,param", [(orc.MORLET, 6), (orc.PAUL, 4), (orc.DOG, 2)])
def test_band_limited_rows_with_several_aliased_terms(emu_library, terms, kind, param):
    """Supports wider than K = 1024 bins handled in one pass (compile-time kernels, default geometry)."""
    one = run_case(emu_library, 16384, 16001, kind, param, 14, opts={"narrow_terms": 1, "narrow_big": 0})
    many = run_case(emu_library, 16384, 16001, kind, param, 14, opts={"narrow_terms": terms, "narrow_big": 0})
    assert many["narrow"] > one["narrow"]


def test_default_geometry_fp32(emu_library):
    run_case(emu_library, 16384, 16384, orc.MORLET, 6, 10, prec=32)
    run_case(emu_library, 2048, 2048, orc.DOG, 2, 10, prec=32, opts={"lmax": 64, "wg_points": 1024})
    run_case(emu_library, 1024, 1024, orc.PAUL, 4, 8, prec=32)


def test_row_subsets_and_unsorted_scales(emu_library):
    """Rows may come in any order (freqs= argument of cwt); out_row bookkeeping must hold."""
    x = np.random.default_rng(9).standard_normal(4096)
    m = orc.Mother(orc.MORLET, 6)
    sj = np.array([900.0, 2.0, 55.5, 2.0, 4000.0, 17.0])
    plan = _hip.Plan(4096, 64, max_rows=8, lib=emu_library, options={"lmax": 64, "wg_points": 1024})
    W, _ = plan.execute_host(x, orc.MORLET, 6, 1.0, sj)
    plan.close()
    per_row, _ = row_errors(W, orc.cwt_rows(x, 1.0, sj, m))
    assert per_row.max() < 1e-12


def test_icwt_reduce(emu_library):
    rng = np.random.default_rng(2)
    W = rng.standard_normal((13, 300)) + 1j * rng.standard_normal((13, 300))
    sj = 2.0 ** np.arange(13)
    plan = _hip.Plan(512, 64, max_rows=16, lib=emu_library)
    Wd = _hip.DeviceBuffer(W.nbytes, lib=emu_library)
    od = _hip.DeviceBuffer(300 * 8, lib=emu_library)
    Wd.upload(plan, W)
    plan.icwt_reduce(Wd.ptr, 300, 300, sj, 0.37, od.ptr)
    out = od.download(plan, (300,), np.float64)
    terms = 0.37 * (W.real / np.sqrt(sj)[:, None])
    # the kernel adds the rows in another order than NumPy: the bound is relative to the sum of |terms| of a column, not to the
    # (possibly cancelling) sum itself
    np.testing.assert_allclose(out, terms.sum(axis=0), rtol=0, atol=4e-16 * np.abs(terms).sum(axis=0).max())
    plan.close()


def test_profile_timings_and_errors(emu_library):
    plan = _hip.Plan(4096, 64, max_rows=4, lib=emu_library, options={"lmax": 64, "wg_points": 1024, "profile": 1})
    x = np.random.default_rng(1).standard_normal(4096)
    plan.execute_host(x, orc.MORLET, 6, 1.0, [2.0, 30.0, 900.0])
    t = plan.timings()
    assert {"fwd_pass_a", "fwd_pass_b"} <= set(t) and ("narrow" in t or "pass_a" in t)
    with pytest.raises(_hip.HipError, match="positive"):
        plan.execute_host(x, orc.MORLET, 6, 1.0, [2.0, -1.0])
    with pytest.raises(_hip.HipError, match="max_rows"):
        plan.execute_host(x, orc.MORLET, 6, 1.0, np.ones(5))
    with pytest.raises(_hip.HipError, match="Paul"):
        plan.execute_host(x, orc.PAUL, 2.5, 1.0, [2.0])
    plan.close()


@pytest.mark.parametrize("kind,param,scales", [(orc.MORLET, 6, [380.0, 76.0, 15.0, 3.0]),
                                               (orc.DOG, 2, [600.0, 150.0, 30.0, 1.0]),
                                               (orc.PAUL, 4, [200.0, 40.0, 10.0])])
def test_pass_a_classes_at_full_size(emu_library, kind, param, scales):
    """N = 2^20 (R = K = 1024): rows whose support spans <= 16 / 64 / 256 column bins use the aliased
    short column FFTs of pass_a_band_body, wider ones the full column FFT; all through one launch."""
    N = 1 << 20
    x = np.random.default_rng(5).standard_normal(N - 3)
    m = orc.Mother(kind, param)
    sj = np.array(scales)
    ref = orc.cwt_rows(x, 1.0, sj, m)[:, :x.size]
    for opts in ({"narrow_big": 0, "ols": 0, "poly": 0}, {"narrow_big": 0, "band_pass_a": 0, "ols": 0, "poly": 0}):
        plan = _hip.Plan(N, 64, max_rows=4, lib=emu_library, options=opts)
        W, _ = plan.execute_host(x, kind, param, 1.0, sj, want_xhat=False)
        assert plan.last_split()["two_pass"] == len(scales)
        plan.close()
        per_row, _ = row_errors(W, ref)
        assert per_row.max() < 1e-12, (opts, per_row)


@pytest.mark.parametrize("logn", [15, 17, 19])
def test_odd_column_lengths_default_geometry(emu_library, logn):
    """N = 2^15, 2^17, 2^19 -> column FFTs of 32, 128, 512 points (compile-time kernels for every R)."""
    split = run_case(emu_library, 1 << logn, (1 << logn) - 9, orc.MORLET, 6, 6 if logn < 19 else 4)
    assert split["two_pass"] > 0


@pytest.mark.parametrize("kind,param", [(orc.MORLET, 6), (orc.DOG, 2)])
def test_k2048_single_pass_rows(emu_library, kind, param):
    """fp64: supports of 1025..2048 bins (one term) and 4097..8192 bins (3-4 terms of 2048) run in the
    16384-point-workgroup kernel instead of the two-pass transform."""
    N = 1 << 16
    x = np.random.default_rng(3).standard_normal(N)
    m = orc.Mother(kind, param)
    # support in bins ~ c*N/s with c = 2.9 (Morlet) / 2.5 (DOG m=2): aim at 1500, 3000, 6000, 8000, 12000 bins
    c = 2.9 if kind == orc.MORLET else 2.5
    sj = c * N / np.array([1500.0, 3000.0, 6000.0, 7900.0, 12000.0])
    ref = orc.cwt_rows(x, 1.0, sj, m)
    splits = {}
    for big in (1, 0):
        plan = _hip.Plan(N, 64, max_rows=8, lib=emu_library, options={"narrow_big": big, "ols": 0})
        W, _ = plan.execute_host(x, kind, param, 1.0, sj, want_xhat=False)
        splits[big] = plan.last_split()
        plan.close()
        per_row, _ = row_errors(W, ref)
        assert per_row.max() < 1e-12, (big, per_row)
    assert splits[1]["narrow"] > splits[0]["narrow"]


def test_lab_only_options_are_refused_by_the_product_sources(emu_library):
    """The options of the measured-and-rejected variants and diagnostics of rounds 1-3 (EXPERIMENTS.md) are refused by name:
    that code left the sources in round 4."""
    plan = _hip.Plan(1 << 12, 64, max_rows=4, lib=emu_library)
    for key in ("overlap", "pass_b_prefetch", "pass_b_small", "stamps", "ols_tile", "ols_fwd_real", "sched", "narrow_wave"):
        with pytest.raises(_hip.HipError, match="EXPERIMENTS.md"):
            plan.set_option(key, 1)
    plan.close()


def test_row_table_cache_two_slots(emu_library):
    """The plan keeps the two most recent classified row tables (no rebuild / upload / host sync when calls alternate
    between two fixed argument sets, as the coherence pipeline does): hits, misses and evictions must all give the
    rows of the arguments actually passed."""
    N = 1 << 13
    x = np.random.default_rng(3).standard_normal(N)
    m = orc.Mother(orc.MORLET, 6)
    grids = {"A": np.array([2.0, 30.0, 500.0]), "B": np.array([3.0, 9.0, 81.0, 700.0]), "C": np.array([5.0, 6.0])}
    refs = {k: orc.cwt_rows(x, 1.0, v, m) for k, v in grids.items()}
    plan = _hip.Plan(N, 64, max_rows=8, lib=emu_library)
    for name in "ABABCACBBA":
        W, _ = plan.execute_host(x, orc.MORLET, 6, 1.0, grids[name], want_xhat=False)
        per_row, _ = row_errors(W, refs[name])
        assert per_row.max() < 1e-12, name
    W, _ = plan.execute_host(x, orc.DOG, 2, 1.0, grids["A"], want_xhat=False)      # same scales, other mother: a miss
    per_row, _ = row_errors(W, orc.cwt_rows(x, 1.0, grids["A"], orc.Mother(orc.DOG, 2)))
    assert per_row.max() < 1e-12
    plan.set_option("narrow", 0)                                                    # options invalidate both slots
    W, _ = plan.execute_host(x, orc.MORLET, 6, 1.0, grids["A"], want_xhat=False)
    assert plan.last_split()["narrow"] == 0
    per_row, _ = row_errors(W, refs["A"])
    assert per_row.max() < 1e-12
    plan.close()


@pytest.mark.parametrize("prec,opts,expect", [
    (64, {"narrow_terms": 8, "narrow_big": 0}, "narrow_many"),       # K = 1024, 5..8 terms in two LDS batches (fp64)
    (64, {"narrow_terms": 16, "narrow_big": 0}, "narrow_many"),      # ... up to 16 terms, four batches
    (64, {"big_terms": 8, "narrow_terms": 1}, "narrow_k2048"),       # K = 2048, up to 8 terms in two batches
    (32, {"narrow_terms": 8}, "narrow_many"),                        # fp32: 8 terms fit one batch
    (32, {"narrow_terms": 16}, "narrow_many"),                       # fp32: two batches
])
@pytest.mark.parametrize("kind,param", [(orc.MORLET, 6), (orc.DOG, 2)])
def test_many_aliased_terms_in_lds_batches(emu_library, prec, opts, expect, kind, param):
    """Supports of 5000..15000 bins in ONE pass: the aliased terms go through LDS in batches (what the exchange buffer
    holds), the Horner value of every FFT input stays in registers between batches."""
    N = 1 << 16
    x = np.random.default_rng(13).standard_normal(N)
    m = orc.Mother(kind, param)
    c = 2.9 if kind == orc.MORLET else 2.5
    sj = c * N / np.array([5000.0, 7000.0, 9000.0, 12500.0, 15500.0, 3000.0])
    plan = _hip.Plan(N, prec, max_rows=8, lib=emu_library, options=dict(opts, ols=0))
    W, _ = plan.execute_host(x, kind, param, 1.0, sj, want_xhat=False)
    split, classes = plan.last_split(), plan.row_classes()
    plan.close()
    assert split[expect] >= 1, (split, classes)
    per_row, _ = row_errors(W, orc.cwt_rows(x, 1.0, sj, m))
    assert per_row.max() < TOL[prec], (classes, per_row)


# ---- overlap-save rows (k_ols_fwd / k_ols_ct): time-compact wavelets, cwt_transform / cwt_execute_host only ----
@pytest.mark.parametrize("kind,param,prec,N,n0,rows", [
    (orc.MORLET, 6, 64, 1 << 16, 1 << 16, 64),        # circular edges (n0 = N), K = 8192 (full block) .. 256
    (orc.MORLET, 6, 64, 1 << 15, 30000, 24),          # smallest transform that takes the form, ragged last block
    (orc.MORLET, 6, 32, 1 << 16, 50001, 40),
    (orc.PAUL, 4, 32, 1 << 16, 40001, 48),            # polynomial tails: c_H = 29.5
    (orc.DOG, 2, 32, 1 << 16, 65000, 48),             # two-sided spectrum, K = 16384 (full block)
    (orc.DOG, 2, 64, 1 << 16, 50000, 40),
    (orc.DOG, 1, 64, 1 << 15, 32768, 20),             # odd order: imaginary mother constant
    (orc.MORLET, 2.0, 64, 1 << 16, (1 << 15) + 1, 40),   # low f0: the band reaches far into negative frequencies; the
                                                       # padded half of the transform is never written (blocks cover n0)
    (orc.DOG, 6, 32, 1 << 16, 65536, 40),
])
def test_overlap_save_rows(emu_library, kind, param, prec, N, n0, rows):
    """Rows whose filter is not clipped at Nyquist and whose wavelet fits a quarter tile in time are computed block
    by block from the signal itself; same values as the N-point transform of the spectrum (oracle) and as the library's
    own two-pass / band-limited kernels (option ols = 0)."""
    x = np.random.default_rng(21).standard_normal(n0)
    m = orc.Mother(kind, param)
    sj = grid(n0, 1.0, m, rows)
    ref = orc.cwt_rows(x, 1.0, sj, m, N=N)[:, :n0]
    out = {}
    for ols in (1, 0):
        plan = _hip.Plan(N, prec, max_rows=len(sj), lib=emu_library, options={"ols": ols, "ols_min_logn": 15})
        W, _ = plan.execute_host(x, kind, param, 1.0, sj, want_xhat=False)
        out[ols] = (W, plan.last_split(), plan.row_classes())
        plan.close()
    W, split, classes = out[1]
    assert split["ols"] >= 4 and out[0][1]["ols"] == 0, (split, out[0][1])
    assert split["ols"] + split["narrow"] + split["two_pass"] + split["aols"] + split["poly"] == len(sj)
    per_row, _ = row_errors(W, ref)
    assert per_row.max() < TOL[prec], (per_row.argmax(), per_row.max(), classes[per_row.argmax()])
    mine = [i for i, c in enumerate(classes) if c.startswith("ols/")]
    same, _ = row_errors(W[mine], out[0][0][mine])
    assert same.max() < TOL[prec]


def test_overlap_save_needs_the_signal_and_follows_its_options(emu_library):
    """cwt_transform_rows (spectrum only) never takes the form; cwt_transform does; ols_max_halo and ols_fwd_weight steer
    which rows and how many halo classes."""
    from pycwt_amd._hip import DeviceBuffer
    N = 1 << 16
    x = np.random.default_rng(2).standard_normal(N)
    m = orc.Mother(orc.MORLET, 6)
    sj = grid(N, 1.0, m, 48)
    ref = orc.cwt_rows(x, 1.0, sj, m)
    plan = _hip.Plan(N, 64, max_rows=len(sj), lib=emu_library, options={"ols_min_logn": 15})
    xd, xh, Wd = DeviceBuffer(x.nbytes, lib=emu_library), DeviceBuffer(16 * N, lib=emu_library), DeviceBuffer(16 * N * len(sj), lib=emu_library)
    xd.upload(plan, x)
    plan.forward_fft(xd.ptr, N, xh.ptr)
    plan.transform_rows(xh.ptr, orc.MORLET, 6, 1.0, sj, Wd.ptr, N, N)
    assert plan.last_split()["ols"] == 0
    W0 = Wd.download(plan, (len(sj), N), np.complex128)
    plan.transform(xd.ptr, N, orc.MORLET, 6, 1.0, sj, xh.ptr, Wd.ptr, N, N)
    n_all = plan.last_split()["ols"]
    assert n_all > 0
    W1 = Wd.download(plan, (len(sj), N), np.complex128)
    xhat = xh.download(plan, (N,), np.complex128)
    assert np.abs(xhat - np.fft.fft(x)).max() / np.abs(xhat).max() < 1e-13
    for W in (W0, W1):
        assert row_errors(W, ref)[0].max() < 1e-12
    plan.set_option("ols_max_halo", 256)
    plan.transform(xd.ptr, N, orc.MORLET, 6, 1.0, sj, xh.ptr, Wd.ptr, N, N)
    assert 0 < plan.last_split()["ols"] < n_all
    assert row_errors(Wd.download(plan, (len(sj), N), np.complex128), ref)[0].max() < 1e-12
    with pytest.raises(_hip.HipError):
        plan.set_option("ols_max_halo", 100)
    plan.set_option("ols_max_halo", 0)
    plan.set_option("ols_fwd_weight", 1000)            # block spectra priced at 10 rows: few, wide classes
    plan.transform(xd.ptr, N, orc.MORLET, 6, 1.0, sj, xh.ptr, Wd.ptr, N, N)
    assert plan.last_split()["ols"] == n_all
    assert row_errors(Wd.download(plan, (len(sj), N), np.complex128), ref)[0].max() < 1e-12
    for b in (xd, xh, Wd):
        b.free()
    plan.close()


@pytest.mark.parametrize("kind,param,prec,opts", [
    (orc.MORLET, 6, 64, {"ols_big": 1, "ols_big_min_halo": 256}),   # fp64: double-length blocks are opt-in
    (orc.MORLET, 6, 64, {"ols_big": 0}),
    (orc.PAUL, 4, 32, {"ols_big_min_halo": 512}),
    (orc.MORLET, 6, 64, {"ols_big": 2, "ols_big4_min_halo": 1024}),   # blocks of four tiles (one 16384-point packed block spectrum)
    (orc.MORLET, 6, 64, {"ols_small_big": 0}),             # no 8192-point blocks on pairs of half-size tiles
    (orc.MORLET, 6, 64, {"ols_small_max_halo": 0}),        # every row on the default tile
    (orc.DOG, 2, 32, {"ols_small_max_halo": 1024}),        # half-size tiles up to their limit (half the block is halo)
])
def test_overlap_save_block_and_tile_options(emu_library, kind, param, prec, opts):
    N = 1 << 17
    x = np.random.default_rng(8).standard_normal(N - 77)
    m = orc.Mother(kind, param)
    sj = grid(x.size, 1.0, m, 72)
    plan = _hip.Plan(N, prec, max_rows=len(sj), lib=emu_library, options=dict(opts, ols_min_logn=15, poly=0))
    W, _ = plan.execute_host(x, kind, param, 1.0, sj, want_xhat=False)
    split, classes = plan.last_split(), plan.row_classes()
    plan.close()
    assert split["ols"] >= 6, split
    if opts == {"ols_big": 0}:     # default tiles: short halos on half-size tiles, the rest on the default tile, in ONE transform
        assert any(c.endswith("/half") for c in classes) and any(c.startswith("ols/") and not c.endswith("/half") for c in classes)
    big = [c for c in classes if c.startswith("ols2/")]
    if opts.get("ols_big", 0) == 2:
        assert any(c.startswith("ols4/") for c in classes), sorted(set(classes))
    else:
        assert bool(big) == (opts.get("ols_big", int(prec == 32)) == 1), sorted(set(classes))
    if "ols_small_max_halo" in opts:
        assert any(c.endswith("/half") for c in classes) == (opts["ols_small_max_halo"] > 0)
    if opts == {"ols_small_big": 0}:
        # halos in (512, 1024] with a block support <= 512 bins: on the default tile here, by default on 8192-point blocks of two
        # half-size tiles each
        plan = _hip.Plan(N, prec, max_rows=len(sj), lib=emu_library, options=dict(ols_min_logn=15, poly=0))
        dflt = plan.classify(kind, param, 1.0, sj, x.size)
        plan.close()
        on_default_tile = lambda cl: sum(1 for c in cl if c.startswith("ols/") and not c.endswith("/half"))
        assert on_default_tile(classes) > on_default_tile(dflt), (sorted(set(classes)), sorted(set(dflt)))
    per_row, _ = row_errors(W, orc.cwt_rows(x, 1.0, sj, m, N=N)[:, :x.size])
    assert per_row.max() < TOL[prec], (per_row.argmax(), classes[per_row.argmax()], per_row.max())


@pytest.mark.parametrize("prec", [64, 32])
def test_launch_order_of_the_band_limited_rows_does_not_change_a_bit(emu_library, prec):
    """Option narrow_mix (default on in fp64): the band-limited rows are launched light / heavy alternating instead of sorted
    by class; every row is computed by the same code either way."""
    N = 1 << 16
    x = np.random.default_rng(23).standard_normal(N - 9)
    m = orc.Mother(orc.MORLET, 6)
    sj = grid(x.size, 1.0, m, 80)
    out = []
    for mix in (0, 1):
        plan = _hip.Plan(N, prec, max_rows=len(sj), lib=emu_library, options={"narrow_mix": mix, "ols_min_logn": 15, "poly": 0})
        W, _ = plan.execute_host(x, orc.MORLET, 6, 1.0, sj, want_xhat=False)
        assert plan.last_split()["narrow"] >= 30
        plan.close()
        out.append(W)
    assert np.array_equal(out[0], out[1])
    per_row, _ = row_errors(out[1], orc.cwt_rows(x, 1.0, sj, m, N=N)[:, :x.size])
    assert per_row.max() < TOL[prec]


@pytest.mark.parametrize("kind,param,prec,logn,nb,rows,n0_off,opts", [
    (orc.MORLET, 6, 64, 16, 4, 24, 0, None),
    (orc.DOG, 2, 32, 15, 8, 16, 0, None),
    (orc.MORLET, 6, 64, 15, 8, 40, 777, None),              # padded signals; scales that share a halo class and a halo
    (orc.DOG, 2, 64, 15, 9, 20, 5, {"ols_small_max_halo": 0}),
    (orc.MORLET, 6, 64, 16, 5, 18, 0, {"ols_big": 2, "ols_big4_min_halo": 1024}),
])
def test_batch_of_signals_takes_the_overlap_save_form(emu_library, kind, param, prec, logn, nb, rows, n0_off, opts):
    """cwt_transform_batch: forward transforms + rows of a batch from the SIGNALS.  The batch counts towards the length
    threshold of the overlap-save form, the block spectra are per signal, the filter tables per scale; every (signal,
    scale) pair against the oracle, the spectra against numpy, and against the rows computed from the spectra alone."""
    N = 1 << logn
    n0 = N - n0_off
    real, cplx = (np.float64, np.complex128) if prec == 64 else (np.float32, np.complex64)
    es = 8 if prec == 64 else 4
    X = np.random.default_rng(5).standard_normal((nb, n0)).astype(real)
    m = orc.Mother(kind, param)
    sj = grid(N, 1.0, m, rows)
    nr = len(sj)
    plan = _hip.Plan(N, prec, max_rows=nb * nr, lib=emu_library, options=opts)
    xd = _hip.DeviceBuffer(X.nbytes, lib=emu_library)
    xh = _hip.DeviceBuffer(nb * N * 2 * es, lib=emu_library)
    Wd = _hip.DeviceBuffer(nb * nr * n0 * 2 * es, lib=emu_library)
    xd.upload(plan, X)
    plan.transform_batch(xd.ptr, nb, n0, n0, kind, param, 1.0, sj, xh.ptr, Wd.ptr, n0, n0)
    got = Wd.download(plan, (nb, nr, n0), cplx)
    labels = plan.row_classes()
    assert len(labels) == nb * nr and labels[:nr] == labels[-nr:]            # every signal: the same classes
    assert sum(l.startswith("ols") for l in labels[:nr]) >= 3, labels[:nr]
    xg = xh.download(plan, (nb, N), cplx)
    xs = np.fft.fft(X.astype(np.float64), n=N, axis=1)
    assert np.abs(xg - xs).max() < TOL[prec] * np.abs(xs).max()
    for b in range(nb):
        ref = orc.cwt_rows(X[b].astype(np.float64), 1.0, sj, m, N=N)[:, :n0]
        per_row, _ = row_errors(got[b], ref)
        assert per_row.max() < TOL[prec], (b, per_row.argmax(), per_row.max(), labels[per_row.argmax()])
    # the same rows from the spectra alone (no overlap-save rows there)
    plan.transform_rows_batch(xh.ptr, nb, N, kind, param, 1.0, sj, Wd.ptr, n0, n0)
    assert not any(l.startswith("ols") for l in plan.row_classes())
    per_row, _ = row_errors(Wd.download(plan, (nb * nr, n0), cplx), got.reshape(nb * nr, n0))
    assert per_row.max() < TOL[prec]
    for b in (xd, xh, Wd):
        b.free()
    plan.close()
    # argument checks
    plan = _hip.Plan(N, prec, max_rows=nr, lib=emu_library)
    with pytest.raises(_hip.HipError):
        plan.transform_batch(1, 2, n0, n0, kind, param, 1.0, sj, 1, 1, n0, n0)            # nbatch * nrows > max_rows
    with pytest.raises(_hip.HipError):
        plan.transform_batch(1, 1, n0 - 1, n0, kind, param, 1.0, sj, 1, 1, n0, n0)        # x_ld < n0
    plan.close()
