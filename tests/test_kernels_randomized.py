"""Randomised geometry sweep of the kernel sources on the CPU emulation (hypothesis): transform length,
ragged signal length, mother, scale set (unsorted, repeated, extreme), precision and the plan's geometry
overrides are drawn at random and the result is compared with the oracle.  Complements the hand-picked
cases of test_kernels_emulated.py."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from conftest import row_errors
from oracle import cwt_oracle as orc
from pycwt_amd import _hip

MOTHERS = [(orc.MORLET, 6), (orc.MORLET, 3.5), (orc.PAUL, 4), (orc.PAUL, 1), (orc.DOG, 2), (orc.DOG, 5), (orc.DOG, 0)]


@st.composite
def cases(draw):
    logn = draw(st.integers(2, 15))
    N = 1 << logn
    n0 = draw(st.integers(N // 2 + 1, N))
    kind, param = draw(st.sampled_from(MOTHERS))
    nrows = draw(st.integers(1, 9))
    # scales from far below the resolvable range to far above the series length
    expo = draw(st.lists(st.floats(-2.0, float(logn) + 2.0), min_size=nrows, max_size=nrows))
    prec = draw(st.sampled_from([64, 64, 32]))
    opts = {}
    if logn >= 6 and draw(st.booleans()):
        lmax = 1 << draw(st.integers(max(4, (logn + 1) // 2), min(12, logn)))
        opts["lmax"] = lmax
        if lmax < N:
            opts["wg_points"] = 1 << draw(st.integers(8, 13))
            opts["narrow"] = draw(st.integers(0, 1))
            opts["chunk_rows"] = draw(st.integers(0, 3))
            if draw(st.booleans()):
                opts["narrow_max_k"] = 1 << draw(st.integers(4, 9))
    if draw(st.booleans()):
        opts["ct"] = draw(st.integers(0, 1))
        opts["narrow_terms"] = draw(st.integers(1, 4))
        opts["band_pass_a"] = draw(st.integers(0, 1))
        opts["narrow_big"] = draw(st.integers(0, 1))
        opts["pass_a_small"] = draw(st.integers(0, 1))
        opts["narrow_small"] = draw(st.integers(0, 1))
    return N, n0, kind, param, np.array([2.0 ** e for e in expo]), prec, opts, draw(st.integers(0, 2 ** 31))


@settings(max_examples=120, deadline=None, suppress_health_check=list(HealthCheck))
@given(cases())
def test_random_geometry_matches_oracle(emu_library, case):
    N, n0, kind, param, sj, prec, opts, seed = case
    m = orc.Mother(kind, param)
    with np.errstate(all="ignore"):                       # Paul rows the reference turns into NaN (and drops)
        sj = sj[~np.isnan(m.psi_ft(sj * (-np.pi / 0.7)))]
    if sj.size == 0:
        return
    x = np.random.default_rng(seed).standard_normal(n0)
    try:
        plan = _hip.Plan(N, prec, max_rows=16, lib=emu_library, options=opts)
    except _hip.HipError as e:           # option combinations the plan rejects by contract
        assert "lmax" in str(e) or "nfft" in str(e) or "wg_points" in str(e)
        return
    W, xhat = plan.execute_host(x, kind, param, 0.7, sj)
    plan.close()
    ref = orc.cwt_rows(x, 0.7, sj, m, N=N)[:, :n0]
    tol = 1e-11 if prec == 64 else 5e-5
    xref = np.fft.fft(x, n=N)
    assert np.abs(xhat - xref).max() <= tol * max(np.abs(xref).max(), 1e-300)
    scale = np.abs(ref).max(axis=1)
    err = np.abs(W - ref).max(axis=1)
    # The engine treats bins whose filter profile is below 1e-18 (fp64) / 1e-9 (fp32) of the PROFILE'S PEAK as
    # zero.  A scale whose support falls between the bins of a short series therefore comes out as exactly 0
    # where the reference returns ~1e-60: allow an absolute error of that order relative to what a resolved
    # row of this signal would carry (|W| <= sqrt(2 pi s/dt) * max|psi_ft| * ||x||_1 / ... ).
    floor = (1e-15 if prec == 64 else 1e-7) * np.sqrt(2 * np.pi * sj / 0.7) * np.abs(x).sum()
    assert (err <= tol * scale + floor).all(), (opts, err, scale)


@st.composite
def ols_cases(draw):
    """Lengths at which the overlap-save form is allowed (forced down to 2^15 for the emulator), scales drawn where
    time-compact rows live (a few samples to a quarter tile of halo) plus a few outside, unsorted and repeated."""
    logn = draw(st.integers(15, 16))
    N = 1 << logn
    n0 = draw(st.integers(N // 2 + 1, N))
    kind, param = draw(st.sampled_from([(orc.MORLET, 6), (orc.MORLET, 2.5), (orc.MORLET, 9.0), (orc.DOG, 2), (orc.DOG, 1),
                                        (orc.DOG, 7), (orc.PAUL, 4), (orc.PAUL, 2)]))
    prec = 32 if kind == orc.PAUL else draw(st.sampled_from([64, 32]))
    nrows = draw(st.integers(1, 14))
    expo = draw(st.lists(st.floats(0.5, 9.5), min_size=nrows, max_size=nrows))
    dt = draw(st.sampled_from([1.0, 0.25, 7.0]))
    opts = {"ols_min_logn": 15}
    if draw(st.booleans()):
        opts["ols_big"] = draw(st.integers(0, 1))
        opts["ols_fwd_weight"] = draw(st.sampled_from([0, 100, 1000]))
        opts["ols_max_halo"] = draw(st.sampled_from([0, 256, 1024]))
        opts["ols_side"] = draw(st.integers(0, 1))
        opts["ols_early"] = draw(st.integers(0, 1))
        opts["ols_small_max_halo"] = draw(st.sampled_from([0, 256, 512, 1024]))
    return N, n0, kind, param, dt * np.array([2.0 ** e for e in expo]), dt, prec, opts, draw(st.integers(0, 2 ** 31))


@settings(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck))
@given(ols_cases())
def test_random_overlap_save_rows_match_oracle(emu_library, case):
    N, n0, kind, param, sj, dt, prec, opts, seed = case
    m = orc.Mother(kind, param)
    with np.errstate(all="ignore"):
        sj = sj[~np.isnan(m.psi_ft(sj * (-np.pi / dt)))]
    if sj.size == 0:
        return
    x = np.random.default_rng(seed).standard_normal(n0)
    plan = _hip.Plan(N, prec, max_rows=16, lib=emu_library, options=opts)
    W, _ = plan.execute_host(x, kind, param, dt, sj, want_xhat=False)
    split, classes = plan.last_split(), plan.row_classes()
    plan.close()
    ref = orc.cwt_rows(x, dt, sj, m, N=N)[:, :n0]
    tol = 1e-11 if prec == 64 else 5e-5
    err = np.abs(W - ref).max(axis=1)
    scale = np.abs(ref).max(axis=1)
    assert (err <= tol * scale).all(), (opts, split, [(c, e / s) for c, e, s in zip(classes, err, scale)])


@st.composite
def batch_cases(draw):
    """Batches of signals through cwt_transform_batch: lengths around the threshold of the overlap-save form (which counts
    the batch), ragged signal length and leading dimension, unsorted / repeated scales, the block and tile options."""
    logn = draw(st.integers(13, 16))
    N = 1 << logn
    n0 = draw(st.integers(N // 2 + 1, N))
    x_ld = n0 + draw(st.sampled_from([0, 0, 3, 64]))
    nb = draw(st.integers(1, 9))
    kind, param = draw(st.sampled_from([(orc.MORLET, 6), (orc.MORLET, 9.0), (orc.DOG, 2), (orc.DOG, 1), (orc.PAUL, 4)]))
    prec = 32 if kind == orc.PAUL else draw(st.sampled_from([64, 32]))
    nrows = draw(st.integers(1, 10))
    expo = draw(st.lists(st.floats(0.5, 10.5), min_size=nrows, max_size=nrows))
    opts = {"ols_min_logn": draw(st.sampled_from([15, 16, 18]))}
    if draw(st.booleans()):
        opts["ols_big"] = draw(st.integers(0, 2))
        opts["ols_small_max_halo"] = draw(st.sampled_from([0, 256, 512]))
        opts["ols_fwd_weight"] = draw(st.sampled_from([0, 100, 1000]))
    return N, n0, x_ld, nb, kind, param, np.array([2.0 ** e for e in expo]), prec, opts, draw(st.integers(0, 2 ** 31))


@settings(max_examples=30, deadline=None, suppress_health_check=list(HealthCheck))
@given(batch_cases())
def test_random_batches_match_oracle(emu_library, case):
    N, n0, x_ld, nb, kind, param, sj, prec, opts, seed = case
    m = orc.Mother(kind, param)
    with np.errstate(all="ignore"):
        sj = sj[~np.isnan(m.psi_ft(sj * (-np.pi)))]
    if sj.size == 0:
        return
    real, cplx, es = (np.float64, np.complex128, 8) if prec == 64 else (np.float32, np.complex64, 4)
    X = np.zeros((nb, x_ld), dtype=real)
    X[:, :n0] = np.random.default_rng(seed).standard_normal((nb, n0))
    X[:, n0:] = 1e30                                          # the padding of the leading dimension must never be read
    plan = _hip.Plan(N, prec, max_rows=nb * sj.size, lib=emu_library, options=opts)
    xd = _hip.DeviceBuffer(X.nbytes, lib=emu_library)
    xh = _hip.DeviceBuffer(nb * N * 2 * es, lib=emu_library)
    Wd = _hip.DeviceBuffer(nb * sj.size * n0 * 2 * es, lib=emu_library)
    xd.upload(plan, X)
    plan.transform_batch(xd.ptr, nb, x_ld, n0, kind, param, 1.0, sj, xh.ptr, Wd.ptr, n0, n0)
    got = Wd.download(plan, (nb, sj.size, n0), cplx)
    classes = plan.row_classes()
    for b in (xd, xh, Wd):
        b.free()
    plan.close()
    tol = 1e-11 if prec == 64 else 5e-5
    for b in range(nb):
        ref = orc.cwt_rows(X[b, :n0].astype(np.float64), 1.0, sj, m, N=N)[:, :n0]
        err = np.abs(got[b] - ref).max(axis=1)
        scale = np.abs(ref).max(axis=1)
        assert (err <= tol * scale).all(), (opts, b, [(c, e / s) for c, e, s in zip(classes[:sj.size], err, scale)])
