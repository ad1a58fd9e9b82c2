"""The two row forms of round 4 on the CPU emulation of the HIP runtime (tests/emu), against the oracle and against the
library's own older kernels:

* polynomial rows (k_poly_band / k_poly_coef / k_poly_rows): band-limited rows as K' intervals of a degree-D polynomial
  times a carrier;
* rows clipped at the Nyquist bins as overlap-save rows on the band-passed complex signal (k_aols_*).

The suite default (conftest: CWT_TOLERANCE = 1e-16) puts every truncation at round-off; the tests that pass a target check
the error against that target.
"""
import numpy as np
import pytest

from conftest import row_errors
from oracle import cwt_oracle as orc
from pycwt_amd import _hip
from test_kernels_emulated import TOL, grid


def transform(lib, N, x, kind, param, sj, prec, opts, with_signal=True):
    real, cplx = (np.float64, np.complex128) if prec == 64 else (np.float32, np.complex64)
    plan = _hip.Plan(N, prec, max_rows=len(sj), lib=lib, options=opts)
    if with_signal:
        W, _ = plan.execute_host(x, kind, param, 1.0, sj, want_xhat=False)
    else:
        es = np.dtype(real).itemsize
        xd, xh = _hip.DeviceBuffer(x.size * es, lib=lib), _hip.DeviceBuffer(2 * es * N, lib=lib)
        Wd = _hip.DeviceBuffer(2 * es * len(sj) * x.size, lib=lib)
        xd.upload(plan, x.astype(real))
        plan.forward_fft(xd.ptr, x.size, xh.ptr)
        plan.transform_rows(xh.ptr, kind, param, 1.0, sj, Wd.ptr, x.size, x.size)
        W = Wd.download(plan, (len(sj), x.size), cplx)
        for b in (xd, xh, Wd):
            b.free()
    out = (W, plan.last_split(), plan.row_classes())
    plan.close()
    return out


@pytest.mark.parametrize("kind,param,prec,logn,n0_off,rows", [
    (orc.MORLET, 6, 64, 16, 0, 96),
    (orc.MORLET, 6, 64, 16, 37, 64),          # ragged: the last workgroups of a row store a part of their span
    (orc.MORLET, 6, 32, 16, 1, 64),           # complex64: two outputs per lane, an odd number of columns
    (orc.DOG, 6, 64, 15, 0, 48),              # (the Paul rows the reference keeps are all too wide for the form: B > N / 64)
    (orc.DOG, 2, 32, 16, 100, 64),
    (orc.DOG, 3, 64, 15, 5, 40),              # odd order: imaginary mother constant
    (orc.MORLET, 6, 64, 17, 1000, 40),
])
def test_polynomial_rows(emu_library, kind, param, prec, logn, n0_off, rows):
    """Band-limited rows in polynomial form against the oracle (at round-off: the suite's accuracy target) and against the
    same rows through the transform-per-residue kernels (option poly = 0)."""
    N = 1 << logn
    n0 = N - n0_off
    x = np.random.default_rng(31).standard_normal(n0)
    m = orc.Mother(kind, param)
    sj = grid(n0, 1.0, m, rows)
    ref = orc.cwt_rows(x, 1.0, sj, m, N=N)[:, :n0]
    W, split, classes = transform(emu_library, N, x, kind, param, sj, prec, {"poly_min_logn": 14})
    assert split["poly"] >= len(sj) // 3, split
    per_row, _ = row_errors(W, ref)
    assert per_row.max() < 3 * TOL[prec], (per_row.argmax(), per_row.max(), classes[per_row.argmax()])
    W0, split0, _ = transform(emu_library, N, x, kind, param, sj, prec, {"poly": 0})
    assert split0["poly"] == 0
    mine = [i for i, c in enumerate(classes) if c.startswith("poly/")]
    assert row_errors(W[mine], W0[mine])[0].max() < 3 * TOL[prec]
    # from the spectrum alone (cwt_transform_rows) the same rows take the form
    W1, split1, classes1 = transform(emu_library, N, x, kind, param, sj, prec, {"poly_min_logn": 14}, with_signal=False)
    assert split1["poly"] >= split["poly"]
    assert row_errors(W1, ref)[0].max() < 3 * TOL[prec]


@pytest.mark.parametrize("prec,target,bar", [(64, 1e-9, 1e-9), (64, 1e-6, 1e-6), (32, 3e-5, 3e-5)])
def test_polynomial_rows_follow_the_accuracy_target(emu_library, prec, target, bar):
    """A looser target means fewer intervals or lower degrees; the error stays inside the target."""
    N = 1 << 16
    x = np.random.default_rng(4).standard_normal(N)
    m = orc.Mother(orc.MORLET, 6)
    sj = grid(N, 1.0, m, 80)[30:]
    ref = orc.cwt_rows(x, 1.0, sj, m)
    W, split, classes = transform(emu_library, N, x, orc.MORLET, 6, sj, prec, {"tolerance": target, "poly_min_logn": 14})
    Wt, _, tight = transform(emu_library, N, x, orc.MORLET, 6, sj, prec, {"poly_min_logn": 14})
    assert split["poly"] > 20
    per_row, _ = row_errors(W, ref)
    assert per_row.max() < bar, (per_row.max(), classes[per_row.argmax()])

    def cost(labels):
        return sum(int(c.split("/")[1][1:]) * (int(c.split("/")[2][1:]) + 1) for c in labels if c.startswith("poly/"))
    assert cost(classes) < cost(tight)


def test_polynomial_rows_with_a_strong_spectral_line(emu_library):
    """The truncation rule weighs a bin's Taylor remainder by the filter's value there, like the support threshold: both are
    relative to the FILTER's peak, so a spectral line far above the rest of the spectrum (here 1e4 x the noise: a dynamic
    range of 1.3e6 in |xhat|) raises the error relative to the row's own peak by that range -- 1e-17 x 1.3e6 at the
    round-off target.  The target a caller wants must be divided by the spectral dynamic range (pycwt_amd.cwt does)."""
    N = 1 << 16
    n = np.arange(N)
    x = np.random.default_rng(9).standard_normal(N) + 1e4 * np.cos(2 * np.pi * 300 * n / N)
    m = orc.Mother(orc.MORLET, 6)
    sj = grid(N, 1.0, m, 80)[30:70]
    ref = orc.cwt_rows(x, 1.0, sj, m)
    W, split, classes = transform(emu_library, N, x, orc.MORLET, 6, sj, 64, {"poly_min_logn": 14})
    assert split["poly"] > 20
    per_row, _ = row_errors(W, ref)
    assert per_row.max() < 1e-10, (per_row.max(), classes[per_row.argmax()])


@pytest.mark.parametrize("kind,param,prec,logn,n0_off,rows", [
    (orc.MORLET, 6, 64, 15, 0, 48),           # circular edges
    (orc.MORLET, 6, 64, 16, 4099, 64),        # padded signal, ragged last block
    (orc.MORLET, 5, 64, 15, 1, 40),           # lower f0: the mask reaches further into the negative frequencies
    (orc.PAUL, 4, 32, 16, 77, 96),
    (orc.PAUL, 2, 32, 15, 0, 48),             # slow t^-3 tails: long halos or no row at all
    (orc.MORLET, 6, 32, 15, 11, 48),
])
def test_rows_clipped_at_nyquist_on_the_band_passed_signal(emu_library, kind, param, prec, logn, n0_off, rows):
    """The smallest scales (filter cut at the Nyquist bins) as overlap-save rows on x_M = IFFT(xhat * mask): against the
    oracle and against the two-pass rows they replace (option aols = 0)."""
    N = 1 << logn
    n0 = N - n0_off
    x = np.random.default_rng(17).standard_normal(n0)
    m = orc.Mother(kind, param)
    sj = grid(n0, 1.0, m, rows)
    ref = orc.cwt_rows(x, 1.0, sj, m, N=N)[:, :n0]
    opts = {"ols_min_logn": 15, "poly": 0}
    W, split, classes = transform(emu_library, N, x, kind, param, sj, prec, opts)
    W0, split0, classes0 = transform(emu_library, N, x, kind, param, sj, prec, dict(opts, aols=0))
    assert split0["aols"] == 0 and split0["two_pass"] >= 3
    if (kind, param) != (orc.PAUL, 2):
        assert split["aols"] >= 3 and split["two_pass"] < split0["two_pass"], (split, split0)
    bar = 20 * TOL[prec] if prec == 64 else TOL[prec]          # fp64: the kernel's tail bound is floored at 2e-14
    per_row, _ = row_errors(W, ref)
    assert per_row.max() < bar, (per_row.argmax(), per_row.max(), classes[per_row.argmax()])
    mine = [i for i, c in enumerate(classes) if c.startswith("aols/")]
    if mine:
        assert all(classes0[i].startswith("two_pass") for i in mine)
        assert row_errors(W[mine], W0[mine])[0].max() < bar
    # the form needs the spectrum only: cwt_transform_rows takes it too
    W1, split1, _ = transform(emu_library, N, x, kind, param, sj, prec, opts, with_signal=False)
    assert split1["aols"] == split["aols"]
    assert row_errors(W1, ref)[0].max() < bar


def test_few_clipped_rows_stay_two_pass(emu_library):
    """The band-passed signal costs about one two-pass row: below aols_min_rows rows the form is not taken."""
    N = 1 << 15
    x = np.random.default_rng(3).standard_normal(N)
    m = orc.Mother(orc.MORLET, 6)
    sj = np.array([2.0, 2.5, 40.0, 300.0])
    _, split, _ = transform(emu_library, N, x, orc.MORLET, 6, sj, 64, {"ols_min_logn": 15})
    assert split["aols"] == 0 and split["two_pass"] == 2
    _, split, classes = transform(emu_library, N, x, orc.MORLET, 6, sj, 64, {"ols_min_logn": 15, "aols_min_rows": 2})
    assert split["aols"] == 2 and split["two_pass"] == 0, classes


@pytest.mark.parametrize("m_order,prec,logn,n0_off", [(2, 64, 15, 0), (2, 32, 16, 333), (1, 64, 15, 1), (3, 64, 15, 7), (6, 32, 15, 0)])
def test_dog_rows_clipped_at_nyquist(emu_library, m_order, prec, logn, n0_off):
    """DOG filters are two-sided and real: with the REAL signal at hand (cwt_transform) the positive bins are masked, the
    negative ones are their mirror image (W = 2 Re y for even orders, -2 Im y for odd ones) and the Nyquist bin, which the
    reference counts once at -pi/dt, is added by the kernel.  From the spectrum alone (cwt_transform_rows: the spectrum
    might belong to a complex signal) those rows keep the N-point transform."""
    N = 1 << logn
    n0 = N - n0_off
    x = np.random.default_rng(3).standard_normal(n0)
    m = orc.Mother(orc.DOG, m_order)
    sj = grid(n0, 1.0, m, 48)
    ref = orc.cwt_rows(x, 1.0, sj, m, N=N)[:, :n0]
    opts = {"ols_min_logn": 15, "poly": 0}
    W, split, classes = transform(emu_library, N, x, orc.DOG, m_order, sj, prec, opts)
    W0, split0, _ = transform(emu_library, N, x, orc.DOG, m_order, sj, prec, dict(opts, aols=0))
    assert split0["aols"] == 0 and split0["two_pass"] >= 3
    assert split["aols"] >= 3 and split["two_pass"] < split0["two_pass"], (split, split0)
    bar = 20 * TOL[prec] if prec == 64 else TOL[prec]
    per_row, _ = row_errors(W, ref)
    assert per_row.max() < bar, (per_row.argmax(), per_row.max(), classes[per_row.argmax()])
    assert np.abs(W.imag).max() <= (1e-12 if prec == 64 else 1e-4) * np.abs(W).max() or m_order % 2      # even orders: W is real
    W1, split1, _ = transform(emu_library, N, x, orc.DOG, m_order, sj, prec, opts, with_signal=False)
    assert split1["aols"] == 0 and split1["two_pass"] >= split0["two_pass"]      # (no overlap-save rows either without the signal)
    assert row_errors(W1, ref)[0].max() < bar


@pytest.mark.parametrize("kind,param,prec,logn,nb,rows,n0_off,opts", [
    (orc.MORLET, 6, 64, 16, 4, 48, 0, {}),
    (orc.MORLET, 6, 64, 15, 8, 40, 777, {"chunk_rows": 3}),     # padded signals; the batch goes through in chunks of 3 signals
    (orc.PAUL, 4, 32, 15, 8, 40, 1, {}),
    (orc.DOG, 2, 32, 15, 9, 32, 5, {}),                          # two-sided filter: the Nyquist bin of every signal's own spectrum
    (orc.DOG, 3, 64, 15, 8, 32, 0, {"chunk_rows": 5}),
])
def test_batch_rows_clipped_at_nyquist_on_the_band_passed_signals(emu_library, kind, param, prec, logn, nb, rows, n0_off, opts):
    """cwt_transform_batch: the rows clipped at the Nyquist bins run as overlap-save rows on every signal's band-passed
    complex signal (one mask pass and one set of block spectra per signal, one filter table per scale) instead of
    two-pass rows; every (signal, scale) pair against the oracle and against the same call without the form."""
    N = 1 << logn
    n0 = N - n0_off
    real, cplx = (np.float64, np.complex128) if prec == 64 else (np.float32, np.complex64)
    es = np.dtype(real).itemsize
    X = np.random.default_rng(9).standard_normal((nb, n0)).astype(real)
    m = orc.Mother(kind, param)
    sj = grid(N, 1.0, m, rows)
    nr = len(sj)
    out = {}
    for label, extra in (("aols", {}), ("two_pass", {"aols": 0})):
        plan = _hip.Plan(N, prec, max_rows=nb * nr, lib=emu_library, options=dict(opts, **extra))
        xd = _hip.DeviceBuffer(X.nbytes, lib=emu_library)
        xh = _hip.DeviceBuffer(nb * N * 2 * es, lib=emu_library)
        Wd = _hip.DeviceBuffer(nb * nr * n0 * 2 * es, lib=emu_library)
        xd.upload(plan, X)
        plan.transform_batch(xd.ptr, nb, n0, n0, kind, param, 1.0, sj, xh.ptr, Wd.ptr, n0, n0)
        out[label] = (Wd.download(plan, (nb, nr, n0), cplx), plan.row_classes(), plan.last_split())
        for b in (xd, xh, Wd):
            b.free()
        plan.close()
    got, labels, split = out["aols"]
    assert len(labels) == nb * nr and labels[:nr] == labels[-nr:]
    n_a = sum(l.startswith("aols") for l in labels[:nr])
    assert n_a >= 3 and split["aols"] == nb * n_a, (labels[:nr], split)
    assert out["two_pass"][2]["aols"] == 0 and out["two_pass"][2]["two_pass"] >= nb * n_a
    assert split["two_pass"] == out["two_pass"][2]["two_pass"] - nb * n_a
    for b in range(nb):
        ref = orc.cwt_rows(X[b].astype(np.float64), 1.0, sj, m, N=N)[:, :n0]
        per_row, _ = row_errors(got[b], ref)
        assert per_row.max() < TOL[prec], (b, per_row.argmax(), per_row.max(), labels[per_row.argmax()])
        per_row, _ = row_errors(got[b], out["two_pass"][0][b])
        assert per_row.max() < TOL[prec]


@pytest.mark.parametrize("prec", [64, 32])
def test_polynomial_rows_in_chunks_of_bounded_coefficient_volume(emu_library, prec):
    """The polynomial rows go through in chunks (coefficient planes of a chunk computed, then consumed): any chunk size gives
    the bits of the single-chunk run."""
    N = 1 << 16
    x = np.random.default_rng(21).standard_normal(N - 77)
    m = orc.Mother(orc.MORLET, 6)
    sj = grid(N, 1.0, m, 64)[24:]                        # (the band-limited end of the grid: parity of the form is test_polynomial_rows)
    base, split, _ = transform(emu_library, N, x, orc.MORLET, 6, sj, prec, {"poly_chunk_mb": 0})
    assert split["poly"] >= 20
    for mb in (1, 48):                                   # (1 MiB: two to three chunks at this size)
        W, s2, _ = transform(emu_library, N, x, orc.MORLET, 6, sj, prec, {"poly_chunk_mb": mb})
        assert s2 == split
        np.testing.assert_array_equal(W, base)


def test_paul_rows_continued_through_zero_frequency(emu_library, monkeypatch):
    """Round 5, form A for fp64 Paul: rows whose wavelet's 1/t^5 tail (the kink of f^m H(f) at f = 0) left them to the two-pass
    fallback run on the band-passed signal with the profile continued analytically THROUGH f = 0 and cut by a taper of the
    row's own width (AolsGeom::zc_*): two tile sizes, also the rows still alive at Nyquist where both continuations fit.
    Every row -- the reference's NaN rows included -- against the intended-value oracle at the accuracy target; the form is
    off at round-off (its price is 1.5e3 x eps of rounding noise) and in fp32."""
    monkeypatch.delenv("CWT_TOLERANCE", raising=False)
    N = 1 << 17
    n0 = N - 91
    m = orc.Mother(orc.PAUL, 4)
    s0 = 2 / m.flambda()
    sj = s0 * 2 ** (np.arange(64) * np.log2(n0 / s0) / 63)                      # (no rows dropped: the oracle below has them all)
    x = np.random.default_rng(77).standard_normal(n0)
    ref = orc.cwt_rows(x, 1.0, sj, m, N=N, intended=True)[:, :n0]
    opts = {"ols_min_logn": 15, "poly_min_logn": 15}
    W, split, classes = transform(emu_library, N, x, orc.PAUL, 4, sj, 64, dict(opts, tolerance=1e-9))
    W0, split0, classes0 = transform(emu_library, N, x, orc.PAUL, 4, sj, 64, dict(opts, tolerance=1e-9, aols_zc=0))
    assert split["aols"] >= split0["aols"] + 10 and split["two_pass"] <= split0["two_pass"] - 10, (split, split0)
    assert any(c == "aols/P8192" for c in classes) and any(c == "aols/P4096" for c in classes)
    per_row = row_errors(W, ref)[0]
    assert per_row.max() < 1e-9, (per_row.argmax(), classes[per_row.argmax()], per_row.max())
    moved = [j for j, (a, b) in enumerate(zip(classes, classes0)) if a.startswith("aols") and not b.startswith("aols")]
    assert per_row[moved].max() < 1e-10                                           # measured 8e-12: the lobe below 0 costs 1.5e3 x eps
    assert row_errors(W0, ref)[0].max() < 1e-9
    # not at round-off, not in complex64
    for prec, tol in ((64, 1e-16), (32, 3e-5)):
        _, s2, c2 = transform(emu_library, N, x, orc.PAUL, 4, sj, prec, dict(opts, tolerance=tol))
        assert not any(c == "aols/P8192" for c in c2), (prec, s2)


@pytest.mark.parametrize("kind,param", [(orc.DOG, 2), (orc.PAUL, 4), (orc.MORLET, 6)])
def test_complex64_overlap_save_rows_as_block_pairs(emu_library, kind, param):
    """complex64: a workgroup of k_ols_ct<float, 12> / k_aols_rows<float> transforms blocks 2u and 2u + 1 of its row in the two
    halves of packed registers.  Signal lengths one block apart, so that rows end on a full pair AND on the odd block out;
    every row of both forms against the oracle."""
    N = 1 << 16
    m = orc.Mother(kind, param)
    seen = set()
    for n0 in (N, N - 2900, N - 5800, N - 8700 - 1):
        x = np.random.default_rng(n0).standard_normal(n0)
        sj = grid(n0, 1.0, m, 64)
        W, split, classes = transform(emu_library, N, x, kind, param, sj, 32, {"ols_min_logn": 15, "poly": 0})
        mine = [i for i, c in enumerate(classes) if c.startswith(("ols", "aols"))]
        assert split["ols"] >= 4 and split["aols"] >= 3, split
        seen.update(classes[i].split("/")[0] for i in mine)
        ref = orc.cwt_rows(x, 1.0, sj[mine], m, N=N)[:, :n0]
        per_row, _ = row_errors(W[mine], ref)
        assert per_row.max() < TOL[32], (n0, per_row.argmax(), per_row.max(), classes[mine[per_row.argmax()]])
    assert {"ols", "aols"} <= seen


@pytest.mark.parametrize("logn,scales", [(20, [230.0, 300.0, 400.0]), (19, [150.0, 200.0, 260.0])])
def test_interval_coefficients_of_long_transforms_on_small_workgroups(emu_library, logn, scales):
    """K' = 8192 / 16384 interval coefficients as 2 / 4 decimated 4096-point transforms per job on 256-thread workgroups
    (k_poly_coef_all, option coef_small, the default) against the oracle and against the one-workgroup-per-job tiles."""
    N = 1 << logn
    n0 = N - 77
    x = np.random.default_rng(5).standard_normal(n0)
    m = orc.Mother(orc.MORLET, 6)
    sj = np.array(scales)
    ref = orc.cwt_rows(x, 1.0, sj, m, N=N)[:, :n0]
    W, split, classes = transform(emu_library, N, x, orc.MORLET, 6, sj, 64, {"coef_small": 1}, with_signal=False)
    big = [i for i, c in enumerate(classes) if c.startswith("poly/K%d/" % (N >> 6)) or c.startswith("poly/K%d/" % (N >> 7))]
    assert big and any(c.startswith("poly/K%d/" % (N >> 6)) for c in classes), classes
    assert row_errors(W, ref)[0].max() < 3 * TOL[64]
    W0, _, classes0 = transform(emu_library, N, x, orc.MORLET, 6, sj, 64, {"coef_small": 0}, with_signal=False)
    assert classes0 == classes
    assert row_errors(W[big], W0[big])[0].max() < 3 * TOL[64]


@pytest.mark.parametrize("kind,param,prec,target,bar", [
    (orc.MORLET, 6, 64, 1e-9, 2e-10),
    (orc.MORLET, 6, 64, 0.0, None),            # round-off: degrees up to 24
    (orc.PAUL, 4, 64, 1e-9, 1e-9),             # lopsided filter: carrier off the band's centre, |theta| up to ~pi at the far edge
    (orc.DOG, 2, 32, 3e-5, 1e-5),
])
def test_chebyshev_economised_polynomial_rows(emu_library, kind, param, prec, target, bar):
    """Polynomial rows with the Chebyshev-economised weights (option poly_cheb, the default: k_poly_rtab) against the oracle and
    against the Taylor weights (poly_cheb = 0) at the same accuracy target: same bar, fewer / shorter coefficient planes."""
    N = 1 << 16
    n0 = N - 321
    x = np.random.default_rng(41).standard_normal(n0)
    m = orc.Mother(kind, param)
    sj = grid(n0, 1.0, m, 96)[24:] if kind != orc.PAUL else np.geomspace(400.0, 9000.0, 40)    # (Paul: bands of <= N / 64 bins --
    ref = orc.cwt_rows(x, 1.0, sj, m, N=N, intended=kind == orc.PAUL)[:, :n0]                 # rows the reference drops: intended values)
    out = {}
    for cheb in (1, 0):
        opts = {"poly_min_logn": 14, "poly_cheb": cheb}
        if target:
            opts["tolerance"] = target
        W, split, classes = transform(emu_library, N, x, kind, param, sj, prec, opts, with_signal=False)
        idx = [i for i, c in enumerate(classes) if c.startswith("poly/")]
        planes = sum((int(c.split("/")[2][1:]) + 1) * int(c.split("/")[1][1:]) for c in classes if c.startswith("poly/"))
        out[cheb] = (W, idx, planes)
    W, idx, planes = out[1]
    assert len(idx) >= 20 and idx == out[0][1]
    assert planes < out[0][2], (planes, out[0][2])
    limit = bar if bar else 3 * TOL[prec]
    per_row, _ = row_errors(W[idx], ref[idx])
    assert per_row.max() < limit, (per_row.argmax(), per_row.max())
    per_row0, _ = row_errors(out[0][0][idx], ref[idx])
    assert per_row0.max() < limit


@pytest.mark.parametrize("prec", [64, 32])
def test_filter_rows_takes_the_polynomial_form_with_its_weight_tables(emu_library, prec):
    """cwt_filter_rows (the smoothing of xwt / wct: one Gaussian per row, its own spectrum per row) builds a row table of its own:
    its polynomial rows need the tables of the economised weights like those of a transform (a table built without them made the
    coefficient kernel read through a stale pointer: the access fault of round 6's first evidence run)."""
    N = 1 << 16
    n0 = N - 100
    rows = 12
    real, cplx = (np.float64, np.complex128) if prec == 64 else (np.float32, np.complex64)
    es = np.dtype(real).itemsize
    rng = np.random.default_rng(3)
    X = (rng.standard_normal((rows, n0)) + 1j * rng.standard_normal((rows, n0))).astype(cplx)
    s = np.geomspace(40.0, 4000.0, rows)                   # Gaussian widths in samples: exp(-0.5 (s k)^2), k = 2 pi fftfreq
    a = s * (2 * np.pi / N)
    plan = _hip.Plan(N, prec, max_rows=rows, lib=emu_library, options={"poly_min_logn": 14})
    Xd, spec, out = (_hip.DeviceBuffer(X.nbytes, lib=emu_library), _hip.DeviceBuffer(2 * es * rows * N, lib=emu_library),
                     _hip.DeviceBuffer(X.nbytes, lib=emu_library))
    Xd.upload(plan, X)
    plan.fft_rows(Xd.ptr, True, rows, n0, n0, spec.ptr)
    for rep in range(2):
        plan.filter_rows(spec.ptr, N, _hip.DOG, 0.0, a, 1.0, out.ptr, n0, n0)
        assert plan.last_split()["poly"] >= rows // 2, plan.last_split()
        Y = out.download(plan, (rows, n0), cplx)
        k = 2 * np.pi * np.fft.fftfreq(N)
        ref = np.fft.ifft(np.fft.fft(X.astype(np.complex128), N, axis=1) * np.exp(-0.5 * (s[:, None] * k[None, :]) ** 2), axis=1)[:, :n0]
        err = np.abs(Y - ref).max(axis=1) / np.abs(ref).max(axis=1)
        assert err.max() < (1e-11 if prec == 64 else 2e-5), err
    for b in (Xd, spec, out):
        b.free()
    plan.close()


def test_clipped_paul_rows_with_long_halos_on_the_second_tile_class(emu_library, monkeypatch):
    """complex128 Paul, scales of 2 ... 5 samples: the filter is cut at Nyquist AND the kink at f = 0 leaves a 1/t^5 tail with no room for the
    continuation through it -- halos of 512 ... 2048 samples.  Option aols_long (default): the second, 8192-point class of the rows on the
    band-passed signal takes them (plain window, not the continuation), where they were two-pass rows."""
    monkeypatch.delenv("CWT_TOLERANCE", raising=False)
    N = 1 << 16
    n0 = N - 33
    m = orc.Mother(orc.PAUL, 4)
    sj = np.geomspace(1.5, 12.0, 28)
    x = np.random.default_rng(78).standard_normal(n0)
    ref = orc.cwt_rows(x, 1.0, sj, m, N=N, intended=True)[:, :n0]
    opts = {"ols_min_logn": 15, "poly_min_logn": 15, "tolerance": 1e-9}
    W, split, classes = transform(emu_library, N, x, orc.PAUL, 4, sj, 64, opts)
    W0, split0, classes0 = transform(emu_library, N, x, orc.PAUL, 4, sj, 64, dict(opts, aols_long=0))
    moved = [j for j, (a, b) in enumerate(zip(classes, classes0)) if a == "aols/P8192" and b.startswith("two_pass")]
    assert len(moved) >= 4, (sorted(set(classes)), sorted(set(classes0)))
    per_row = row_errors(W, ref)[0]
    assert per_row.max() < 1e-9, (per_row.argmax(), classes[per_row.argmax()], per_row.max())
    assert row_errors(W[moved], W0[moved])[0].max() < 1e-9
