"""pycwt_amd.cwt / icwt (the drop-in Python API) against the reference-generated fixtures, with the
kernels running on the CPU emulation.  Mirrors how a pycwt user calls the functions
(sample/simple_sample.py:58-60: positional arguments, mother instance or name)."""
import numpy as np
import pytest

import pycwt_amd
from conftest import load_golden, row_errors
from oracle import cwt_oracle as orc


def check_tuple(out, g, tol=1e-12):
    W, sj, freqs, coi, fft, fftfreqs = out
    assert W.shape == g["W"].shape and W.dtype == np.complex128
    per_row, l2 = row_errors(W, g["W"])
    assert per_row.max() < tol and l2 < tol
    np.testing.assert_allclose(sj, g["sj"], rtol=1e-15)
    np.testing.assert_allclose(freqs, g["freqs"], rtol=1e-15)
    np.testing.assert_allclose(coi, g["coi"], rtol=1e-15)
    np.testing.assert_allclose(fft, g["fft"], rtol=0, atol=tol * np.abs(g["fft"]).max())
    np.testing.assert_allclose(fftfreqs, g["fftfreqs"], rtol=1e-15)


def test_nino3_simple_sample_recipe(emulated):
    g = load_golden("nino3_simple")
    mother = pycwt_amd.Morlet(6)
    out = pycwt_amd.cwt(g["x"], 0.25, 1 / 12, 0.5, 84, mother)          # positional, as the sample does
    check_tuple(out, g)
    iw = pycwt_amd.icwt(out[0], out[1], 0.25, 1 / 12, mother)
    assert iw.dtype == g["icwt"].dtype == np.complex128                 # Morlet psi(0) is complex
    np.testing.assert_allclose(iw, g["icwt"], rtol=1e-11, atol=1e-12)


def test_nino3_default_scales_and_string_mother(emulated):
    g = load_golden("nino3_default")
    out = pycwt_amd.cwt(list(g["x"]), 0.25, wavelet="morlet")           # list input is accepted
    check_tuple(out, g)


def test_repeated_calls_return_private_arrays(emulated):
    """The scale grid / cone of influence of a repeated call come from a cache: what a caller does to the returned arrays
    (the reference hands out fresh ones every call) must not show up in the next call."""
    g = load_golden("nino3_default")
    first = pycwt_amd.cwt(g["x"], 0.25, wavelet="morlet")
    for a in first[1:4] + first[5:]:
        a[...] = -1.0
    check_tuple(pycwt_amd.cwt(g["x"], 0.25, wavelet="morlet"), g)
    check_tuple(pycwt_amd.cwt(g["x"], 0.25, wavelet=pycwt_amd.Morlet(6)), g)
    other = pycwt_amd.cwt(g["x"], 0.25, wavelet=pycwt_amd.Morlet(5))              # a different parameter is a different grid
    assert other[2].shape != g["freqs"].shape or not np.allclose(other[2], g["freqs"])


@pytest.mark.parametrize("name,cls", [("morlet", pycwt_amd.Morlet), ("paul", pycwt_amd.Paul),
                                      ("dog", pycwt_amd.DOG)])
def test_small_fixture_per_mother_including_nan_row_drop(emulated, name, cls):
    g = load_golden("small_" + name)
    out = pycwt_amd.cwt(g["x"], 0.5, 0.25, -1, -1, cls())
    check_tuple(out, g)                                                 # shape check covers the Paul row drop
    iw = pycwt_amd.icwt(out[0], out[1], 0.5, 0.25, name)
    assert iw.dtype == g["icwt"].dtype
    np.testing.assert_allclose(iw, g["icwt"], rtol=1e-11, atol=1e-12)


def test_freqs_argument_and_mexican_hat(emulated):
    g = load_golden("small_dog")
    f = g["freqs"][[3, 10, 20]]
    W, sj, freqs, *_ = pycwt_amd.cwt(g["x"], 0.5, freqs=f, wavelet="mexicanhat")
    per_row, _ = row_errors(W, g["W"][[3, 10, 20]])
    assert per_row.max() < 1e-12
    np.testing.assert_allclose(sj, g["sj"][[3, 10, 20]], rtol=1e-14)


def test_fp32_engine_within_1e3(emulated):
    g = load_golden("small_morlet")
    W = pycwt_amd.cwt(g["x"].astype(np.float32), 0.5, 0.25, precision=32)[0]
    assert W.dtype == np.complex128
    per_row, l2 = row_errors(W, g["W"])
    assert per_row.max() < 1e-3 and l2 < 1e-4


def test_error_conventions(emulated):
    x = np.zeros(64)
    with pytest.raises(KeyError):                       # wavelet.py:658-661
        pycwt_amd.cwt(x, 1.0, wavelet="Morlet")
    with pytest.raises(Warning, match="dimensions"):    # wavelet.py:166
        pycwt_amd.icwt(np.zeros((4, 64), complex), np.ones(5), 1.0)


def test_mother_protocol_matches_reference_constants():
    m = pycwt_amd.Morlet()
    assert (m.cdelta, m.gamma, m.deltaj0, m.dofmin, m.name) == (0.776, 2.32, 0.60, 2, "Morlet")
    assert abs(m.flambda() - 1.0330436477492537) < 1e-15
    assert pycwt_amd.Morlet(5).cdelta == -1
    p = pycwt_amd.Paul()
    assert (p.cdelta, p.gamma, p.deltaj0, p.dofmin) == (1.132, 1.17, 1.5, 2)
    d = pycwt_amd.DOG(6)
    assert (d.cdelta, d.gamma, d.deltaj0, d.dofmin) == (1.966, 1.37, 0.97, 1)
    assert pycwt_amd.MexicanHat().m == 2 and pycwt_amd.MexicanHat().name == "Mexican Hat"
    assert isinstance(pycwt_amd.Morlet().psi(0), complex) or np.iscomplexobj(pycwt_amd.Morlet().psi(0))


class _OneSidedCustomMother:
    """A mother that exists only as a Python object: complex, one-sided, not one of the built-ins."""
    name = "custom"

    def psi_ft(self, f):
        return (0.8 - 0.3j) * np.where(f > 0, f ** 1.5, 0.0) * np.exp(-0.5 * (f - 2.0) ** 2)

    def flambda(self):
        return 2.5

    def coi(self):
        return 1.1


def _numpy_cwt(x, dt, sj, mother, N):
    import scipy.fft as sfft
    w = 2 * np.pi * np.fft.fftfreq(N, dt)
    bank = (sj[:, None] * w[1] * N) ** .5 * np.conjugate(mother.psi_ft(sj[:, None] * w))
    return sfft.ifft(sfft.fft(x, n=N) * bank, axis=1)[:, :x.size]


@pytest.mark.parametrize("n0", [300, 5000, 20000])
def test_duck_typed_custom_mother_via_explicit_filter_bank(emulated, n0):
    """wavelet.py:650-663 passes any object through; the engine takes its psi_ft as an explicit table
    (single-workgroup, band-limited and two-pass paths depending on n0)."""
    x = np.random.default_rng(12).standard_normal(n0)
    m = _OneSidedCustomMother()
    W, sj, freqs, coi, fft, fftfreqs = pycwt_amd.cwt(x, 0.5, 0.5, wavelet=m)
    N = int(2 ** np.ceil(np.log2(n0)))
    ref = _numpy_cwt(x, 0.5, sj, m, N)
    per_row, l2 = row_errors(W, ref)
    assert per_row.max() < 1e-12
    np.testing.assert_allclose(coi[:3], m.flambda() * m.coi() * 0.5 * (n0 / 2 - np.abs(np.arange(3) - (n0 - 1) / 2)))


def test_reference_mother_objects_are_accepted(emulated):
    """Passing the REFERENCE's own mother instances (no device_id) must give the reference's result,
    including the Paul NaN-row drop.  Needs /root/reference (build container only)."""
    import os, sys, warnings
    if not os.path.isdir("/root/reference/pycwt"):
        pytest.skip("live reference only exists in the build container")
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import pycwt as ref
        x = np.random.default_rng(5).standard_normal(700)
        for m in (ref.Morlet(6), ref.Paul(4), ref.DOG(3), ref.MexicanHat()):
            out_ref = ref.cwt(x, 0.3, 0.25, -1, -1, m)
            out = pycwt_amd.cwt(x, 0.3, 0.25, -1, -1, m)
            assert out[0].shape == out_ref[0].shape, m.name
            per_row, _ = row_errors(out[0], out_ref[0])
            assert per_row.max() < 1e-12, m.name
            np.testing.assert_allclose(out[1], out_ref[1])


@pytest.mark.parametrize("n0,nb,name", [(500, 5, "morlet"), (9000, 3, "dog"), (9000, 2, "paul")])
def test_batch_of_signals_matches_loop_of_single_calls(emulated, n0, nb, name):
    """BASELINE config 4 shape (many signals, one scale grid) at test size: the batched launch must equal
    a Python loop of `cwt` calls (which is what a reference user would write)."""
    X = np.random.default_rng(n0).standard_normal((nb, n0))
    Wb, sj, freqs, coi, fftb, fftfreqs = pycwt_amd.cwt_batch(X, 0.5, 0.5, wavelet=name)
    assert Wb.shape == (nb, sj.size, n0) and fftb.shape[0] == nb
    for b in range(nb):
        W1, sj1, freqs1, coi1, fft1, ff1 = pycwt_amd.cwt(X[b], 0.5, 0.5, wavelet=name)
        per_row, _ = row_errors(Wb[b], W1)
        assert per_row.max() < 1e-13
        np.testing.assert_allclose(fftb[b], fft1, rtol=0, atol=1e-12 * np.abs(fft1).max())
        np.testing.assert_allclose(sj, sj1)
    # slabs smaller than the batch give the same result
    Wb2 = pycwt_amd.cwt_batch(X, 0.5, 0.5, wavelet=name, max_batch_bytes=1)[0]
    assert np.abs(Wb2 - Wb).max() == 0


def test_device_resident_transform_and_reductions(emulated):
    """The canonical workflow of sample/simple_sample.py:58-91 with W kept on the device."""
    g = load_golden("nino3_simple")
    m = pycwt_amd.Morlet(6)
    dt, dj = 0.25, 1 / 12
    T = pycwt_amd.cwt_device(g["x"], dt, dj, 0.5, 84, m)
    W = g["W"]
    power = np.abs(W) ** 2
    assert T.shape == W.shape
    per_row, _ = row_errors(T.W(), W)
    assert per_row.max() < 1e-12
    np.testing.assert_allclose(T.global_power(), power.mean(axis=1), rtol=1e-11)
    period = 1 / g["freqs"]
    sel = (period >= 2) & (period < 8)                  # simple_sample.py:85-91
    s1, s2 = g["sj"][sel].min(), np.nextafter(g["sj"][sel].max(), np.inf)
    expect = dj * dt / m.cdelta * (power / g["sj"][:, None])[sel].sum(axis=0)
    np.testing.assert_allclose(T.scale_average(s1, s2, dj), expect, rtol=1e-11)
    np.testing.assert_allclose(T.icwt(dj), g["icwt"], rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(T.sj, g["sj"])
    np.testing.assert_allclose(T.coi, g["coi"])
    T.close()


def test_plan_cache_growth_keeps_live_transforms_valid(emulated):
    """A later call that needs more rows replaces the cached plan; a DeviceTransform made before must keep working
    (its plan is dropped from the cache, not destroyed)."""
    rng = np.random.default_rng(4)
    x = rng.standard_normal(300)
    T = pycwt_amd.cwt_device(x, 0.5, 0.5, -1, -1, "morlet")
    before = T.global_power()
    pycwt_amd.cwt_batch(rng.standard_normal((90, 300)), 0.5, 0.25, -1, -1, "morlet")   # 90 x 29 rows > 1024
    np.testing.assert_array_equal(T.global_power(), before)
    W = T.W()
    ref = orc.cwt(x, 0.5, 0.5, -1, -1, "morlet")[0]
    assert np.abs(W - ref).max() < 1e-12 * np.abs(ref).max()
    T.close()


def test_threads_sharing_a_cached_plan_get_the_serial_results(emulated):
    """pycwt.cwt is re-entrant; here every (length, precision, device) has ONE cached C plan, so concurrent callers
    are serialised on its lock.  Eight threads, two signal lengths, different scale grids: every result must equal
    the one computed alone."""
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(8)
    jobs = [(rng.standard_normal(n), dj, w) for n in (500, 4000) for dj in (0.5, 0.25)
            for w in ("morlet", "dog")]
    serial = [pycwt_amd.cwt(x, 1.0, dj, -1, -1, w)[0] for x, dj, w in jobs]
    with ThreadPoolExecutor(max_workers=8) as ex:
        par = list(ex.map(lambda j: pycwt_amd.cwt(j[0], 1.0, j[1], -1, -1, j[2])[0], jobs * 3))
    for i, W in enumerate(par):
        np.testing.assert_array_equal(W, serial[i % len(jobs)])


def test_rejected_option_leaves_the_plan_usable(emulated):
    from pycwt_amd import _hip
    N = 1 << 13
    x = np.random.default_rng(2).standard_normal(N)
    sj = np.array([2.0, 20.0, 700.0])
    plan = _hip.Plan(N, 64, max_rows=4, lib=emulated)
    ref = orc.cwt_rows(x, 1.0, sj, orc.Mother(orc.MORLET, 6))
    W0, _ = plan.execute_host(x, orc.MORLET, 6, 1.0, sj)
    for key, val in (("two_pass_logk", 1), ("two_pass_logk", 99), ("wg_points", 7), ("lmax", 8)):
        with pytest.raises(_hip.HipError):
            plan.set_option(key, val)
        W, _ = plan.execute_host(x, orc.MORLET, 6, 1.0, sj)
        np.testing.assert_array_equal(W, W0)
    assert np.abs(W0 - ref).max() < 1e-12 * np.abs(ref).max()
    plan.close()


@pytest.mark.parametrize("precision,tol", [(64, 1e-11), (32, 2e-4)])
def test_unpadded_transform_lengths_bluestein(emulated, precision, tol):
    """pad=False: the reference with pyfftw transforms at len(signal) (helpers.py:15-19).  Lengths that are not powers
    of two run through Bluestein's identity on the power-of-two engine; fixture from the reference's own branch."""
    g = load_golden("unpadded")
    for tag in "abc":
        x, name = g[f"{tag}_x"], str(g[f"{tag}_name"])
        out = pycwt_amd.cwt(x, 0.5, 1 / 4, -1, -1, name, pad=False, precision=precision)
        assert out[0].shape == g[f"{tag}_W"].shape and out[0].dtype == np.complex128
        per_row, l2 = row_errors(out[0], g[f"{tag}_W"])
        assert per_row.max() < tol and l2 < tol, (tag, per_row.max())
        np.testing.assert_allclose(out[1], g[f"{tag}_sj"], rtol=1e-15)
        np.testing.assert_allclose(out[3], g[f"{tag}_coi"], rtol=1e-15)
        np.testing.assert_allclose(out[4], g[f"{tag}_fft"], rtol=0, atol=tol * np.abs(g[f"{tag}_fft"]).max())
        np.testing.assert_allclose(out[5], g[f"{tag}_fftfreqs"], rtol=1e-15)


@pytest.mark.parametrize("n0", [3, 17, 1000, 4097, 65521])
def test_unpadded_lengths_against_the_oracle(emulated, n0):
    x = np.random.default_rng(n0).standard_normal(n0)
    for name in ("morlet", "dog"):
        out = pycwt_amd.cwt(x, 1.0, 1.0, -1, -1, name, pad=False)
        ref = orc.cwt(x, 1.0, 1.0, -1, -1, name, pad=False)
        assert out[0].shape == ref[0].shape
        per_row, _ = row_errors(out[0], ref[0])
        assert per_row.max() < 1e-11, (n0, name, per_row.max())
        np.testing.assert_allclose(out[4], ref[4], rtol=0, atol=1e-11 * max(np.abs(ref[4]).max(), 1e-300) if ref[4].size else 0)


@pytest.mark.parametrize("name,dt,n0", [("morlet", 0.25, 40000), ("dog", 3.0, 32768), ("mexicanhat", 0.5, 50001)])
def test_drop_in_call_with_overlap_save_rows(emulated, monkeypatch, name, dt, n0):
    """The drop-in call with time-compact rows on the overlap-save kernels (forced on at these short lengths through
    wavelet.PLAN_OPTIONS; by default the form starts at N = 2^18): cwt, icwt and the device-resident variant against
    the oracle at dt != 1, with a ragged length, default and explicit scale grids."""
    from pycwt_amd import wavelet
    monkeypatch.setattr(wavelet, "PLAN_OPTIONS", {"ols_min_logn": 15})
    x = np.random.default_rng(n0).standard_normal(n0)
    mother = wavelet._check_parameter_wavelet(name)
    m = orc.Mother(*wavelet._device_id(mother))
    W, sj, freqs, coi, fft, fftfreqs = pycwt_amd.cwt(x, dt, 1 / 4, -1, -1, name)
    plan = next(iter(wavelet._plans.values()))
    assert plan.last_split()["ols"] >= 4, plan.last_split()
    N = 1 << int(np.ceil(np.log2(n0)))
    ref = orc.cwt_rows(x, dt, sj, m, N=N)[:, :n0]
    per_row, _ = row_errors(W, ref)
    assert per_row.max() < 1e-12, (per_row.argmax(), per_row.max())
    np.testing.assert_allclose(fft, (np.fft.fft(x, n=N)[1:N // 2]) / np.sqrt(N), rtol=0, atol=1e-12 * np.abs(fft).max())
    T = pycwt_amd.cwt_device(x, dt, 1 / 4, -1, -1, name)
    np.testing.assert_allclose(T.global_power(), (np.abs(ref) ** 2).mean(axis=1), rtol=1e-11)
    iw = T.icwt(1 / 4)
    np.testing.assert_allclose(iw, pycwt_amd.icwt(W, sj, dt, 1 / 4, mother), rtol=1e-11, atol=1e-12)
    sel = sj[3:40:4]
    W2 = pycwt_amd.cwt(x, dt, wavelet=name, freqs=1 / (mother.flambda() * sel))[0]
    per_row, _ = row_errors(W2, ref[3:40:4])
    assert per_row.max() < 1e-12


@pytest.mark.parametrize("bad", [np.nan, np.inf, -np.inf])
@pytest.mark.parametrize("name", ["morlet", "paul"])
def test_non_finite_sample_poisons_every_row_like_the_reference(emulated, monkeypatch, bad, name):
    """wavelet.py:91 transforms the whole padded signal, so ONE NaN / inf sample makes every bin of the spectrum and
    every element of W NaN, and :111-115 then keeps all rows (also the ones Paul would lose otherwise).  The
    overlap-save rows would confine the damage to the blocks that contain the sample; the shim routes such signals
    through the spectrum-only entry points instead."""
    from pycwt_amd import wavelet
    monkeypatch.setattr(wavelet, "PLAN_OPTIONS", {"ols_min_logn": 15})
    n0 = 40000
    x = np.random.default_rng(3).standard_normal(n0)
    good = pycwt_amd.cwt(x, 1.0, 1 / 4, -1, -1, name)
    plan = next(iter(wavelet._plans.values()))
    if name == "morlet":
        assert plan.last_split()["ols"] >= 4                # the clean signal does use the overlap-save rows
    x[12345] = bad
    W, sj, freqs, coi, fft, fftfreqs = pycwt_amd.cwt(x, 1.0, 1 / 4, -1, -1, name)
    ref = orc.cwt(x, 1.0, 1 / 4, -1, -1, name)
    assert W.shape == ref[0].shape and W.shape[0] >= good[0].shape[0]
    assert np.isnan(W).all() and np.isnan(ref[0]).all()
    np.testing.assert_allclose(sj, ref[1], rtol=1e-15)
    assert not np.isfinite(fft).any()
    T = pycwt_amd.cwt_device(x, 1.0, 1 / 4, -1, -1, name)
    assert np.isnan(T.W()).all() and T.shape == W.shape


@pytest.mark.parametrize("name", ["mauna", "monsoon", "sunspot", "soi"])
def test_reference_sample_datasets_through_the_shim(emulated, name):
    """The reference's other sample datasets (sample/dataset.py:68-135) with sample/sample.py's recipe, fixtures from
    the unmodified reference (every third row of W kept)."""
    g = load_golden("sample_" + name)
    W, sj, freqs, coi, fft, fftfreqs = pycwt_amd.cwt(g["x"], float(g["dt"]), 1 / 12, -1, -1, pycwt_amd.Morlet(6))
    assert W.shape == (int(g["nrows"]), g["x"].size) and W.dtype == np.complex128
    per_row, l2 = row_errors(W[g["rows"]], g["W"])
    assert per_row.max() < 1e-12 and l2 < 1e-12
    for a, b in ((sj, g["sj"]), (freqs, g["freqs"]), (coi, g["coi"]), (fftfreqs, g["fftfreqs"])):
        np.testing.assert_allclose(a, b, rtol=1e-14)
    np.testing.assert_allclose(fft, g["fft"], rtol=0, atol=1e-12 * np.abs(g["fft"]).max())
    np.testing.assert_allclose(pycwt_amd.icwt(W, sj, float(g["dt"]), 1 / 12, pycwt_amd.Morlet(6)), g["icwt"], rtol=1e-10, atol=1e-11)
