"""Callers of the hot path: significance, xwt, Morlet.smooth, wct and the host
helpers, against fixtures produced by the unmodified reference (oracle/gen_golden.py: callers()).
Kernels run on the CPU emulation here; tests/test_gpu_parity.py repeats the device part on the GPU."""
import numpy as np
import pytest

import pycwt_amd
from conftest import load_golden
from oracle import cwt_oracle as orc


@pytest.fixture(scope="module")
def g():
    return load_golden("callers")


def test_host_helpers(g):
    np.testing.assert_allclose(pycwt_amd.ar1(g["y1"]), g["ar1_y1"], rtol=1e-12)
    np.testing.assert_allclose(pycwt_amd.ar1(g["y2"]), g["ar1_y2"], rtol=1e-12)
    m = pycwt_amd.Morlet(6)
    freqs = 1 / (m.flambda() * g["sj"])
    np.testing.assert_allclose(pycwt_amd.ar1_spectrum(freqs * float(g["dt"]), 0.55), g["ar1_spec"], rtol=1e-13)
    np.testing.assert_allclose(pycwt_amd.helpers.rect(7, normalize=True), g["rect7"])
    np.testing.assert_allclose(pycwt_amd.helpers.rect(1, True), g["rect1"])
    assert list(pycwt_amd.find(np.array([[0, 1], [1, 0]]))) == [1, 2]
    with pytest.raises(Warning):
        pycwt_amd.ar1(np.arange(5.0))              # strong trend: no real root, raised like the reference
    r = pycwt_amd.rednoise(4000, 0.7, ar1=True)        # the AR(1) process the name promises
    assert r.shape == (4000,) and abs(pycwt_amd.ar1(r)[0] - 0.7) < 0.08
    r = pycwt_amd.rednoise(4000, 0.7)                  # what the reference really draws: white (helpers.py:170)
    assert r.shape == (4000,) and abs(pycwt_amd.ar1(r)[0]) < 0.08
    mc = load_golden("mc_significance")                # ... seed for seed
    np.random.seed(int(mc["rednoise_seed"]))
    np.testing.assert_array_equal(pycwt_amd.rednoise(400, 0.72, 1), mc["rednoise_a"])
    np.testing.assert_array_equal(pycwt_amd.rednoise(100, 0.3, 2.0), mc["rednoise_b"])
    assert pycwt_amd.rednoise(10, 0).shape == (10,)  # the reference crashes here (np.randn)
    assert pycwt_amd.get_cache_dir().endswith("/.cache/pycwt/")


def test_significance_three_tests(g):
    y1, dt, sj = g["y1"], float(g["dt"]), g["sj"]
    m = pycwt_amd.Morlet(6)
    s0, f0 = pycwt_amd.significance(y1, dt, sj, 0, None, 0.95, -1, m)
    np.testing.assert_allclose(s0, g["sig0"], rtol=1e-12)
    np.testing.assert_allclose(f0, g["fft0"], rtol=1e-12)
    s1, f1 = pycwt_amd.significance(y1.std() ** 2, dt, sj, 1, 0.6, 0.95, y1.size - sj, m)
    np.testing.assert_allclose(s1, g["sig1"], rtol=1e-12)
    s2, f2 = pycwt_amd.significance(y1.std() ** 2, dt, sj, 2, 0.6, 0.95, [2, 8], "morlet")
    np.testing.assert_allclose(np.atleast_1d(s2), g["sig2"], rtol=1e-12)
    np.testing.assert_allclose(np.atleast_1d(f2), g["fft2"], rtol=1e-12)
    with pytest.raises(ValueError):
        pycwt_amd.significance(1.0, dt, sj, 3, 0.5)
    with pytest.raises(ValueError):
        pycwt_amd.significance(1.0, dt, sj, 2, 0.5, dof=[2, 8], wavelet=pycwt_amd.Morlet(5))


def test_xwt(emulated, g):
    W12, coi, freq, signif = pycwt_amd.xwt(g["y1"], g["y2"], float(g["dt"]), float(g["dj"]), -1, -1, 0.95,
                                           pycwt_amd.Morlet(6), True)
    assert W12.shape == g["xwt_W12"].shape
    assert np.abs(W12 - g["xwt_W12"]).max() < 1e-11 * np.abs(g["xwt_W12"]).max()
    np.testing.assert_allclose(coi, g["xwt_coi"], rtol=1e-14)
    np.testing.assert_allclose(freq, g["xwt_freq"], rtol=1e-14)
    np.testing.assert_allclose(signif, g["xwt_signif"], rtol=1e-12)
    W12u, _, _, sigu = pycwt_amd.xwt(g["y1"], g["y2"], float(g["dt"]), float(g["dj"]), significance_level=0.9,
                                     normalize=False)
    assert np.abs(W12u - g["xwt_W12_unnorm"]).max() < 1e-11 * np.abs(g["xwt_W12_unnorm"]).max()
    np.testing.assert_allclose(sigu, g["xwt_signif_unnorm"], rtol=1e-12)


def test_morlet_smooth_complex_and_real(emulated, g):
    m = pycwt_amd.Morlet(6)
    dt, dj, sj = float(g["dt"]), float(g["dj"]), g["sj"]
    W = pycwt_amd.cwt(g["y1"], dt, dj, -1, -1, m)[0]
    sc = m.smooth(W / sj[:, None], dt, dj, sj)
    assert np.iscomplexobj(sc) and np.abs(sc - g["smooth_complex"]).max() < 1e-12 * np.abs(g["smooth_complex"]).max()
    sr = m.smooth(np.abs(W) ** 2 / sj[:, None], dt, dj, sj)
    assert not np.iscomplexobj(sr)
    assert np.abs(sr - g["smooth_real"]).max() < 1e-12 * np.abs(g["smooth_real"]).max()


def test_wct_without_monte_carlo(emulated, g):
    WCT, aWCT, coi, freq, sig = pycwt_amd.wct(g["y1"], g["y2"], float(g["dt"]), float(g["dj"]), -1, -1, False,
                                              0.95, pycwt_amd.Morlet(6), True)
    assert WCT.shape == g["wct"].shape and WCT.dtype == np.float64
    assert np.abs(WCT - g["wct"]).max() < 1e-10
    d = np.angle(np.exp(1j * (aWCT - g["awct"])))
    assert np.abs(d).max() < 1e-9
    np.testing.assert_allclose(coi, g["wct_coi"], rtol=1e-14)
    np.testing.assert_allclose(freq, g["wct_freq"], rtol=1e-14)
    assert sig.shape == (1,) and sig[0] == 0
    assert WCT.min() >= 0 and WCT.max() <= 1 + 1e-12
    with pytest.raises(AttributeError):                # only Morlet defines smooth (mothers.py:61)
        pycwt_amd.wct(g["y1"], g["y2"], 0.5, wavelet="paul", sig=False)


def test_wct_significance_monte_carlo_small(emulated, tmp_path, monkeypatch):
    """Statistical check of the Monte-Carlo path on a tiny problem + cache round trip."""
    from pycwt_amd import wavelet
    monkeypatch.setattr(wavelet, "get_cache_dir", lambda: str(tmp_path) + "/")
    np.random.seed(3)
    kw = dict(dt=1.0, dj=0.5, s0=2.0, J=6, mc_count=12, progress=False, wavelet="morlet")
    sig = pycwt_amd.wct_significance(0.3, 0.5, **kw)
    assert sig.shape == (7,)
    finite = np.isfinite(sig)
    assert finite.any() and (sig[finite] > 0.3).all() and (sig[finite] <= 1).all()
    again = pycwt_amd.wct_significance(0.3, 0.5, **kw)          # served from the cache file
    np.testing.assert_allclose(again, sig, equal_nan=True)
    assert len(list(tmp_path.glob("wct_sig_*_Morlet.gz"))) == 1


def _seeded_significance_cases(precision=64):
    """wct_significance under np.random.seed against values the unmodified reference produced with the same seed
    (oracle/gen_golden.py: mc_significance): same NaN pattern, every finite level within one histogram bin."""
    mc = load_golden("mc_significance")
    for i in range(int(mc["n_cases"])):
        c = {k[len(f"c{i}_"):]: mc[k] for k in mc.files if k.startswith(f"c{i}_")}
        np.random.seed(int(c["seed"]))
        sig = pycwt_amd.wct_significance(float(c["al1"]), float(c["al2"]), dt=float(c["dt"]), dj=float(c["dj"]),
                                         s0=float(c["s0"]), J=int(c["J"]), significance_level=0.95, wavelet="morlet",
                                         mc_count=int(c["mc_count"]), progress=False, cache=False,
                                         precision=precision)
        ref = c["sig95"]
        assert sig.shape == ref.shape
        np.testing.assert_array_equal(np.isnan(sig), np.isnan(ref))
        ok = ~np.isnan(ref)
        assert ok.sum() >= 5
        assert np.abs(sig[ok] - ref[ok]).max() <= 1.0e-3 + 1e-12, (i, sig, ref)


def test_wct_significance_seed_for_seed_with_the_reference(emulated):
    _seeded_significance_cases()


def _histogram_case(lib, precision):
    """cwt_coherence_histogram against numpy on a matrix with NaNs, negatives and values >= 1."""
    from pycwt_amd import _hip
    rng = np.random.default_rng(11)
    rows, n, nbins = 5, 3000, 1000
    real = np.float64 if precision == 64 else np.float32
    r2 = rng.random((rows, n)).astype(real)
    r2[0, ::7] = np.nan
    r2[1, ::5] = 1.0            # floor(1.0 * 1000) = 1000: skipped (the reference would raise IndexError)
    r2[2, ::3] = -0.25
    r2[3, 10] = real(0.999999)
    lo = np.array([0, 100, 1499, 2999, 7], dtype=np.int64)
    hi = np.array([n, 2900, 1500, 2999, 8], dtype=np.int64)     # full row, interior, 1 column, empty, 1 column
    plan = _hip.Plan(4096, precision, max_rows=8, lib=lib)
    bufs = [_hip.DeviceBuffer(b, lib=lib) for b in (r2.nbytes, lo.nbytes, hi.nbytes, rows * nbins * 8)]
    try:
        for b, a in zip(bufs, (r2, lo, hi, np.zeros((rows, nbins), dtype=np.uint64))):
            b.upload(plan, a)
        for _ in range(2):                                       # accumulates across calls
            plan.coherence_histogram(bufs[0].ptr, n, rows, bufs[1].ptr, bufs[2].ptr, int((hi - lo).max()), nbins,
                                     bufs[3].ptr)
        got = bufs[3].download(plan, (rows, nbins), np.uint64)
    finally:
        for b in bufs:
            b.free()
        plan.close()
    want = np.zeros((rows, nbins), dtype=np.uint64)
    for s in range(rows):
        with np.errstate(invalid="ignore"):
            v = np.floor(r2[s, lo[s]:hi[s]] * real(nbins))
        v = v[(v >= 0) & (v < nbins)].astype(int)
        want[s] = 2 * np.bincount(v, minlength=nbins)
    np.testing.assert_array_equal(got, want)
    assert got[3].sum() == 0 and got[2].sum() == 2 and got[4].sum() == 2


@pytest.mark.parametrize("precision", [64, 32])
def test_coherence_histogram_matches_numpy(emu_library, precision):
    _histogram_case(emu_library, precision)


def _boxcar_case(lib, precision):
    """cwt_boxcar_scales == scipy.signal.convolve2d(T, win[:, None], 'same') for window lengths on both sides of
    the sliding-window kernel's limit, rows not a multiple of its strip height, ragged column count."""
    from scipy.signal import convolve2d
    from pycwt_amd import _hip
    rng = np.random.default_rng(4)
    rows, n = 70, 300
    cplx = np.complex128 if precision == 64 else np.complex64
    T = (rng.standard_normal((rows, n)) + 1j * rng.standard_normal((rows, n))).astype(cplx)
    plan = _hip.Plan(512, precision, max_rows=128, lib=lib)
    a, b = _hip.DeviceBuffer(T.nbytes, lib=lib), _hip.DeviceBuffer(T.nbytes, lib=lib)
    try:
        a.upload(plan, T)
        for L in (1, 2, 3, 14, 16, 17, 33, 70):
            win = rng.random(L)
            plan.boxcar_scales(a.ptr, rows, n, n, win, b.ptr)
            got = b.download(plan, (rows, n), cplx)
            want = convolve2d(T.astype(np.complex128), win[:, None], "same")
            np.testing.assert_allclose(got, want, rtol=0, atol=(1e-12 if precision == 64 else 2e-5) * L)
    finally:
        a.free(); b.free(); plan.close()


@pytest.mark.parametrize("precision", [64, 32])
def test_boxcar_matches_convolve2d(emu_library, precision):
    _boxcar_case(emu_library, precision)


def test_device_resident_xwt_and_wct_give_the_host_results(emulated, g):
    """`wct_device` / `xwt_device`: the same kernels as `wct(sig=False)` / `xwt`, the result matrices left on the GPU until
    asked for (the download was 50 of the 52 / 61 ms of those calls at two 2^20-point series)."""
    dt, dj = float(g["dt"]), float(g["dj"])
    WCT, aWCT, coi, freq, _ = pycwt_amd.wct(g["y1"], g["y2"], dt, dj, -1, -1, False, 0.95, pycwt_amd.Morlet(6), True)
    with pycwt_amd.wct_device(g["y1"], g["y2"], dt, dj, wavelet=pycwt_amd.Morlet(6)) as D:
        assert D.shape == WCT.shape and D.wct_ptr and D.angle_ptr
        assert np.array_equal(D.wct(), WCT) and np.array_equal(D.angle(), aWCT)
        np.testing.assert_array_equal(D.coi, coi)
        np.testing.assert_array_equal(D.freq, freq)
    W12, coi2, freq2, signif = pycwt_amd.xwt(g["y1"], g["y2"], dt, dj=dj, wavelet="morlet")
    T, signif_d = pycwt_amd.xwt_device(g["y1"], g["y2"], dt, dj=dj, wavelet="morlet")
    try:
        assert np.array_equal(T.W(), W12)
        np.testing.assert_array_equal(T.coi, coi2)
        np.testing.assert_array_equal(signif_d, signif)
    finally:
        T.close()
    with pytest.raises(ValueError):
        pycwt_amd.xwt_device(g["y1"], g["y2"][:-1], dt)


@pytest.mark.parametrize("surrogates,al", [("reference", (0.3, 0.5)), ("ar1", (0.6, 0.8))])
def test_wct_significance_with_surrogates_made_on_the_device(emulated, tmp_path, monkeypatch, surrogates, al):
    """`rng="device"`: Philox surrogates instead of NumPy's.  Another generator, the same distributions: the 95 % levels of
    40 device draws against 40 NumPy draws of the same problem agree within the Monte-Carlo error (the levels of two
    independent NumPy runs differ by as much: checked here too, as the yardstick), are reproducible per seed, follow
    np.random.seed when no seed is given, and cache under their own file name."""
    from pycwt_amd import wavelet
    monkeypatch.setattr(wavelet, "get_cache_dir", lambda: str(tmp_path) + "/")
    kw = dict(dt=1.0, dj=0.5, s0=2.0, J=6, mc_count=40, progress=False, wavelet="morlet", surrogates=surrogates, cache=False)
    np.random.seed(1)
    host_a = pycwt_amd.wct_significance(*al, **kw)
    np.random.seed(2)
    host_b = pycwt_amd.wct_significance(*al, **kw)
    dev_a = pycwt_amd.wct_significance(*al, rng="device", seed=11, **kw)
    dev_b = pycwt_amd.wct_significance(*al, rng="device", seed=12, **kw)
    ok = np.isfinite(host_a)
    np.testing.assert_array_equal(np.isnan(dev_a), np.isnan(host_a))
    yard = max(np.abs(host_a[ok] - host_b[ok]).max(), np.abs(dev_a[ok] - dev_b[ok]).max(), 0.01)
    assert np.abs(dev_a[ok] - host_a[ok]).max() <= 2.5 * yard, (dev_a, host_a, yard)
    assert np.abs(0.5 * (dev_a + dev_b)[ok] - 0.5 * (host_a + host_b)[ok]).max() <= 2.0 * yard
    np.testing.assert_array_equal(pycwt_amd.wct_significance(*al, rng="device", seed=11, **kw), dev_a)      # reproducible
    assert not np.array_equal(dev_a[ok], dev_b[ok])
    np.random.seed(5)
    s1 = pycwt_amd.wct_significance(*al, rng="device", **dict(kw, mc_count=6))
    np.random.seed(5)
    s2 = pycwt_amd.wct_significance(*al, rng="device", **dict(kw, mc_count=6))
    np.testing.assert_array_equal(s1, s2)
    pycwt_amd.wct_significance(*al, rng="device", seed=3, **dict(kw, mc_count=4, cache=True))
    assert len(list(tmp_path.glob("wct_sig_*_devrng_seed3.gz"))) == 1        # (an explicit seed is part of the cache name)
    with pytest.raises(ValueError):
        pycwt_amd.wct_significance(*al, rng="gpu", **kw)


def test_scratch_of_a_call_is_kept_for_the_next(emulated, monkeypatch):
    """Work matrices go back to a bounded pool instead of to the driver (a Monte-Carlo call's ~60 GB cost seconds to allocate
    and free): the second call allocates nothing, results are the same, the bound and release_scratch() hold."""
    import pycwt_amd
    from pycwt_amd import wavelet as w, _hip
    pycwt_amd.release_scratch()
    made = []
    orig = _hip.DeviceBuffer.__init__

    def counting(self, nbytes, device=0, lib=None):
        made.append(int(nbytes))
        orig(self, nbytes, device, lib)
    monkeypatch.setattr(_hip.DeviceBuffer, "__init__", counting)
    rng = np.random.default_rng(8)
    y1, y2 = rng.standard_normal(600), rng.standard_normal(600)
    a = pycwt_amd.wct(y1, y2, 1.0, dj=0.5, sig=False)
    first = len([n for n in made if n > 256])
    assert first > 0 and w._POOL_HELD[0] > 0
    made.clear()
    b = pycwt_amd.wct(y1, y2, 1.0, dj=0.5, sig=False)
    assert [n for n in made if n > 256] == []                    # every matrix came from the pool
    for u, v in zip(a[:2], b[:2]):
        assert np.array_equal(u, v)
    held = w._POOL_HELD[0]
    assert held == sum(x.nbytes for v in w._POOL.values() for x in v)
    pycwt_amd.release_scratch()
    assert w._POOL_HELD[0] == 0 and not w._POOL
    monkeypatch.setenv("PYCWT_AMD_SCRATCH_POOL_GB", "0")          # keep nothing
    pycwt_amd.wct(y1, y2, 1.0, dj=0.5, sig=False)
    assert w._POOL_HELD[0] == 0 and not any(w._POOL.values())


def test_failed_allocation_empties_the_scratch_pool_and_retries(emulated, monkeypatch):
    """Memory kept from finished calls must never be what makes the next allocation fail: a failing cwt_malloc runs the release
    hook (the pool goes back to the driver) and is tried once more."""
    import pycwt_amd
    from pycwt_amd import wavelet as w, _hip
    rng = np.random.default_rng(9)
    y1, y2 = rng.standard_normal(400), rng.standard_normal(400)
    pycwt_amd.wct(y1, y2, 1.0, dj=0.5, sig=False)
    assert w._POOL_HELD[0] > 0
    lib = _hip.load()
    real, state = lib.cwt_malloc, {"fail": 1, "calls": 0}

    def flaky(device, pptr, nbytes):
        state["calls"] += 1
        if state["fail"]:
            state["fail"] -= 1
            return -2                                   # CWT_ENOMEM
        return real(device, pptr, nbytes)
    monkeypatch.setattr(lib, "cwt_malloc", flaky)
    buf = _hip.DeviceBuffer(1 << 20, lib=lib)
    assert buf.ptr and state["calls"] == 2 and w._POOL_HELD[0] == 0 and not w._POOL
    buf.free()
    state["fail"] = 2                                   # still failing after the release: the error reaches the caller
    with pytest.raises(_hip.HipError):
        _hip.DeviceBuffer(1 << 20, lib=lib)


def test_scratch_pool_is_bounded_by_the_card_and_library_allocations_retry(emulated, monkeypatch):
    """(i) the pool's bound is a fraction of the DEVICE's memory (cwt_device_memory), not a constant: the emulated card
    reports 8 GiB, so the default keeps at most 2 GiB, an absolute cap lowers it, a 64-GB card would never be filled by kept
    scratch; (ii) an allocation that fails INSIDE the library (a plan call returning CWT_ENOMEM) runs the same release hook as a
    failing DeviceBuffer and is retried once."""
    import pycwt_amd
    from pycwt_amd import wavelet as w, _hip
    lib = _hip.load()
    free, total = lib.device_memory(0)
    assert total == 8 << 30 and 0 < free <= total
    monkeypatch.delenv("PYCWT_AMD_SCRATCH_POOL_GB", raising=False)
    monkeypatch.delenv("PYCWT_AMD_SCRATCH_POOL_FRACTION", raising=False)
    assert w._pool_limit(lib, 0) == total // 4
    monkeypatch.setenv("PYCWT_AMD_SCRATCH_POOL_FRACTION", "0.5")
    assert w._pool_limit(lib, 0) == total // 2
    monkeypatch.setenv("PYCWT_AMD_SCRATCH_POOL_GB", "1")
    assert w._pool_limit(lib, 0) == 1 << 30
    monkeypatch.delenv("PYCWT_AMD_SCRATCH_POOL_GB")
    monkeypatch.delenv("PYCWT_AMD_SCRATCH_POOL_FRACTION")
    rng = np.random.default_rng(10)
    y1, y2 = rng.standard_normal(400), rng.standard_normal(400)
    pycwt_amd.wct(y1, y2, 1.0, dj=0.5, sig=False)
    assert w._POOL_HELD[0] > 0
    # a plan call whose first attempt reports CWT_ENOMEM: the pool is released, the call runs again and succeeds
    plan = _hip.Plan(1024, 64, max_rows=4, lib=lib)
    real, state = lib.cwt_execute_host, {"fail": 1, "calls": 0}

    def flaky(*a):
        state["calls"] += 1
        if state["fail"]:
            state["fail"] -= 1
            return -3                                   # CWT_ENOMEM
        return real(*a)
    monkeypatch.setattr(lib, "cwt_execute_host", flaky)
    W, _ = plan.execute_host(rng.standard_normal(1000), orc.MORLET, 6.0, 1.0, np.array([4.0, 8.0]), want_xhat=False)
    assert state["calls"] == 2 and w._POOL_HELD[0] == 0 and not w._POOL and np.isfinite(W).all()
    state["fail"] = 2                                   # failing again after the release: the error reaches the caller, with its code
    with pytest.raises(_hip.HipError) as e:
        plan.execute_host(rng.standard_normal(1000), orc.MORLET, 6.0, 1.0, np.array([4.0, 8.0]), want_xhat=False)
    assert e.value.code == -3
    plan.close()
