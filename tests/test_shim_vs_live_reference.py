"""Randomised drop-in check of pycwt_amd.cwt / icwt (kernels on the CPU emulation) against the UNMODIFIED
reference imported from /root/reference -- build container only (skipped elsewhere; the GPU box never
needs the reference)."""
import os
import sys
import warnings

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import pycwt_amd
from conftest import row_errors

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/pycwt"),
                                reason="live reference only exists in the build container")


def _ref():
    sys.dont_write_bytecode = True
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import pycwt
    return pycwt


@st.composite
def calls(draw):
    n0 = draw(st.integers(20, 3000))
    dt = draw(st.sampled_from([0.25, 1.0, 1 / 12, 7.0]))
    dj = draw(st.sampled_from([1 / 12, 0.25, 0.5, 1.0]))
    name = draw(st.sampled_from(["morlet", "paul", "dog", "mexicanhat"]))
    use_defaults = draw(st.booleans())
    s0 = -1 if use_defaults else dt * draw(st.floats(0.5, 8.0))
    J = -1 if use_defaults else draw(st.integers(1, 40))
    use_freqs = draw(st.booleans()) and not use_defaults
    return n0, dt, dj, name, s0, J, use_freqs, draw(st.integers(0, 2 ** 31))


@settings(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck))
@given(calls())
def test_cwt_and_icwt_match_reference_call_for_call(emulated, call):
    n0, dt, dj, name, s0, J, use_freqs, seed = call
    ref = _ref()
    x = np.random.default_rng(seed).standard_normal(n0).cumsum() * 0.1 + np.random.default_rng(seed + 1).standard_normal(n0)
    kw = {}
    if use_freqs:
        kw["freqs"] = np.geomspace(0.4 / dt, 1.0 / (n0 * dt), 7)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out_ref = ref.cwt(x, dt, dj, s0, J, name, **kw)
    out = pycwt_amd.cwt(x, dt, dj, s0, J, name, **kw)
    assert len(out) == 6
    for a, b in zip(out, out_ref):
        assert a.shape == b.shape and a.dtype == b.dtype
    W, Wr = out[0], out_ref[0]
    if np.isnan(Wr).any():          # every row NaN: the reference keeps them (wavelet.py:112); nothing to compare
        return
    scale = np.abs(Wr).max(axis=1)
    floor = 1e-15 * np.sqrt(2 * np.pi * out_ref[1] / dt) * np.abs(x).sum()
    assert (np.abs(W - Wr).max(axis=1) <= 1e-11 * scale + floor).all()
    for a, b in zip(out[1:], out_ref[1:]):
        np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-11 * max(1.0, np.abs(b).max()))
    mother = {"morlet": ref.Morlet(), "paul": ref.Paul(), "dog": ref.DOG(), "mexicanhat": ref.MexicanHat()}[name]
    if mother.cdelta != -1 and W.shape[0] > 1:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            iw_ref = ref.icwt(Wr, out_ref[1], dt, dj, name)
        iw = pycwt_amd.icwt(W, out[1], dt, dj, name)
        assert iw.dtype == iw_ref.dtype
        np.testing.assert_allclose(iw, iw_ref, rtol=1e-9, atol=1e-10 * max(1.0, np.abs(iw_ref).max()))


@pytest.mark.parametrize("name", ["morlet", "paul", "dog"])
@pytest.mark.parametrize("bad", [np.nan, np.inf])
def test_non_finite_sample_against_the_live_reference(emulated, monkeypatch, name, bad):
    """One NaN / inf sample: the unmodified reference returns an all-NaN W with every row kept (wavelet.py:91, :111-115);
    so does the shim, also at a length where clean signals use the overlap-save rows."""
    from pycwt_amd import wavelet
    monkeypatch.setattr(wavelet, "PLAN_OPTIONS", {"ols_min_logn": 15})
    ref = _ref()
    x = np.random.default_rng(5).standard_normal(33000)
    x[777] = bad
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out_ref = ref.cwt(x, 0.5, 0.5, -1, -1, name)
    out = pycwt_amd.cwt(x, 0.5, 0.5, -1, -1, name)
    assert out[0].shape == out_ref[0].shape
    assert np.isnan(out_ref[0]).all() and np.isnan(out[0]).all()
    for a, b in zip(out[1:4], out_ref[1:4]):
        np.testing.assert_allclose(a, b, rtol=1e-12)
    assert not np.isfinite(out[4]).any() and not np.isfinite(out_ref[4]).any()
