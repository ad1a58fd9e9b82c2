"""Pins oracle/cwt_oracle.py (the CPU restatement) to the reference.

(a) fixtures generated from the unmodified reference (oracle/gen_golden.py),
(b) the golden numbers recorded in SURVEY.md section 8c,
(c) the live reference when /root/reference is mounted (build container only).
"""
import os
import sys
import warnings

import numpy as np
import pytest

from conftest import load_golden, row_errors
from oracle import cwt_oracle as orc

MOTHERS = {"morlet": orc.Mother(orc.MORLET, 6), "paul": orc.Mother(orc.PAUL, 4),
           "dog": orc.Mother(orc.DOG, 2)}
TOL = 1e-12     # oracle and reference share pocketfft: agreement is round-off level


def test_nino3_simple_matches_fixture_and_survey_numbers():
    g = load_golden("nino3_simple")
    W, sj, freqs, coi, fft, fftfreqs = orc.cwt(g["x"], 0.25, 1 / 12, 0.5, 84,
                                               orc.Mother(orc.MORLET, 6))
    assert W.shape == (85, 504) and W.dtype == np.complex128
    per_row, l2 = row_errors(W, g["W"])
    assert per_row.max() < TOL and l2 < TOL
    np.testing.assert_allclose(sj, g["sj"], rtol=1e-15)
    np.testing.assert_allclose(freqs, g["freqs"], rtol=1e-15)
    np.testing.assert_allclose(coi, g["coi"], rtol=1e-15)
    np.testing.assert_allclose(fft, g["fft"], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(fftfreqs, g["fftfreqs"], rtol=1e-15)
    # SURVEY.md 8c(4) golden numbers (recorded from the reference by the survey)
    assert abs(g["std"] - 0.7295786149668999) < 1e-15
    assert abs(W[0, 0] - (0.07330141288334911 + 0.1738289767280871j)) < 1e-13
    assert abs(W[42, 252] - (-0.5998903691900097 - 0.9977145969302366j)) < 1e-13
    assert abs(W[84, 503] - (0.4262714280043598 + 0.6072150181842293j)) < 1e-13
    assert abs((np.abs(W) ** 2).sum() - 90362.0554906526) < 1e-7
    assert abs(W.sum() - (129.634658313203 - 226.8848858675642j)) < 1e-9
    assert sj[0] == 0.5 and abs(sj[42] - 5.656854249492381) < 1e-15 and abs(sj[84] - 64.0) < 1e-13
    assert abs(coi[0] - 0.09130902107314805) < 1e-16
    assert abs(fft[0] - (1.491186882417564 - 1.0821823421611552j)) < 1e-13
    iw = orc.icwt(W, sj, 0.25, 1 / 12, orc.Mother(orc.MORLET, 6))
    assert iw.dtype == np.complex128
    np.testing.assert_allclose(iw, g["icwt"], rtol=1e-12, atol=1e-13)
    assert abs(iw[0] - (-0.187029432193866 + 0j)) < 1e-13
    assert abs(np.sqrt(np.mean(np.abs(iw - g["x"]) ** 2)) - 0.1271027510091352) < 1e-12


def test_nino3_default_scales():
    g = load_golden("nino3_default")
    W, sj, freqs, coi, fft, fftfreqs = orc.cwt(g["x"], 0.25, 1 / 12, -1, -1, "morlet")
    assert W.shape == (97, 504)
    assert abs(sj[0] - 0.48400665459719555) < 1e-15
    assert abs(sj[-1] - 123.90570357688206) < 1e-11
    per_row, l2 = row_errors(W, g["W"])
    assert per_row.max() < TOL
    assert abs((np.abs(W) ** 2).sum() - 111890.05833555722) < 1e-6


@pytest.mark.parametrize("name", ["morlet", "paul", "dog"])
def test_small_fixtures_all_mothers(name):
    g = load_golden("small_" + name)
    m = MOTHERS[name]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        W, sj, freqs, coi, fft, fftfreqs = orc.cwt(g["x"], 0.5, 0.25, -1, -1, m)
    assert W.shape == g["W"].shape          # includes the Paul NaN-row drop
    per_row, l2 = row_errors(W, g["W"])
    assert per_row.max() < TOL
    np.testing.assert_allclose(sj, g["sj"], rtol=1e-15)
    iw = orc.icwt(W, sj, 0.5, 0.25, m)
    assert iw.dtype == g["icwt"].dtype
    np.testing.assert_allclose(iw, g["icwt"], rtol=1e-11, atol=1e-12)
    # closed-form drop rule agrees with what the reference really dropped
    sj_all, _ = orc.scale_grid(1000, 0.5, 0.25, -1, -1, m)
    assert (~orc.dropped_rows(sj_all, 0.5, m)).sum() == W.shape[0]


@pytest.mark.parametrize("name", ["morlet", "paul", "dog"])
def test_mid_fixtures_rows(name):
    g = load_golden("mid_" + name)
    x = np.random.default_rng(int(g["seed"])).standard_normal(int(g["N"]))
    W = orc.cwt_rows(x, float(g["dt"]), g["sj"], MOTHERS[name])
    per_row, l2 = row_errors(W, g["W"])
    assert per_row.max() < TOL and l2 < TOL


@pytest.mark.slow
def test_big_fixture_morlet_rows():
    g = load_golden("big_morlet")
    N = int(g["N"])
    x = np.random.default_rng(int(g["seed"])).standard_normal(N)
    W = orc.cwt_rows(x, 1.0, g["sj"], MOTHERS["morlet"])
    err = np.abs(W[:, g["cols"]] - g["Wcols"]).max(axis=1) / g["rowmax"]
    assert err.max() < TOL
    np.testing.assert_allclose(np.abs(W).max(axis=1), g["rowmax"], rtol=1e-12)


def test_cosine_known_answer():
    """closed form for x = cos(w_m n dt), n0 a power of two."""
    N, dt, mm = 4096, 0.5, 37
    n = np.arange(N)
    wm = 2 * np.pi * mm / (N * dt)
    x = np.cos(wm * n * dt)
    for name, m in MOTHERS.items():
        sj = np.array([1.0, 3.0, 10.0, 30.0])
        W = orc.cwt_rows(x, dt, sj, m)
        with np.errstate(all="ignore"):
            pos = np.conj(m.psi_ft(sj * wm))
            neg = np.nan_to_num(np.conj(m.psi_ft(-sj * wm)))
        ref = 0.5 * np.sqrt(2 * np.pi * sj / dt)[:, None] * (
            pos[:, None] * np.exp(1j * wm * n * dt) + neg[:, None] * np.exp(-1j * wm * n * dt))
        assert np.abs(W - ref).max() < 1e-11, name


@pytest.mark.skipif(not os.path.isdir("/root/reference/pycwt"),
                    reason="live reference only exists in the build container")
@pytest.mark.parametrize("name", ["morlet", "paul", "dog", "mexicanhat"])
def test_live_reference(name):
    sys.dont_write_bytecode = True
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import pycwt as ref
        x = np.random.default_rng(3).standard_normal(777)
        out_ref = ref.cwt(x, 0.3, 1 / 6, -1, -1, name)
        out = orc.cwt(x, 0.3, 1 / 6, -1, -1, name)
        for a, b in zip(out, out_ref):
            assert a.shape == b.shape and a.dtype == b.dtype
            np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12)
        if name != "paul":
            np.testing.assert_allclose(orc.icwt(out[0], out[1], 0.3, 1 / 6, name),
                                       ref.icwt(out_ref[0], out_ref[1], 0.3, 1 / 6, name),
                                       rtol=1e-12, atol=1e-13)


def test_unpadded_branch_of_the_reference():
    """oracle.cwt(pad=False) against the reference's pyfftw branch (fft_kwargs -> n = len(signal), helpers.py:15-19;
    fixture made by oracle/gen_golden.py: unpadded)."""
    g = load_golden("unpadded")
    for tag in "abc":
        out = orc.cwt(g[f"{tag}_x"], 0.5, 1 / 4, -1, -1, str(g[f"{tag}_name"]), pad=False)
        per_row, l2 = row_errors(out[0], g[f"{tag}_W"])
        assert out[0].shape == g[f"{tag}_W"].shape and per_row.max() < 1e-12 and l2 < 1e-12
        for got, key in zip(out[1:], ("sj", "freqs", "coi", "fft", "fftfreqs")):
            np.testing.assert_allclose(got, g[f"{tag}_{key}"], rtol=1e-12, atol=1e-13)


SAMPLES = ["mauna", "monsoon", "sunspot", "soi"]


@pytest.mark.parametrize("name", SAMPLES)
def test_reference_sample_datasets(name):
    """sample/sample.py's recipe on the reference's other datasets (sample/dataset.py:68-135; fixtures hold every third
    row of W): lengths 456 / 496 / 992 / 400, dt = 1/12 and 1/4."""
    g = load_golden("sample_" + name)
    W, sj, freqs, coi, fft, fftfreqs = orc.cwt(g["x"], float(g["dt"]), 1 / 12, -1, -1, orc.Mother(orc.MORLET, 6))
    assert W.shape == (int(g["nrows"]), g["x"].size)
    per_row, l2 = row_errors(W[g["rows"]], g["W"])
    assert per_row.max() < TOL and l2 < TOL
    for a, b in ((sj, g["sj"]), (freqs, g["freqs"]), (coi, g["coi"]), (fftfreqs, g["fftfreqs"])):
        np.testing.assert_allclose(a, b, rtol=1e-14)
    np.testing.assert_allclose(fft, g["fft"], rtol=0, atol=1e-13 * np.abs(g["fft"]).max())
    np.testing.assert_allclose(orc.icwt(W, sj, float(g["dt"]), 1 / 12, orc.Mother(orc.MORLET, 6)), g["icwt"], rtol=1e-12, atol=1e-13)


def test_intended_value_oracle_equals_the_reference_arithmetic_where_the_reference_is_finite():
    """oracle.cwt_rows(..., intended=True): Paul's Heaviside applied before the exponential (SURVEY.md 8a quirk iii) -- the
    value the GPU computes on the rows the reference turns into NaN.  On every row the reference KEEPS it must be the
    reference's own arithmetic bit for bit; on the dropped rows it must be finite and continue the kept rows smoothly
    (row energy against the neighbouring kept row: Parseval on the filter)."""
    n = 4096
    x = np.random.default_rng(17).standard_normal(n)
    m = orc.Mother(orc.PAUL, 4)
    s0 = 2 / m.flambda()
    sj = s0 * 2 ** (np.arange(64) * np.log2(n / s0) / 63)
    dropped = orc.dropped_rows(sj, 1.0, m)
    assert dropped.any() and not dropped.all()
    with np.errstate(all="ignore"):
        ref = orc.cwt_rows(x, 1.0, sj, m)
    want = orc.cwt_rows(x, 1.0, sj, m, intended=True)
    assert np.array_equal(ref[~dropped], want[~dropped])
    assert np.isnan(ref[dropped]).all() and np.isfinite(want[dropped]).all()
    # independent check of the dropped rows: direct sum over the positive bins with the filter written out
    k = np.arange(1, n // 2)
    xh = np.fft.fft(x)
    c = 2.0 ** 4 / np.sqrt(4 * 5040.0)
    for j in np.flatnonzero(dropped)[[0, -1]]:
        f = sj[j] * 2 * np.pi * k / n
        F = np.sqrt(sj[j] * 2 * np.pi / n * n) * c * f ** 4 * np.exp(-f)
        row = np.zeros(n, complex)
        row[1:n // 2] = xh[1:n // 2] * F
        direct = np.fft.ifft(row)
        assert np.abs(direct - want[j]).max() <= 1e-12 * np.abs(direct).max()
    for kind, p in ((orc.MORLET, 6), (orc.DOG, 2)):                        # other mothers: the flag changes nothing
        mo = orc.Mother(kind, p)
        assert np.array_equal(orc.cwt_rows(x, 1.0, sj[:8], mo), orc.cwt_rows(x, 1.0, sj[:8], mo, intended=True))
