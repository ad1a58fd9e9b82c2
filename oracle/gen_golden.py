#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the UNMODIFIED reference.

Runs only in the build container (needs /root/reference); the GPU box and the
test-suite consume the committed fixtures.  Usage:

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

Cases:
  nino3_simple   sample/simple_sample.py:29-60 recipe on sst_nino3.dat (C1)
  nino3_default  sample/sample.py recipe ((x-mean)/std, s0 = J = -1)
  small_<mother> n0 = 1000 (Npad 1024) seeded noise, default scales, 3 mothers
  mid_<mother>   N = 2**13 seeded noise, 16 rows spanning s0..N*dt, 3 mothers
  big_morlet     N = 2**20, the 256-row grid of BASELINE config 2, 7 rows
  big_paul/dog   same N, config 3 grids (fp64 values), 5 rows each
  callers        significance / xwt / Morlet.smooth / wct (deterministic part)
  mc_significance  wct_significance with np.random.seed (two small cases) + rednoise seed for seed
  unpadded       the pyfftw branch (transform length = len(signal), helpers.py:15-19) at n0 = 504, 1000, 331
  sample_<name>  the sample/sample.py recipe on the reference's other datasets (sample/dataset.py:68-135): mauna,
                 monsoon, sunspot, soi -- anomaly / std, default s0 and J, Morlet(6); every third row of W is kept
The signal itself is stored too (NINO3 data is 504 floats).
"""
import os
import sys
import warnings

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
warnings.simplefilter("ignore")
import pycwt as ref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def mother_of(name):
    return {"morlet": ref.Morlet(6), "paul": ref.Paul(4), "dog": ref.DOG(2)}[name]


def save(name, **kw):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **kw)
    print(name, {k: getattr(v, "shape", v) for k, v in kw.items()},
          os.path.getsize(path) // 1024, "KiB")


def nino3():
    dat = np.loadtxt("/root/reference/pycwt/sample/sst_nino3.dat")
    dt, t0 = 0.25, 1871.0
    t = np.arange(dat.size) * dt + t0
    p = np.polyfit(t - t0, dat, 1)
    x = (dat - np.polyval(p, t - t0))
    std = x.std()
    x = x / std
    W, sj, freqs, coi, fft, fftfreqs = ref.cwt(x, dt, 1 / 12, 0.5, 84, ref.Morlet(6))
    iW = ref.icwt(W, sj, dt, 1 / 12, ref.Morlet(6))
    save("nino3_simple", raw=dat, x=x, dt=dt, dj=1 / 12, s0=0.5, J=84, W=W, sj=sj,
         freqs=freqs, coi=coi, fft=fft, fftfreqs=fftfreqs, icwt=iW, std=std)
    x2 = (dat - dat.mean()) / dat.std()
    W, sj, freqs, coi, fft, fftfreqs = ref.cwt(x2, dt, 1 / 12, -1, -1, "morlet")
    iW = ref.icwt(W, sj, dt, 1 / 12, "morlet")
    save("nino3_default", x=x2, dt=dt, dj=1 / 12, W=W, sj=sj, freqs=freqs, coi=coi,
         fft=fft, fftfreqs=fftfreqs, icwt=iW)


def datasets():
    """sample/sample.py:41-72 on the datasets of sample/dataset.py other than NINO3 (different lengths and sampling
    steps: 456 x 1/12, 496 x 1/4, 992 x 1/4, 399 x 1/4)."""
    for name, dt in (("mauna", 0.08333333), ("monsoon", 0.25), ("sunspot", 0.25), ("soi", 0.25)):
        dat = np.loadtxt("/root/reference/pycwt/sample/%s.dat" % name)
        x = (dat - dat.mean()) / dat.std()
        m = ref.Morlet(6)
        W, sj, freqs, coi, fft, fftfreqs = ref.cwt(x, dt, 1 / 12, -1, -1, m)
        iW = ref.icwt(W, sj, dt, 1 / 12, m)
        save("sample_" + name, x=x, dt=dt, dj=1 / 12, rows=np.arange(0, W.shape[0], 3), W=W[::3], nrows=W.shape[0],
             sj=sj, freqs=freqs, coi=coi, fft=fft, fftfreqs=fftfreqs, icwt=iW)


def small():
    rng = np.random.default_rng(7)
    x = rng.standard_normal(1000)
    for name in ("morlet", "paul", "dog"):
        m = mother_of(name)
        W, sj, freqs, coi, fft, fftfreqs = ref.cwt(x, 0.5, 1 / 4, -1, -1, m)
        iW = ref.icwt(W, sj, 0.5, 1 / 4, m)
        save("small_" + name, x=x, dt=0.5, dj=0.25, W=W, sj=sj, freqs=freqs,
             coi=coi, fft=fft, fftfreqs=fftfreqs, icwt=iW)


def subset_rows(x, dt, m, sj_all, idx):
    """Rows idx of the full transform via the reference's own freqs= argument."""
    f = 1 / (m.flambda() * sj_all[idx])
    W, sj, freqs, coi, fft, fftfreqs = ref.cwt(x, dt, freqs=f, wavelet=m)
    return W, sj


def grid(N, dt, m, rows):
    s0 = 2 * dt / m.flambda()
    dj = np.log2(N * dt / s0) / (rows - 1)
    return s0 * 2 ** (np.arange(rows) * dj), s0, dj


def kept(m, sj, dt):
    """Rows whose filter is NaN-free at the most negative frequency (-pi/dt)."""
    with np.errstate(all="ignore"):
        return ~np.isnan(m.psi_ft(sj * (-np.pi / dt)))


def mid():
    N = 2 ** 13
    x = np.random.default_rng(11).standard_normal(N)
    for name in ("morlet", "paul", "dog"):
        m = mother_of(name)
        sj_all, s0, dj = grid(N, 1.0, m, 16)
        idx = np.arange(16)
        idx = idx[kept(m, sj_all, 1.0)]   # Paul: rows the reference keeps
        W, sj = subset_rows(x, 1.0, m, sj_all, idx)
        save("mid_" + name, seed=11, N=N, dt=1.0, rows=idx, sj=sj, W=W)


def big():
    N = 2 ** 20
    x = np.random.default_rng(1234).standard_normal(N)
    cols = np.r_[0:64, N // 2 - 32:N // 2 + 32, N - 64:N, 12345:12345 + 64]
    for name, rows in (("morlet", [0, 1, 64, 128, 200, 240, 255]),
                       ("paul", [0, 17, 50, 80, 95]),
                       ("dog", [0, 40, 128, 200, 255])):
        m = mother_of(name)
        sj_all, s0, dj = grid(N, 1.0, m, 256)
        rows = np.array(rows)
        rows = rows[kept(m, sj_all[rows], 1.0)]   # Paul: rows the reference keeps
        W, sj = subset_rows(x, 1.0, m, sj_all, rows)
        # store a column subset plus per-row norms: full rows would be 100+ MB
        save("big_" + name, seed=1234, N=N, dt=1.0, s0=s0, dj=dj, rows=rows,
             sj=sj, cols=cols, Wcols=W[:, cols], rowmax=np.abs(W).max(axis=1),
             rowsum=W.sum(axis=1), rowl2=np.sqrt((np.abs(W) ** 2).sum(axis=1)))


def callers():
    """Fixtures for the callers of the hot path: significance, xwt, Morlet.smooth, wct (sig=False)."""
    import pycwt.helpers as rh
    rng = np.random.default_rng(21)
    n = 300
    e1, e2 = rng.standard_normal(n), rng.standard_normal(n)
    y1, y2 = np.zeros(n), np.zeros(n)
    for i in range(1, n):                          # two correlated AR(1) series with a common 16-sample cycle
        y1[i] = 0.6 * y1[i - 1] + e1[i]
        y2[i] = 0.4 * y2[i - 1] + 0.5 * e1[i] + e2[i]
    y1 += 2 * np.sin(2 * np.pi * np.arange(n) / 16.0)
    y2 += 1.5 * np.sin(2 * np.pi * np.arange(n) / 16.0 + 1.0)
    dt, dj = 0.5, 1 / 6
    m = ref.Morlet(6)
    W, sj, freqs, coi, _, _ = ref.cwt(y1, dt, dj, -1, -1, m)
    out = dict(y1=y1, y2=y2, dt=dt, dj=dj, sj=sj,
               ar1_y1=np.array(rh.ar1(y1)), ar1_y2=np.array(rh.ar1(y2)),
               ar1_spec=rh.ar1_spectrum(freqs * dt, 0.55), rect7=rh.rect(7, normalize=True), rect1=rh.rect(1, True))
    s0, f0 = ref.significance(y1, dt, sj, 0, None, 0.95, -1, m)
    out.update(sig0=s0, fft0=f0)
    dof = y1.size - sj                                # as sample/simple_sample.py:79-81
    s1, f1 = ref.significance(y1.std() ** 2, dt, sj, 1, 0.6, 0.95, dof, m)
    out.update(sig1=s1, fft1=f1)
    s2, f2 = ref.significance(y1.std() ** 2, dt, sj, 2, 0.6, 0.95, [2, 8], m)
    out.update(sig2=np.atleast_1d(s2), fft2=np.atleast_1d(f2))
    W12, xcoi, xfreq, xsig = ref.xwt(y1, y2, dt, dj, -1, -1, 0.95, m, True)
    out.update(xwt_W12=W12, xwt_coi=xcoi, xwt_freq=xfreq, xwt_signif=xsig)
    W12u, _, _, xsigu = ref.xwt(y1, y2, dt, dj, -1, -1, 0.9, "morlet", False)
    out.update(xwt_W12_unnorm=W12u, xwt_signif_unnorm=xsigu)
    out["smooth_complex"] = m.smooth(W / sj[:, None], dt, dj, sj)
    out["smooth_real"] = m.smooth(np.abs(W) ** 2 / sj[:, None], dt, dj, sj)
    WCT, aWCT, wcoi, wfreq, wsig = ref.wct(y1, y2, dt, dj, -1, -1, False, 0.95, m, True)
    out.update(wct=WCT, awct=aWCT, wct_coi=wcoi, wct_freq=wfreq, wct_sig=wsig)
    save("callers", **out)


def unpadded():
    """The reference's pyfftw branch (helpers.py:15-19): `fft_kwargs` returns n = len(signal), so nothing is padded.
    pyfftw is not installed here; the branch is reproduced by giving the unmodified `wavelet.cwt` exactly that
    `fft_kwargs` (the FFT itself stays scipy.fftpack, the same arithmetic as pyfftw's scipy_fftpack interface)."""
    import pycwt.wavelet as rw
    keep = rw.fft_kwargs
    rw.fft_kwargs = lambda signal, **kw: {"n": len(signal)}
    try:
        out = {}
        for tag, n0, name in (("a", 504, "morlet"), ("b", 1000, "paul"), ("c", 331, "dog")):
            x = np.random.default_rng(n0).standard_normal(n0)
            W, sj, freqs, coi, fft, fftfreqs = ref.cwt(x, 0.5, 1 / 4, -1, -1, mother_of(name))
            out.update({f"{tag}_x": x, f"{tag}_W": W, f"{tag}_sj": sj, f"{tag}_freqs": freqs, f"{tag}_coi": coi,
                        f"{tag}_fft": fft, f"{tag}_fftfreqs": fftfreqs, f"{tag}_name": name})
        save("unpadded", **out)
    finally:
        rw.fft_kwargs = keep


def mc_significance():
    """Seeded Monte-Carlo coherence significance (wavelet.py:531-647): the reference draws from the global NumPy RNG
    (helpers.py:170), one series BEFORE the loop (wavelet.py:594) and two per draw, so a seed pins the whole result."""
    cases = []
    for seed, (al1, al2, dt, dj, s0, J, mc) in enumerate([(0.72, 0.45, 0.25, 0.25, 0.5, 24, 30),
                                                           (0.30, 0.30, 1.0, 0.5, 2.0, 10, 40)]):
        np.random.seed(100 + seed)
        sig = ref.wct_significance(al1, al2, dt=dt, dj=dj, s0=s0, J=J, significance_level=0.95,
                                   wavelet=ref.Morlet(6), mc_count=mc, progress=False, cache=False)
        cases.append(dict(seed=100 + seed, al1=al1, al2=al2, dt=dt, dj=dj, s0=s0, J=J, mc_count=mc, sig95=sig))
    # the surrogate generator itself, seed for seed (helpers.py:146-173)
    import pycwt.helpers as rh
    np.random.seed(5)
    r1 = rh.rednoise(400, 0.72, 1)
    r2 = rh.rednoise(100, 0.3, 2.0)
    save("mc_significance", n_cases=len(cases), rednoise_seed=5, rednoise_a=r1, rednoise_b=r2,
         **{f"c{i}_{k}": v for i, c in enumerate(cases) for k, v in c.items()})


if __name__ == "__main__":
    if "--mc-only" in sys.argv:
        mc_significance()
        sys.exit(0)
    if "--unpadded-only" in sys.argv:
        unpadded()
        sys.exit(0)
    if "--datasets-only" in sys.argv:
        datasets()
        sys.exit(0)
    callers()
    sys.exit(0) if "--callers-only" in sys.argv else None
    mc_significance()
    unpadded()
    nino3()
    datasets()
    small()
    mid()
    big()
