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


if __name__ == "__main__":
    nino3()
    small()
    mid()
    big()
