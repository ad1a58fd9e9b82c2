"""Host-side statistics helpers used by the callers of the CWT hot path (significance, xwt, wct).

O(N) / O(J) NumPy arithmetic -- these stay on the host by design.  Same
names and call signatures as pycwt/helpers.py:37-236 so that `pycwt.ar1`, `pycwt.helpers.rect`, ...
keep resolving; the latent crashes of the reference noted in SURVEY.md 8a(ix) are not reproduced
(`rednoise` with g == 0, `boxpdf`), its silent defect in `rednoise` (white surrogates) is -- see there.
"""
from __future__ import annotations

import os

import numpy as np
from scipy.signal import lfilter


# The reference's FFT seam (helpers.py:6-30): a module object `fft` with fft/ifft/fftfreq and `fft_kwargs(signal)`
# giving the padded length.  pycwt_amd.cwt does not go through it (the transforms run on the GPU); the names stay so
# that code which reaches for `pycwt.fft` / `pycwt.helpers.fft_kwargs` (leaked into the package namespace by
# `from .wavelet import *`, pycwt/__init__.py:85) keeps resolving, with the scipy.fftpack branch's semantics.
import scipy.fftpack as fft  # noqa: E402

_FFT_NEXT_POW2 = True


def fft_kwargs(signal, **kwargs):
    """Next power of two >= len(signal) as the transform length (helpers.py:27-30)."""
    if _FFT_NEXT_POW2:
        return {"n": int(2 ** np.ceil(np.log2(len(signal))))}


def find(condition):
    """Indices of the true entries of the flattened condition (helpers.py:37-40)."""
    return np.flatnonzero(np.ravel(condition))


def ar1(x):
    """Allen & Smith (1996) lag-1 autoregression fit: returns (g, a, mu2)  (helpers.py:43-104).

    g: lag-one autocorrelation with the finite-sample bias removed (smaller root of the quadratic
    of Grinsted's substitution), a: innovation standard deviation, mu2: expected squared mean of a
    finite AR(1) segment relative to the process variance.  Raises `Warning` (as the reference does)
    when the quadratic has no real root.
    """
    x = np.asarray(x, dtype=float)
    n = x.size
    d = x - x.mean()
    c0 = d.dot(d) / n                      # lag-0 covariance
    c1 = d[:-1].dot(d[1:]) / (n - 1)       # lag-1 covariance
    qa = c0 * n ** 2
    qb = -c1 * n - c0 * n ** 2 - 2 * c0 + 2 * c1 - c1 * n ** 2 + c0 * n
    qc = n * (c0 + c1 * n - c1)
    disc = qb ** 2 - 4 * qa * qc
    if not disc > 0:
        raise Warning("Cannot place an upperbound on the unbiased AR(1). "
                      "Series is too short or trend is to large.")
    g = (-qb - disc ** 0.5) / (2 * qa)
    mu2 = -1 / n + (2 / n ** 2) * ((n - g ** n) / (1 - g) - g * (1 - g ** (n - 1)) / (1 - g) ** 2)
    a = ((1 - g ** 2) * c0 / (1 - mu2)) ** 0.5
    return g, a, mu2


def ar1_spectrum(freqs, ar1=0.):
    """Theoretical power spectrum of an AR(1) process at normalised frequencies (helpers.py:107-143)."""
    z = np.exp(-2j * np.pi * np.asarray(freqs))
    return (1 - ar1 ** 2) / np.abs(1 - ar1 * z) ** 2


def rednoise(N, g, a=1., *, ar1=False):
    """Surrogate noise of length N for the Monte-Carlo tests (helpers.py:146-173), from the global NumPy RNG like
    the reference: N + tau normal deviates (tau = twice the decorrelation time), the first tau discarded.

    Default (`ar1=False`) = what the reference really returns, draw for draw: it hands the (N + tau, 1) column to
    `lfilter([1, 0], [1, -g], .)` WITHOUT `axis=0` (helpers.py:170), so the AR(1) recursion runs along the axis of
    length one, never sees a previous sample, and the output is the white input times `a` (lag-1 correlation 0.01
    where `g` = 0.7 was asked for).  Reproducing that is what makes `wct_significance` agree with the reference seed
    for seed (tests/golden/mc_significance.npz).  `ar1=True` filters along time, i.e. the AR(1) process the name
    promises; `wct_significance(..., surrogates="ar1")` uses it and keeps its cache files apart.
    g == 0 returns white noise (the reference raises AttributeError there: `np.randn`, helpers.py:166)."""
    if g == 0:
        return np.random.randn(N) * a
    tau = int(np.ceil(-2 / np.log(np.abs(g))))
    w = np.random.randn(N + tau, 1)
    if a != 1:
        w *= a
    if ar1:
        return lfilter([1, 0], [1, -g], w, axis=0)[tau:].flatten()
    # lfilter([1, 0], [1, -g], w) along the last axis, which has length one: every sample is a sequence of its own,
    # y[0] = 1 * x[0] + 0 -- the input itself, bit for bit.  Not calling it saves 0.3 s per 6 million samples, which was
    # three quarters of a Monte-Carlo draw at BASELINE config 5 (profiles/r04_wct.txt).
    return w[tau:].reshape(-1)                       # (a view of the draw: no 8 N byte copy per surrogate)


def rect(x, normalize=False):
    """Boxcar with half-weight end taps (helpers.py:176-191).  x: length or shape."""
    if isinstance(x, (int, float)):
        shape = [x, ]
    elif isinstance(x, (list, dict)):
        shape = x
    else:
        shape = np.asarray(x).shape
    X = np.zeros(shape)
    X[0] = X[-1] = 0.5
    X[1:-1] = 1
    if normalize:
        X /= X.sum()
    return X


def boxpdf(x):
    """Map data to its empirical CDF ("boxed" distribution): returns (bX, X, Y) (helpers.py:194-225)."""
    x = np.asarray(x)
    X, counts = np.unique(x, return_counts=True)
    edges = np.concatenate([[0], np.cumsum(counts)])
    Y = 0.5 * (edges[:-1] + edges[1:]) / x.size
    return np.interp(x, X, Y), X, Y


def get_cache_dir():
    """~/.cache/pycwt/, created on first use (helpers.py:228-236)."""
    path = os.path.join(os.path.expanduser("~"), ".cache", "pycwt", "")
    os.makedirs(path, exist_ok=True)
    return path
