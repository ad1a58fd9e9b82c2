"""NumPy prototype of the analytic-signal overlap-save form for rows whose filter is clipped at Nyquist.

W_j = IFFT_N(xhat * F_j), F_j = amp * G(s w) on w in [-pi, pi).  Where G has not died out at the Nyquist bins the
cyclic filter jumps there and h_j = IFFT(F_j) has a 1/t tail: no overlap-save on the real signal.  But
    xhat * F_j = (xhat * mask) * E_j,   mask = bins in (w_s, pi),  E_j = G(s w) u(w) on w in (w_s, w_s + 2 pi)
with G continued PAST Nyquist (no wrap) and a smooth window u that is 1 on the mask and falls to 0 over the rest of the
circle: E_j is cyclically smooth, its kernel e_j is short, and W_j = x_M (*) e_j is an overlap-save convolution of the
complex band-passed signal x_M = IFFT_N(xhat * mask), computed once per transform.
Prototype only (tools/): checks the error of that form against the direct N-point transform.
"""
import math
import sys

import numpy as np
from scipy.special import erfc

sys.path.insert(0, ".")
from oracle import cwt_oracle as orc  # noqa: E402


def profile(kind, p, f):
    if kind == orc.MORLET:
        return np.exp(-0.5 * (f - p) ** 2)
    if kind == orc.PAUL:
        with np.errstate(all="ignore"):
            return np.where(f > 0, np.abs(f) ** p * np.exp(-np.abs(f)), 0.0)
    return f ** p * np.exp(-0.5 * f * f)


def window(kappa, k_one_lo, k_one_hi, k_end_lo, k_end_hi, tail=1e-18):
    """1 on [k_one_lo, k_one_hi], erfc tapers down to `tail` at k_end_lo / k_end_hi (bins, unwrapped)."""
    z = math.sqrt(-math.log(tail))     # erfc(z) ~ tail
    u = np.ones_like(kappa, dtype=float)
    hi = kappa > k_one_hi
    c, hw = 0.5 * (k_one_hi + k_end_hi), 0.5 * (k_end_hi - k_one_hi)
    u[hi] = 0.5 * erfc((kappa[hi] - c) / hw * z)
    lo = kappa < k_one_lo
    if k_end_lo < k_one_lo:
        c, hw = 0.5 * (k_one_lo + k_end_lo), 0.5 * (k_one_lo - k_end_lo)
        u[lo] = 0.5 * erfc((c - kappa[lo]) / hw * z)
    else:
        u[lo] = 0.0
    return u


def run(kind, param, logN, rows, tau, P=4096, H=64, dtype=np.float64):
    N = 1 << logN
    mother = orc.Mother(kind, param)
    rng = np.random.default_rng(1234)
    x = rng.standard_normal(N).astype(dtype)
    dt = 1.0
    s0 = 2 * dt / mother.flambda()
    dj = math.log2(N * dt / s0) / 255
    sj = s0 * 2 ** (np.arange(256) * dj)
    sel = sj[rows]
    Wref = orc.cwt_rows(x.astype(np.float64), dt, sel, mother)
    xhat = np.fft.fft(x.astype(np.float64))
    # mask: signed bins (k_s, N/2); Morlet: k_s from the support threshold of the smallest scale, others: 1
    a_min = sel.min() * 2 * math.pi / (N * dt)
    if kind == orc.MORLET:
        c_lo = (param - math.sqrt(-2 * math.log(tau * 0.1))) / a_min       # bins (negative)
        k_one_lo = math.floor(c_lo)
        k_s = k_one_lo - N // 16                                            # low taper: N/16 bins
    else:
        k_one_lo, k_s = 1, 1
    k_hi = N // 2 - 1
    ks = np.fft.fftfreq(N, 1.0 / N).astype(int)                             # signed bins, Nyquist negative
    mask = (ks >= k_s) & (ks <= k_hi)
    xm = np.fft.ifft(xhat * mask)                                           # includes 1/N
    L = P - 2 * H
    nblk = -(-N // L)
    errs = []
    for j, s in enumerate(sel):
        # block grid: bin k' of a P-point block <-> bin k' N / P; unwrapped kappa in [k_s', k_s' + P)
        ksp = math.floor(k_s * P / N)
        q = np.arange(P)
        kappa = ksp + ((q - ksp) % P)
        ab = s * 2 * math.pi / (P * dt)
        amp = math.sqrt(2 * math.pi * s / dt) * {orc.MORLET: math.pi ** -0.25,
                                                 orc.PAUL: 2.0 ** param / math.sqrt(param * math.factorial(2 * int(param) - 1)),
                                                 orc.DOG: 1.0}[kind]
        E = amp * profile(kind, param, ab * kappa) * window(kappa.astype(float), k_one_lo * P / N, P // 2,
                                                            ksp, ksp + P)
        W = np.empty(N, complex)
        for b in range(nblk):
            idx = (b * L - H + np.arange(P)) % N
            y = np.fft.ifft(np.fft.fft(xm[idx]) * E)
            n0 = b * L
            n1 = min(N, n0 + L)
            W[n0:n1] = y[H:H + (n1 - n0)]
        err = np.max(np.abs(W - Wref[j])) / np.max(np.abs(Wref[j]))
        # kernel tail: e = ifft(E), mass beyond H
        e = np.abs(np.fft.ifft(E))
        t = np.minimum(np.arange(P), P - np.arange(P))
        tail = e[t > H].sum() / e.sum()
        errs.append((rows[j], s, err, tail))
    return errs


if __name__ == "__main__":
    for kind, param, name in ((orc.MORLET, 6, "morlet"), (orc.PAUL, 4, "paul")):
        for tau in (1e-9, 1e-16):
            rows = list(range(0, 20, 2)) if kind == orc.MORLET else list(range(0, 48, 4))
            print(name, "tau", tau)
            for r in run(kind, param, 16, rows, tau, H=64 if kind == orc.MORLET else 192):
                print("  row %3d s %.3f err %.2e tailmass %.2e" % r)
