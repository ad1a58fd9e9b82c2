"""NumPy prototype of the polynomial ("Taylor") form of the band-limited rows.

A row whose filter lives on the bins k_c + kappa, kappa in [-B/2, B/2), is a slowly varying envelope times a carrier:
    W[n] = e^{2 pi i k_c n / N} v(n),   v(n) = (1/N) sum_kappa Y[kappa] e^{2 pi i kappa n / N}.
Cut the row into K' intervals of R = N/K' samples, n = R m + r, u = (2 r + 1)/R - 1 in (-1, 1):
    e^{2 pi i kappa n / N} = e^{2 pi i kappa m / K'} e^{i pi kappa (1 - 1/R) / K'} e^{i theta u},  theta = pi kappa / K'
and with e^{i theta u} = sum_d (i theta)^d u^d / d! truncated at degree D (|theta| <= pi B / (2 K')):
    v(n) ~ sum_d a_d[m] u^d,   a_d = IFFT_K'( Y[kappa] e^{i pi kappa (1 - 1/R)/K'} (i theta)^d / d! )
i.e. D + 1 SHORT inverse FFTs per row (stage 1), then one Horner evaluation and one modulation per output, streamed to
memory with contiguous stores and no tile structure (stage 2).  Prototype only: checks the truncation rule.
"""
import math
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import cwt_oracle as orc  # noqa: E402


def degree_for(theta, eps):
    d, term = 0, 1.0
    while True:
        term *= theta / (d + 1)          # theta^(d+1)/(d+1)!
        if term <= eps:
            return d
        d += 1


def poly_row(xhat, N, s, dt, tau, logKp_max=14, min_logR=6, rho_min=4.0):
    a = s * 2 * math.pi / (N * dt)
    xc = math.sqrt(-2 * math.log(tau * 0.1))
    klo, khi = math.ceil((6 - xc) / a), math.floor((6 + xc) / a)
    klo, khi = max(klo, -N // 2), min(khi, N // 2 - 1)
    B = khi - klo + 1
    kc = (klo + khi) // 2
    logKp = min(max(8, math.ceil(math.log2(rho_min * B))), logKp_max, int(math.log2(N)) - min_logR)
    Kp = 1 << logKp
    if Kp < B:
        return None
    R = N // Kp
    theta_max = math.pi * (B / 2 + 1) / Kp
    D = degree_for(theta_max, tau * 0.1)
    kap = np.arange(klo, khi + 1) - kc
    k = kap + kc
    amp = math.sqrt(2 * math.pi * s / dt) * math.pi ** -0.25
    Y = xhat[k % N] * amp * np.exp(-0.5 * (a * k - 6) ** 2) / N
    Y = Y * np.exp(1j * math.pi * kap * (1 - 1 / R) / Kp)
    theta = math.pi * kap / Kp
    coef = np.empty((D + 1, Kp), complex)
    w = np.ones_like(theta, dtype=complex)
    for d in range(D + 1):
        buf = np.zeros(Kp, complex)
        np.add.at(buf, kap % Kp, Y * w)
        coef[d] = np.fft.ifft(buf) * Kp
        w = w * (1j * theta) / (d + 1)
    n = np.arange(N)
    m, r = n // R, n % R
    u = (2 * r + 1) / R - 1
    p = coef[D][m]
    for d in range(D - 1, -1, -1):
        p = p * u + coef[d][m]
    return p * np.exp(2j * math.pi * ((kc * n) % N) / N), B, Kp, D


if __name__ == "__main__":
    N = 1 << 18
    x = np.random.default_rng(1234).standard_normal(N)
    xhat = np.fft.fft(x)
    m = orc.Mother(orc.MORLET, 6)
    s0 = 2 / m.flambda()
    dj = math.log2(N / s0) / 255
    sj = s0 * 2 ** (np.arange(256) * dj)
    for tau in (1e-9, 1e-16):
        for rho in (2.0, 4.0, 8.0):
            worst = 0
            for j in range(60, 256, 7):
                out = poly_row(xhat, N, sj[j], 1.0, tau, rho_min=rho)
                if out is None:
                    continue
                W, B, Kp, D = out
                ref = orc.cwt_rows(x, 1.0, sj[j:j + 1], m)[0]
                err = np.abs(W - ref).max() / np.abs(ref).max()
                worst = max(worst, err)
                if rho == 4.0:
                    print(f"  tau {tau:g} row {j} B {B} K' {Kp} D {D} err {err:.2e}")
            print(f"tau {tau:g} rho_min {rho}: worst {worst:.2e}")
