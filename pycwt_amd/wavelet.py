"""``cwt`` / ``icwt`` with the call signatures of regeirk/pycwt (pycwt/wavelet.py:13, :127),
computed by the HIP engine (libcwt_hip.so) on an MI355X.

What stays in Python is exactly what the reference does with O(J) / O(N) host arithmetic around the
hot loops: the scale grid (wavelet.py:75-88), the NaN-row rule for the Paul mother
(wavelet.py:111-115), the cone of influence (wavelet.py:120-121) and the layout of the 6-tuple
(wavelet.py:123-124).  The three hot lines -- forward FFT (:91), filter bank (:102-104), batched
inverse FFT (:105-106) -- and the eq.-11 reduction of ``icwt`` (:169-170) run on the GPU.  There is
no CPU fallback.
"""
from __future__ import annotations

import os

import numpy as np

from . import _hip
from .mothers import DOG, MexicanHat, Morlet, Paul

_MOTHERS = {"morlet": Morlet, "paul": Paul, "dog": DOG, "mexicanhat": MexicanHat}
_plans: dict = {}


def _check_parameter_wavelet(wavelet):
    """wavelet.py:650-663: lower-case name -> default instance (KeyError if unknown); objects pass."""
    if isinstance(wavelet, str):
        return _MOTHERS[wavelet]()
    return wavelet


def _default_precision() -> int:
    return int(os.environ.get("PYCWT_AMD_PRECISION", "64"))


def _plan(nfft: int, precision: int, device: int, rows: int) -> _hip.Plan:
    key = (nfft, precision, device)
    plan = _plans.get(key)
    if plan is None or plan.max_rows < rows:
        if plan is not None:
            plan.close()
        plan = _hip.Plan(nfft, precision, max_rows=max(1024, rows), device=device)
        _plans[key] = plan
    return plan


def _device_id(mother):
    try:
        return mother.device_id()
    except AttributeError:
        raise NotImplementedError(
            "pycwt_amd.cwt needs a built-in mother (Morlet, Paul, DOG, MexicanHat from pycwt_amd); "
            f"got {type(mother).__name__} without device_id()") from None


def _next_pow2(n0: int) -> int:
    return int(2 ** np.ceil(np.log2(n0)))       # helpers.py:27-30


def _nan_rows(mother, sj, N, dt):
    """Rows the reference deletes at wavelet.py:111-115, decided without building W.

    NaN can only come from psi_ft (Paul: c*f**m*exp(-f) overflows to inf for very negative f, then
    inf*0); the most negative angular frequency is the Nyquist bin ftfreqs[N//2].  One NaN in the
    filtered spectrum makes the whole inverse-FFT row NaN, so the row test is a single evaluation.
    """
    w_min = 2 * np.pi * (-(N // 2) * (1.0 / (N * dt)))
    with np.errstate(all="ignore"):
        bad = np.isnan(np.asarray(mother.psi_ft(sj * w_min)))
    return bad


def cwt(signal, dt, dj=1 / 12, s0=-1, J=-1, wavelet="morlet", freqs=None, *, precision=None,
        device=0):
    """Continuous wavelet transform; drop-in for ``pycwt.cwt`` (wavelet.py:13-124).

    Returns ``(W[:, :n0], sj, freqs, coi, fft, fftfreqs)`` exactly as the reference does.  ``W`` is
    complex128; ``precision=32`` (or ``PYCWT_AMD_PRECISION=32``) computes in complex64 on the GPU
    (1e-3 relative parity) and widens on return.  Keyword-only extras do not disturb positional use.
    """
    mother = _check_parameter_wavelet(wavelet)
    precision = _default_precision() if precision is None else int(precision)
    n0 = len(signal)
    if freqs is None:                                   # wavelet.py:75-85
        if s0 == -1:
            s0 = 2 * dt / mother.flambda()
        if J == -1:
            J = int(np.round(np.log2(n0 * dt / s0) / dj))
        sj = s0 * 2 ** (np.arange(0, J + 1) * dj)
        freqs = 1 / (mother.flambda() * sj)
    else:                                               # wavelet.py:86-88
        sj = 1 / (mother.flambda() * freqs)
    sj = np.asarray(sj, dtype=np.float64)

    N = _next_pow2(n0)
    bad = _nan_rows(mother, sj, N, dt)
    if bad.any() and not bad.all():                     # wavelet.py:111-115
        keep = ~bad
        sj = sj[keep]
        freqs = np.asarray(freqs)[keep]

    kind, param = _device_id(mother)
    plan = _plan(N, precision, device, sj.size)
    real = np.float64 if precision == 64 else np.float32
    W, xhat = plan.execute_host(np.asarray(signal, dtype=real), kind, param, dt, sj)
    if W.dtype != np.complex128:
        W = W.astype(np.complex128)
        xhat = xhat.astype(np.complex128)

    coi = n0 / 2 - np.abs(np.arange(0, n0) - (n0 - 1) / 2)      # wavelet.py:120-121
    coi = mother.flambda() * mother.coi() * dt * coi
    ftfreqs = 2 * np.pi * np.fft.fftfreq(N, dt)                 # wavelet.py:94
    return (W, sj, freqs, coi, xhat[1:N // 2] / N ** 0.5, ftfreqs[1:N // 2] / (2 * np.pi))


def icwt(W, sj, dt, dj=1 / 12, wavelet="morlet", *, precision=None, device=0):
    """Inverse transform, TC98 eq. 11; drop-in for ``pycwt.icwt`` (wavelet.py:127-171).

    The column reduction sum_j Re(W[j, n]) / sqrt(s_j) runs on the GPU; the scalar factor
    dj*sqrt(dt)/(cdelta*psi(0)) -- complex for Morlet and Paul, as in the reference -- is applied on
    the host, so the result dtype matches the reference's.
    """
    mother = _check_parameter_wavelet(wavelet)
    precision = _default_precision() if precision is None else int(precision)
    W = np.asarray(W)
    sj = np.asarray(sj, dtype=np.float64)
    a, b = W.shape
    c = sj.size
    if a == c:
        row_scale, col_scale = sj, None
    elif b == c:                                        # wavelet.py:163-164 still sums axis 0
        row_scale, col_scale = np.ones(a), sj
    else:
        raise Warning("Input array dimensions do not match.")   # wavelet.py:166

    plan = _plan(_next_pow2(max(b, 2)), precision, device, a)
    esize = np.dtype(plan.real).itemsize
    Wd = _hip.DeviceBuffer(a * b * 2 * esize, device)
    out = _hip.DeviceBuffer(b * esize, device)
    try:
        Wd.upload(plan, np.ascontiguousarray(W, dtype=plan.cplx))
        plan.icwt_reduce(Wd.ptr, b, b, row_scale, 1.0, out.ptr)
        total = out.download(plan, (b,), plan.real).astype(np.float64)
    finally:
        Wd.free()
        out.free()
    if col_scale is not None:
        total = total / np.sqrt(col_scale)
    return dj * np.sqrt(dt) / (mother.cdelta * mother.psi(0)) * total
