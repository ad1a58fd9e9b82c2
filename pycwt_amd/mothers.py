"""Mother wavelets for the MI355X CWT engine.

Same duck-typed protocol as the reference's ``pycwt/mothers.py`` (``psi_ft``, ``psi``, ``flambda``,
``coi``, attributes ``cdelta``, ``gamma``, ``deltaj0``, ``dofmin``, ``name``; ``f0`` for Morlet,
``m`` for Paul/DOG) so user code and ``significance``-style callers keep working.  On the hot path
only ``device_id()`` is used: the Fourier-domain profile is evaluated inside the HIP kernels
(pycwt_amd/csrc/cwt_kernels.hpp, ``profile``), the NumPy ``psi_ft`` here exists for callers that
want the filter itself and for the shim's NaN-row rule.

Constants are Torrence & Compo (1998) table 2, as in mothers.py:46-59 / 142-155 / 205-222.
"""
from __future__ import annotations

import math

import numpy as np
from scipy.special import gamma as _gamma_fn

from . import _hip

# (cdelta, gamma, deltaj0) per (family, parameter); anything else is "unknown" = -1 as in the reference
_TC98 = {
    ("morlet", 6): (0.776, 2.32, 0.60),
    ("paul", 4): (1.132, 1.17, 1.50),
    ("dog", 2): (3.541, 1.43, 1.40),
    ("dog", 6): (1.966, 1.37, 0.97),
}


class _Mother:
    family = ""
    dofmin = 2

    def _set_constants(self, param):
        self.cdelta, self.gamma, self.deltaj0 = _TC98.get((self.family, param), (-1, -1, -1))

    def device_id(self):
        """(mother id, parameter) understood by cwt_transform_rows (include/cwt_hip.h)."""
        raise NotImplementedError


class Morlet(_Mother):
    """Morlet wavelet, angular wavenumber ``f0`` (reference: mothers.py:13-59)."""

    family = "morlet"

    def __init__(self, f0=6):
        self.f0 = f0
        self.name = "Morlet"
        self._set_constants(f0)

    def _set_f0(self, f0):
        self.f0 = f0
        self._set_constants(f0)

    def psi_ft(self, f):          # mothers.py:26-28
        return np.exp(-0.5 * (f - self.f0) ** 2) * math.pi ** -0.25

    def psi(self, t):             # mothers.py:30-32 (complex even at t = 0)
        return math.pi ** -0.25 * np.exp(1j * self.f0 * t - t ** 2 / 2)

    def flambda(self):            # mothers.py:34-36
        return 4 * math.pi / (self.f0 + math.sqrt(2 + self.f0 ** 2))

    def coi(self):                # mothers.py:38-40
        return 1 / math.sqrt(2)

    def device_id(self):
        return _hip.MORLET, float(self.f0)

    def smooth(self, W, dt, dj, scales, *, precision=None, device=0):
        """Coherence smoothing operator (mothers.py:61-104): Gaussian of width s/dt along time (FFT
        filter, on the GPU) and a 2*deltaj0/dj boxcar along scales.  Real input gives real output."""
        from . import wavelet as _w
        W = np.asarray(W)
        rows, n = W.shape
        precision = _w._default_precision() if precision is None else int(precision)
        plan = _w._plan(_w._next_pow2(n), precision, device, rows)
        es = np.dtype(plan.real).itemsize
        sc = _w._Scratch(device)
        try:
            T, tmp, out = (sc.new(rows * n * 2 * es) for _ in range(3))
            spec = sc.new(rows * plan.nfft * 2 * es)
            T.upload(plan, np.ascontiguousarray(W, dtype=plan.cplx))
            _w._smooth_on_device(plan, self, T, rows, n, dt, dj, np.asarray(scales, dtype=float), spec, tmp, out)
            res = out.download(plan, (rows, n), plan.cplx).astype(np.complex128)
        finally:
            sc.free()
        return res.real if np.isreal(W).all() else res


class Paul(_Mother):
    """Paul wavelet of integer order ``m`` (reference: mothers.py:107-155)."""

    family = "paul"

    def __init__(self, m=4):
        self.m = m
        self.name = "Paul"
        self._set_constants(m)

    def _set_m(self, m):
        self.m = m
        self._set_constants(m)

    def psi_ft(self, f):          # mothers.py:118-122, same operation order (NaN where exp overflows)
        m = self.m
        c = 2 ** m / np.sqrt(m * np.prod(range(2, 2 * m)))
        return c * f ** m * np.exp(-f) * (f > 0)

    def psi(self, t):             # mothers.py:124-128; keeps the reference's (m-2)! factor (sic)
        m = self.m
        return (2 ** m * 1j ** m * np.prod(range(2, m - 1)) /
                np.sqrt(np.pi * np.prod(range(2, 2 * m + 1))) * (1 - 1j * t) ** (-(m + 1)))

    def flambda(self):            # mothers.py:130-132
        return 4 * math.pi / (2 * self.m + 1)

    def coi(self):                # mothers.py:134-136
        return math.sqrt(2)

    def device_id(self):
        return _hip.PAUL, float(self.m)


class DOG(_Mother):
    """m-th derivative of a Gaussian (reference: mothers.py:158-222)."""

    family = "dog"
    dofmin = 1

    def __init__(self, m=2):
        self.m = m
        self.name = "DOG"
        self._set_constants(m)

    def _set_m(self, m):
        self.m = m
        self._set_constants(m)

    def psi_ft(self, f):          # mothers.py:170-173: -(1j**m) / sqrt(Gamma(m + 1/2)) f^m e^{-f^2/2}
        return -(1j ** self.m) / np.sqrt(_gamma_fn(self.m + 0.5)) * f ** self.m * np.exp(-0.5 * f ** 2)

    def psi(self, t):             # mothers.py:175-191 via probabilists' Hermite polynomial He_m
        he = np.polynomial.hermite_e.hermeval(t, [0] * self.m + [1])
        return (-1) ** (self.m + 1) * he * np.exp(-np.square(t) / 2) / np.sqrt(_gamma_fn(self.m + 0.5))

    def flambda(self):            # mothers.py:193-195
        return 2 * math.pi / math.sqrt(self.m + 0.5)

    def coi(self):                # mothers.py:197-199
        return 1 / math.sqrt(2)

    def device_id(self):
        return _hip.DOG, float(self.m)


class MexicanHat(DOG):
    """DOG with m = 2 (reference: mothers.py:225-233)."""

    def __init__(self):
        super().__init__(2)
        self.name = "Mexican Hat"
