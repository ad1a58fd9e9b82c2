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
import threading

import numpy as np

from scipy.stats import chi2

from . import _hip
from .helpers import ar1, ar1_spectrum, find, get_cache_dir, rednoise
from .mothers import DOG, MexicanHat, Morlet, Paul

_MOTHERS = {"morlet": Morlet, "paul": Paul, "dog": DOG, "mexicanhat": MexicanHat}
_plans: dict = {}
_plans_lock = threading.Lock()
# cwt_plan_set_option pairs applied to every plan this module creates from now on (tuning / tests), e.g.
# {"ols": 0} = N-point transform of every row, {"ols_min_logn": 15} = overlap-save rows from N = 2^15 on
PLAN_OPTIONS: dict = {}


# Accuracy of the engine's fast forms (cwt_plan_set_tolerance).  "auto" (the default): the target below holds relative
# to every row's own peak for signals of any spectral shape -- each transform measures the dynamic range of its spectrum
# (max|xhat| / rms|xhat|, one small reduction on the GPU) and tightens the engine's filter-relative tolerance by it; a
# float = that tolerance for every call (1e-16: every truncation below fp64 rounding); None = the engine's own default
# (round-off).  The environment variable CWT_TOLERANCE (read by the engine) overrides "auto".
AUTO_TARGET = {64: 1e-9, 32: 3e-5}
_tolerance = "auto"


def _auto(plan):
    """The automatic mode's target for this plan, or 0.0 when a fixed tolerance is in force."""
    if _tolerance == "auto" and not os.environ.get("CWT_TOLERANCE") and "tolerance" not in PLAN_OPTIONS:
        return AUTO_TARGET[plan.precision]
    return 0.0


def _apply_tolerance(plan):
    if _tolerance == "auto":
        plan.set_auto_tolerance(_auto(plan))
    else:
        plan.set_auto_tolerance(0.0)
        plan.set_tolerance(float(_tolerance or 0.0))


def set_tolerance(rel_tol="auto"):
    """Accuracy target of every transform of this module from now on.

    "auto" (default): aims at max|W - W_reference| / max|W_reference| <= 1e-9 per row (3e-5 in complex64) whatever the
    spectrum's shape: the engine's truncations are relative to the filter, so each call divides the target by the measured
    dynamic range of its spectrum -- largest bin over the quietest 3/4-octave stretch, `cwt_plan_auto_tolerance`.  Measured
    on white / red noise, lines, high-passed and notched signals (tests/test_tolerance_emulated.py), not a proof: a signal
    with almost no energy inside one row's band but a lot just outside can exceed the target on that row.  A float: the engine's filter-relative tolerance
    itself, for every call (`cwt_plan_set_tolerance`; 1e-16 = every truncation below fp64 rounding).  None / 0: the
    engine's default, round-off."""
    global _tolerance
    with _plans_lock:
        _tolerance = rel_tol if rel_tol else None
        PLAN_OPTIONS.pop("tolerance", None)
        for plan in _plans.values():
            if plan.h:
                _apply_tolerance(plan)


def _check_parameter_wavelet(wavelet):
    """wavelet.py:650-663: lower-case name -> default instance (KeyError if unknown); objects pass."""
    if isinstance(wavelet, str):
        return _MOTHERS[wavelet]()
    return wavelet


def _default_precision() -> int:
    return int(os.environ.get("PYCWT_AMD_PRECISION", "64"))


def _plan(nfft: int, precision: int, device: int, rows: int) -> _hip.Plan:
    """The cached plan of (padded length, precision, device), grown when a call needs more rows.  A plan that is
    too small is only dropped from the cache, never closed here: a live `DeviceTransform` (or another thread in
    the middle of a call) may still hold it, and the garbage collector closes it with its last reference.
    Callers run their whole upload -> transform -> download sequence under `plan.lock`, because the C plan (row
    table, workspaces, stream) is shared by every thread that asks for the same key; the reference is re-entrant
    and so is this module, one transform per (length, precision, device) at a time."""
    key = (nfft, precision, device)
    with _plans_lock:
        plan = _plans.get(key)
        if plan is None or plan.max_rows < rows:
            plan = _hip.Plan(nfft, precision, max_rows=max(1024, rows), device=device, options=dict(PLAN_OPTIONS))
            if "tolerance" not in PLAN_OPTIONS:
                _apply_tolerance(plan)
            _plans[key] = plan
        return plan


def _reduction_plan(precision: int, device: int, rows: int) -> _hip.Plan:
    """A plan for the column/row reductions (icwt, power averages): those kernels do not depend on the transform
    length, so any cached plan of the device and precision with enough rows serves; otherwise the smallest one is
    made (no FFT tables worth mentioning) instead of a full-length FFT plan."""
    with _plans_lock:
        for (nfft, prec, dev), plan in _plans.items():
            if prec == precision and dev == device and plan.max_rows >= rows and plan.h:
                return plan
    return _plan(16, precision, device, rows)


def _transform(plan, x_host, xd_ptr, n0, kind, param, dt, sj, xh_ptr, W_ptr, auto=True):
    """Signal (already uploaded at xd_ptr) -> spectrum and rows of W on the device.

    `cwt_transform` computes time-compact rows block by block from the signal itself (overlap-save), so a NaN or inf
    sample would only poison the blocks that contain it.  The reference transforms the whole padded signal
    (wavelet.py:91): one non-finite sample makes EVERY bin of the spectrum, hence every element of W, NaN.  To stay a
    drop-in, such signals go through the spectrum-only entry points (`cwt_forward_fft` + `cwt_transform_rows`, which
    never use the overlap-save form); the O(N) host scan is free next to the PCIe transfer of W."""
    if np.isfinite(x_host).all():
        # (single-workgroup transforms: round-off costs nothing; auto=False: the caller's series are alike -- the Monte-Carlo
        # surrogates -- and the tolerance the first of them set stays)
        target = _auto(plan) if plan.nfft > 4096 and auto else 0.0
        if target:      # automatic accuracy: the tolerance of this call from the dynamic range of its spectrum
            plan.forward_fft(xd_ptr, n0, xh_ptr)
            plan.set_tolerance(plan.auto_tolerance(xh_ptr, target))
        plan.transform(xd_ptr, n0, kind, param, dt, sj, xh_ptr, W_ptr, n0, n0)
    else:
        plan.forward_fft(xd_ptr, n0, xh_ptr)
        plan.transform_rows(xh_ptr, kind, param, dt, sj, W_ptr, n0, n0)


def _cwt_builtin(x, dt, sj, kind, param, N, precision, device, finite):
    """W (rows x n0) and the spectrum (N) for a built-in mother through the device-resident entry points."""
    plan = _plan(N, precision, device, sj.size)
    if finite:
        return plan.execute_host(x, kind, param, dt, sj)
    es = np.dtype(plan.real).itemsize
    n0 = x.size
    sc = _Scratch(device)
    try:
        xd, xh, Wd = sc.new(n0 * es), sc.new(N * 2 * es), sc.new(sj.size * n0 * 2 * es)
        with plan.lock:
            xd.upload(plan, np.ascontiguousarray(x, dtype=plan.real))
            _transform(plan, x, xd.ptr, n0, kind, param, dt, sj, xh.ptr, Wd.ptr)
            return Wd.download(plan, (sj.size, n0), plan.cplx), xh.download(plan, (N,), plan.cplx)
    finally:
        sc.free()


def _device_id(mother, strict=True):
    try:
        return mother.device_id()
    except AttributeError:
        if not strict:
            return None
        raise NotImplementedError(
            "pycwt_amd.cwt needs a built-in mother (Morlet, Paul, DOG, MexicanHat from pycwt_amd); "
            f"got {type(mother).__name__} without device_id()") from None


def _next_pow2(n0: int) -> int:
    return int(2 ** np.ceil(np.log2(n0)))       # helpers.py:27-30


def _scale_grid(mother, n0, dt, dj, s0, J, freqs):
    """Scales and Fourier-equivalent frequencies exactly as wavelet.py:75-88: default s0 = 2*dt/flambda,
    default J = round(log2(n0*dt/s0)/dj), sj = s0*2^(j*dj) for j = 0..J (J + 1 rows); or, when `freqs` is
    given, sj = 1/(flambda*freqs)."""
    if freqs is None:
        if s0 == -1:
            s0 = 2 * dt / mother.flambda()
        if J == -1:
            J = int(np.round(np.log2(n0 * dt / s0) / dj))
        sj = s0 * 2 ** (np.arange(0, J + 1) * dj)
        freqs = 1 / (mother.flambda() * sj)
    else:
        sj = 1 / (mother.flambda() * freqs)
    return np.asarray(sj, dtype=np.float64), freqs


def _coi(mother, n0, dt):
    """Cone of influence in Fourier periods (wavelet.py:120-121)."""
    return mother.flambda() * mother.coi() * dt * (n0 / 2 - np.abs(np.arange(0, n0) - (n0 - 1) / 2))


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


_geometry_cache = {}
_geometry_lock = threading.Lock()


def _geometry(mother, n0, dt, dj, s0, J, freqs, pad):
    """Everything of a call that does not depend on the samples: (N, sj, freqs, coi, fftfreqs[1:N//2], NaN-row mask or
    None).  The reference recomputes these per call (wavelet.py:75-94, :120-121); at its canonical 504-point call that is
    half of what a call costs here, so built-in mothers on the default grid keep the last few results.  The arrays of a
    cached entry are never handed out: `cwt` returns copies."""
    key = None
    if freqs is None and hasattr(mother, "device_id"):
        try:
            key = (type(mother), mother.device_id(), n0, dt, dj, s0, J, pad)
            with _geometry_lock:
                hit = _geometry_cache.get(key)
            if hit is not None:
                return hit[0]
        except TypeError:                                   # an unhashable argument (array-valued dt ...): no cache
            key = None
    sj, freqs = _scale_grid(mother, n0, dt, dj, s0, J, freqs)
    N = _next_pow2(n0) if pad else n0
    bad = _nan_rows(mother, sj, N, dt) if hasattr(mother, "device_id") else None
    if bad is not None and not bad.any():
        bad = None
    ftfreqs = 2 * np.pi * np.fft.fftfreq(N, dt)              # wavelet.py:94
    geo = (N, sj, freqs, _coi(mother, n0, dt), ftfreqs[1:N // 2] / (2 * np.pi), bad)
    if key is not None:
        nbytes = sum(a.nbytes for a in geo[1:] if isinstance(a, np.ndarray))
        with _geometry_lock:
            held = sum(v[1] for v in _geometry_cache.values())
            if len(_geometry_cache) >= 64 or held + nbytes > (128 << 20):      # (a 2^20-point call keeps 12 MB here)
                _geometry_cache.clear()
            _geometry_cache[key] = (geo, nbytes)
    return geo


def _cwt_with_host_filter_bank(x, dt, sj, mother, N, precision, device):
    """Any duck-typed mother (only `psi_ft`, `flambda`, `coi` needed, as in the reference): the filter
    bank of wavelet.py:102-104 is evaluated with NumPy exactly as the reference does and handed to the
    engine as an explicit table; FFTs, multiply and the band-limited / two-pass machinery stay on the
    GPU.  Returns (W, xhat, keep-mask or None)."""
    ftfreqs = 2 * np.pi * np.fft.fftfreq(N, dt)
    with np.errstate(all="ignore"):
        bank = (sj[:, None] * ftfreqs[1] * N) ** .5 * np.conjugate(mother.psi_ft(sj[:, None] * ftfreqs))
    bank = np.asarray(bank, dtype=np.complex128) * np.ones((1, N))
    bad = np.isnan(bank).any(axis=1)                        # a NaN bin makes the whole row NaN (:111-115)
    keep = None
    if bad.any() and not bad.all():
        keep = ~bad
        bank, sj = bank[keep], sj[keep]
    bank = np.nan_to_num(bank)
    rows = sj.size
    mag = np.fft.fftshift(np.abs(bank), axes=1)             # signed-bin order: -N/2 .. N/2-1
    live = mag > (1e-18 if precision == 64 else 1e-9) * np.maximum(mag.max(axis=1, keepdims=True), 1e-300)
    first = np.where(live.any(axis=1), live.argmax(axis=1), 0)
    last = np.where(live.any(axis=1), N - 1 - live[:, ::-1].argmax(axis=1), -1)
    k_lo, nband = first - N // 2, np.maximum(last - first + 1, 0)
    plan = _plan(N, precision, device, rows)
    es = np.dtype(plan.real).itemsize
    sc = _Scratch(device)
    try:
        xd, xh = sc.new(x.size * es), sc.new(N * 2 * es)
        tab, Wd = sc.new(rows * N * 2 * es), sc.new(rows * x.size * 2 * es)
        with plan.lock:
            xd.upload(plan, x)
            tab.upload(plan, np.ascontiguousarray(bank, dtype=plan.cplx))
            plan.forward_fft(xd.ptr, x.size, xh.ptr)
            plan.transform_rows_table(xh.ptr, tab.ptr, k_lo, nband, Wd.ptr, x.size, x.size)
            return Wd.download(plan, (rows, x.size), plan.cplx), xh.download(plan, (N,), plan.cplx), keep
    finally:
        sc.free()


def _cwt_unpadded(x, dt, sj, kind, param, precision, device):
    """W (rows x n0) and the spectrum (n0) at transform length n0 = len(x), not a power of two: Bluestein's chirp-z
    identity on the power-of-two engine (cwt_forward_fft_n / cwt_transform_rows_n)."""
    n0 = x.size
    M = _next_pow2(2 * n0 - 1)
    plan = _plan(M, precision, device, min(sj.size, 1024))
    es = np.dtype(plan.real).itemsize
    sc = _Scratch(device)
    try:
        xd, xh, Wd = sc.new(n0 * es), sc.new(n0 * 2 * es), sc.new(sj.size * n0 * 2 * es)
        with plan.lock:
            xd.upload(plan, x)
            plan.forward_fft_n(xd.ptr, n0, xh.ptr)
            plan.transform_rows_n(xh.ptr, n0, kind, param, dt, sj, Wd.ptr, n0)
            return Wd.download(plan, (sj.size, n0), plan.cplx), xh.download(plan, (n0,), plan.cplx)
    finally:
        sc.free()


def cwt(signal, dt, dj=1 / 12, s0=-1, J=-1, wavelet="morlet", freqs=None, *, precision=None,
        device=0, pad=True):
    """Continuous wavelet transform; drop-in for ``pycwt.cwt`` (wavelet.py:13-124).

    Returns ``(W[:, :n0], sj, freqs, coi, fft, fftfreqs)`` exactly as the reference does.  ``W`` is
    complex128; ``precision=32`` (or ``PYCWT_AMD_PRECISION=32``) computes in complex64 on the GPU
    (1e-3 relative parity) and widens on return.  Keyword-only extras do not disturb positional use.

    ``pad=True`` is the reference with its scipy.fftpack backend: the transform length is the next power of
    two (helpers.py:27-30).  ``pad=False`` is the reference with pyfftw installed: the transform length is
    ``len(signal)`` itself (helpers.py:15-19), no zero padding, circular edges, any length.
    """
    mother = _check_parameter_wavelet(wavelet)
    precision = _default_precision() if precision is None else int(precision)
    if np.iscomplexobj(signal):
        # wavelet.py:91 transforms a complex signal as it is; the engine's forward transform is real-input, and the whole
        # path is linear: W(x) = W(Re x) + i W(Im x), likewise the spectrum of the 5th return value
        z = np.asarray(signal)
        zr, zi = z.real, z.imag
        nonfinite = ~(np.isfinite(zr) & np.isfinite(zi))
        if nonfinite.any():
            # one NaN / inf sample makes every bin of the reference's spectrum NaN, whichever part it sits in: both parts
            # must take the "whole matrix is NaN, every row kept" branch (wavelet.py:111-115), or they would drop
            # different sets of rows
            zr, zi = zr.copy(), zi.copy()
            zr[nonfinite] = np.nan
            zi[nonfinite] = np.nan
        a = cwt(zr, dt, dj, s0, J, mother, freqs, precision=precision, device=device, pad=pad)
        b = cwt(zi, dt, dj, s0, J, mother, freqs, precision=precision, device=device, pad=pad)
        fft5 = a[4] + 1j * b[4]
        if z.dtype == np.complex64:
            fft5 = fft5.astype(np.complex64)            # scipy.fftpack keeps single precision (wavelet.py:91, :123-124)
        return (a[0] + 1j * b[0], a[1], a[2], a[3], fft5, a[5])
    in_dtype = getattr(signal, "dtype", None)
    n0 = len(signal)
    user_freqs = freqs is not None
    N, sj, freqs, coi, fftfreqs, bad = _geometry(mother, n0, dt, dj, s0, J, freqs, pad)
    real = np.float64 if precision == 64 else np.float32
    if N & (N - 1):                                         # pad=False with a length that is not a power of two
        if bad is not None and not bad.all():
            sj, freqs = sj[~bad], np.asarray(freqs)[~bad]
        kind, param = _device_id(mother)
        W, xhat = _cwt_unpadded(np.ascontiguousarray(signal, dtype=real), dt, sj, kind, param, precision, device)
    elif hasattr(mother, "device_id"):
        x = np.asarray(signal, dtype=real)
        finite = bool(np.isfinite(x).all())
        # wavelet.py:111-115 drops the rows that are NaN throughout -- unless EVERY row is, which is what a NaN / inf
        # sample does to the whole matrix (:91): then the reference keeps all rows, and so do we
        if bad is not None and not bad.all() and finite:
            keep = ~bad
            sj = sj[keep]
            freqs = np.asarray(freqs)[keep]
        kind, param = mother.device_id()
        W, xhat = _cwt_builtin(x, dt, sj, kind, param, N, precision, device, finite)
        if bad is not None and bad.all():
            # every row carries a NaN in its filter (Paul, all scales beyond the overflow of exp(-f)): the reference keeps
            # all rows then (wavelet.py:112) and every one of them is NaN throughout
            W = np.full(W.shape, complex(np.nan, np.nan), dtype=W.dtype)
    else:
        W, xhat, keep = _cwt_with_host_filter_bank(np.asarray(signal, dtype=real), dt, sj, mother, N, precision,
                                                   device)
        if keep is not None:
            sj = sj[keep]
            freqs = np.asarray(freqs)[keep]
    if W.dtype != np.complex128:
        W = W.astype(np.complex128)
        xhat = xhat.astype(np.complex128)

    fft5 = xhat[1:N // 2] / N ** 0.5
    if in_dtype == np.float32:
        fft5 = fft5.astype(np.complex64)        # the reference's FFT of a float32 signal is complex64 (wavelet.py:91, :123-124)
    if not user_freqs:
        freqs = np.array(freqs)                 # (cached grids stay private)
    return (W, np.array(sj), freqs, np.array(coi), fft5, np.array(fftfreqs))


class DeviceTransform:
    """A wavelet transform that stays on the GPU: W (rows x n0 complex) is device
    resident; the reductions every caller of `cwt` does next -- power, global spectrum, scale
    averages, reconstruction -- run there and only vectors cross PCIe.  `W()` downloads the matrix."""

    def __init__(self, plan, buf, sj, freqs, coi, fft, fftfreqs, mother, dt, n0, spectrum=None):
        self._plan, self._buf = plan, buf
        self.sj, self.freqs, self.coi, self.fftfreqs = sj, freqs, coi, fftfreqs
        self._fft, self._spectrum = fft, spectrum              # the 5th return value of cwt(): downloaded when first asked for
        self.mother, self.dt, self.n0 = mother, dt, n0
        self.shape = (sj.size, n0)

    @property
    def fft(self):
        if self._fft is None and self._spectrum is not None:
            N = self._plan.nfft
            xhat = self._spectrum.download(self._plan, (N,), self._plan.cplx).astype(np.complex128)
            self._fft = xhat[1:N // 2] / N ** 0.5
            self._spectrum.free()
            self._spectrum = None
        return self._fft

    def close(self):
        if self._buf is not None:
            self._buf.free()
            self._buf = None
        if self._spectrum is not None:
            self._spectrum.free()
            self._spectrum = None

    __del__ = close

    @property
    def device_ptr(self):
        return self._buf.ptr

    def _vector(self, n, fill):
        es = np.dtype(self._plan.real).itemsize
        out = _hip.DeviceBuffer(n * es, self._plan.device)
        try:
            with self._plan.lock:
                fill(out.ptr)
                return out.download(self._plan, (n,), self._plan.real).astype(np.float64)
        finally:
            out.free()

    def W(self):
        return self._buf.download(self._plan, self.shape, self._plan.cplx).astype(np.complex128, copy=False)

    def global_power(self):
        """mean over time of |W|^2 per scale (`power.mean(axis=1)`, sample/simple_sample.py:79)."""
        rows, n0 = self.shape
        return self._vector(rows, lambda p: self._plan.time_mean_power(self._buf.ptr, n0, n0, rows, p))

    def scale_average(self, s1, s2, dj):
        """Scale-averaged power over s1 <= s < s2: dj*dt/cdelta * sum_j |W_j|^2/s_j (TC98 eq. 24,
        sample/simple_sample.py:85-91)."""
        rows, n0 = self.shape
        w = np.where((self.sj >= s1) & (self.sj < s2), 1.0 / self.sj, 0.0)
        coeff = dj * self.dt / self.mother.cdelta
        return self._vector(n0, lambda p: self._plan.reduce_scales(self._buf.ptr, n0, n0, w, True, coeff, p))

    def icwt(self, dj):
        """TC98 eq. 11 without leaving the device (wavelet.py:169-170)."""
        rows, n0 = self.shape
        total = self._vector(n0, lambda p: self._plan.icwt_reduce(self._buf.ptr, n0, n0, self.sj, 1.0, p))
        return dj * np.sqrt(self.dt) / (self.mother.cdelta * self.mother.psi(0)) * total


def cwt_device(signal, dt, dj=1 / 12, s0=-1, J=-1, wavelet="morlet", freqs=None, *, precision=None, device=0):
    """`cwt` whose result stays on the GPU: returns a `DeviceTransform` (attributes sj, freqs, coi, fft,
    fftfreqs as in `cwt` -- read-only views of the cached grids; `fft`, the spectrum, is downloaded when first asked for
    and must be asked for before `close()`; methods W(), global_power(), scale_average(), icwt())."""
    mother = _check_parameter_wavelet(wavelet)
    precision = _default_precision() if precision is None else int(precision)
    n0 = len(signal)
    kind, param = _device_id(mother)
    user_freqs = freqs is not None
    N, sj, freqs, coi, fftfreqs, bad = _geometry(mother, n0, dt, dj, s0, J, freqs, True)
    if bad is not None and not bad.all() and np.isfinite(np.asarray(signal, dtype=np.float64)).all():   # see cwt()
        sj, freqs = sj[~bad], np.asarray(freqs)[~bad]
    plan = _plan(N, precision, device, sj.size)
    es = np.dtype(plan.real).itemsize
    xd, xh = _hip.DeviceBuffer(n0 * es, device), _hip.DeviceBuffer(N * 2 * es, device)
    Wd = _hip.DeviceBuffer(sj.size * n0 * 2 * es, device)
    try:
        with plan.lock:
            xs_host = np.ascontiguousarray(signal, dtype=plan.real)
            xd.upload(plan, xs_host)
            _transform(plan, xs_host, xd.ptr, n0, kind, param, dt, sj, xh.ptr, Wd.ptr)
            plan.sync()
    except Exception:
        Wd.free()
        xh.free()
        raise
    finally:
        xd.free()
    # (the cached grids are handed out as READ-ONLY views here, not copied as in cwt(): at 2^20 points the copies of coi and
    # fftfreqs -- 12 MB of fresh pages per call -- cost more than the transform; profiles/r05_wct.txt)
    def ro(a):
        v = np.asarray(a).view()
        v.flags.writeable = False
        return v
    return DeviceTransform(plan, Wd, ro(sj), ro(freqs), ro(coi), None, ro(fftfreqs), mother, dt, n0, spectrum=xh)


def cwt_batch(signals, dt, dj=1 / 12, s0=-1, J=-1, wavelet="morlet", freqs=None, *, precision=None,
              device=0, max_batch_bytes=8 << 30):
    """`cwt` of a batch of equally long signals (2-D array, one signal per row) with one scale grid --
    an extension beyond the reference's 1-D `signal` (BASELINE config 4).

    Returns `(W, sj, freqs, coi, fft, fftfreqs)` with `W` of shape (batch, rows, n0) and `fft` of shape
    (batch, N//2 - 1); everything else as `cwt`.  The batch is pushed through the GPU in slabs of at
    most `max_batch_bytes` of W; inside a slab every kernel launch covers all signals.
    """
    mother = _check_parameter_wavelet(wavelet)
    precision = _default_precision() if precision is None else int(precision)
    X = np.atleast_2d(np.asarray(signals))
    nb, n0 = X.shape
    sj, freqs = _scale_grid(mother, n0, dt, dj, s0, J, freqs)
    N = _next_pow2(n0)
    bad = _nan_rows(mother, sj, N, dt)
    if bad.any() and not bad.all():
        sj, freqs = sj[~bad], np.asarray(freqs)[~bad]
    kind, param = _device_id(mother)
    rows = sj.size
    es = 8 if precision == 64 else 4
    slab = int(max(1, min(nb, max_batch_bytes // (rows * n0 * 2 * es))))
    plan = _plan(N, precision, device, slab * rows)
    W = np.empty((nb, rows, n0), dtype=np.complex128)
    xhat = np.empty((nb, N), dtype=np.complex128)
    sc = _Scratch(device)
    try:
        xd, xh = sc.new(slab * n0 * es), sc.new(slab * N * 2 * es)
        Wd = sc.new(slab * rows * n0 * 2 * es)
        for b0 in range(0, nb, slab):
            cnt = min(slab, nb - b0)
            with plan.lock:
                xs = np.ascontiguousarray(X[b0:b0 + cnt], dtype=plan.real)
                xd.upload(plan, xs)
                if np.isfinite(xs).all():
                    # with the signals at hand the time-compact rows take the overlap-save form (cwt_transform_batch)
                    plan.transform_batch(xd.ptr, cnt, n0, n0, kind, param, dt, sj, xh.ptr, Wd.ptr, n0, n0)
                else:
                    # a non-finite sample makes every coefficient of ITS signal NaN in the reference (the FFT spreads
                    # it, wavelet.py:91): rows from the spectra alone do the same, block-wise rows would not
                    plan.fft_rows(xd.ptr, False, cnt, n0, n0, xh.ptr)
                    plan.transform_rows_batch(xh.ptr, cnt, N, kind, param, dt, sj, Wd.ptr, n0, n0)
                W[b0:b0 + cnt] = Wd.download(plan, (cnt, rows, n0), plan.cplx)
                xhat[b0:b0 + cnt] = xh.download(plan, (cnt, N), plan.cplx)
    finally:
        sc.free()
    coi = _coi(mother, n0, dt)
    ftfreqs = 2 * np.pi * np.fft.fftfreq(N, dt)
    return (W, sj, freqs, coi, xhat[:, 1:N // 2] / N ** 0.5, ftfreqs[1:N // 2] / (2 * np.pi))


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

    plan = _reduction_plan(precision, device, a)
    esize = np.dtype(plan.real).itemsize
    Wd = _hip.DeviceBuffer(a * b * 2 * esize, device)
    out = _hip.DeviceBuffer(b * esize, device)
    try:
        with plan.lock:
            Wd.upload(plan, np.ascontiguousarray(W, dtype=plan.cplx))
            plan.icwt_reduce(Wd.ptr, b, b, row_scale, 1.0, out.ptr)
            total = out.download(plan, (b,), plan.real).astype(np.float64)
    finally:
        Wd.free()
        out.free()
    if col_scale is not None:
        total = total / np.sqrt(col_scale)
    return dj * np.sqrt(dt) / (mother.cdelta * mother.psi(0)) * total


# =============================================================================================
# Callers of the hot path.  Same signatures as the reference.
# =============================================================================================
def significance(signal, dt, scales, sigma_test=0, alpha=None, significance_level=0.95, dof=-1,
                 wavelet="morlet"):
    """Chi-square significance levels against an AR(1) background (wavelet.py:174-313, TC98 sec. 4-5).

    sigma_test 0: point-wise test (eq. 18); 1: time-averaged (eq. 23, `dof` = number of averaged
    points per scale); 2: scale-averaged over [s1, s2] = `dof` (eq. 25-28).  Host arithmetic, O(J).
    Returns (signif, fft_theor).
    """
    mother = _check_parameter_wavelet(wavelet)
    scales = np.asarray(scales, dtype=float)
    try:
        n0 = len(signal)
    except TypeError:
        n0 = 1
    variance = signal if n0 == 1 else np.asarray(signal).std() ** 2
    if alpha is None:
        alpha, _, _ = ar1(signal)
    dj = np.log2(scales[1] / scales[0])
    freq = dt / (scales * mother.flambda())                    # normalised frequency of each scale
    # red-noise spectrum, TC98 eq. 16
    fft_theor = variance * (1 - alpha ** 2) / (1 + alpha ** 2 - 2 * alpha * np.cos(2 * np.pi * freq / n0))
    dofmin = mother.dofmin
    if sigma_test == 0:
        signif = fft_theor * chi2.ppf(significance_level, dofmin) / dofmin
    elif sigma_test == 1:
        navg = np.atleast_1d(np.asarray(dofmin if np.isscalar(dof) and dof == -1 else dof, dtype=float))
        navg = np.broadcast_to(navg, scales.shape).copy() if navg.size == 1 else navg.copy()
        navg[navg < 1] = 1
        edof = dofmin * np.sqrt(1 + (navg * dt / mother.gamma / scales) ** 2)      # eq. 23
        edof[edof < dofmin] = dofmin
        signif = fft_theor * chi2.ppf(significance_level, edof) / edof
    elif sigma_test == 2:
        if np.isscalar(dof) or len(dof) != 2:
            raise Exception("DOF must be set to [s1, s2], the range of scale-averages")
        if mother.cdelta == -1:
            raise ValueError("Cdelta and dj0 not defined for {} with f0={}".format(
                mother.name, getattr(mother, "f0", getattr(mother, "m", None))))
        s1, s2 = dof
        sel = find((scales >= s1) & (scales <= s2))
        if sel.size == 0:
            raise ValueError("No valid scales between {} and {}.".format(s1, s2))
        savg = 1 / np.sum(1. / scales[sel])                                          # eq. 25
        smid = np.exp((np.log(s1) + np.log(s2)) / 2.)
        edof = (dofmin * sel.size * savg / smid) * np.sqrt(1 + (sel.size * dj / mother.deltaj0) ** 2)  # eq. 28
        fft_theor = savg * np.sum(fft_theor[sel] / scales[sel])                      # eq. 27
        signif = (dj * dt / mother.cdelta / savg) * fft_theor * chi2.ppf(significance_level, edof) / edof
    else:
        raise ValueError("sigma_test must be either 0, 1, or 2.")
    return signif, fft_theor


def _normalised(y, normalize):
    y = np.asarray(y)
    return ((y - y.mean()) / y.std()) if normalize else y


def xwt(y1, y2, dt, dj=1 / 12, s0=-1, J=-1, significance_level=0.95, wavelet="morlet", normalize=True,
        *, precision=None, device=0):
    """Cross wavelet transform W1 conj(W2) with its AR(1) chi-square level (wavelet.py:316-419).
    Returns (W12, coi, freq, signif)."""
    mother = _check_parameter_wavelet(wavelet)
    y1, y2 = np.asarray(y1), np.asarray(y2)
    std1, std2 = (1., 1.) if normalize else (y1.std(), y2.std())
    kw = dict(dj=dj, s0=s0, J=J, wavelet=mother, precision=precision, device=device)
    if _device_id(mother, strict=False) is not None and len(y1) == len(y2):
        # both transforms and the product stay on the device; one matrix crosses PCIe instead of two
        T1 = cwt_device(_normalised(y1, normalize), dt, **kw)
        try:
            T2 = cwt_device(_normalised(y2, normalize), dt, **kw)
            try:
                rows, n0 = T1.shape
                with T1._plan.lock:
                    T1._plan.cross_spectrum(T1.device_ptr, T2.device_ptr, rows, n0, n0, T1.device_ptr)
                    W12, freq, coi = T1.W(), T1.freqs, T1.coi
            finally:
                T2.close()
        finally:
            T1.close()
    else:
        W1, sj, freq, coi, _, _ = cwt(_normalised(y1, normalize), dt, **kw)
        W2, sj, freq, coi, _, _ = cwt(_normalised(y2, normalize), dt, **kw)
        W12 = W1 * W2.conj()
    a1, a2 = ar1(y1)[0], ar1(y2)[0]
    pk = np.sqrt(ar1_spectrum(freq * dt, a1) * ar1_spectrum(freq * dt, a2))
    dof = mother.dofmin
    signif = std1 * std2 * pk * chi2.ppf(significance_level, dof) / dof
    return W12, coi, freq, signif


def xwt_device(y1, y2, dt, dj=1 / 12, s0=-1, J=-1, significance_level=0.95, wavelet="morlet", normalize=True,
               *, precision=None, device=0):
    """`xwt` with the cross spectrum left on the GPU: returns (T, signif) where T is a `DeviceTransform` whose matrix is
    W1 conj(W2) (`T.W()` downloads it; `T.coi`, `T.freqs` as in `xwt`; `T.close()` frees it).  Built-in mothers, equally long
    series.  (`xwt` at two 2^20-point series is 47 ms of which ~40 are the download of the 77 x 2^20 complex matrix.)"""
    mother = _check_parameter_wavelet(wavelet)
    y1, y2 = np.asarray(y1), np.asarray(y2)
    if _device_id(mother, strict=False) is None or len(y1) != len(y2):
        raise ValueError("xwt_device needs a built-in mother and two series of one length")
    std1, std2 = (1., 1.) if normalize else (y1.std(), y2.std())
    kw = dict(dj=dj, s0=s0, J=J, wavelet=mother, precision=precision, device=device)
    T1 = cwt_device(_normalised(y1, normalize), dt, **kw)
    try:
        T2 = cwt_device(_normalised(y2, normalize), dt, **kw)
        try:
            rows, n0 = T1.shape
            with T1._plan.lock:
                T1._plan.cross_spectrum(T1.device_ptr, T2.device_ptr, rows, n0, n0, T1.device_ptr)
                T1._plan.sync()
        finally:
            T2.close()
    except Exception:
        T1.close()
        raise
    a1, a2 = ar1(y1)[0], ar1(y2)[0]
    pk = np.sqrt(ar1_spectrum(T1.freqs * dt, a1) * ar1_spectrum(T1.freqs * dt, a2))
    dof = mother.dofmin
    return T1, std1 * std2 * pk * chi2.ppf(significance_level, dof) / dof


# Work matrices that a call has finished with are kept for the next call instead of going back to the driver: a Monte-Carlo
# call works in ~60 GB (77 scales x 6 M samples), and allocating / freeing that took 0.1 ... 4.7 s per call on the MI355X
# boxes, more than the draws of a short call.  Bounded by a FRACTION of the card (PYCWT_AMD_SCRATCH_POOL_FRACTION of the
# device's total memory, default 0.25: 72 GB on an MI355X, 16 GB on a 64-GB card; PYCWT_AMD_SCRATCH_POOL_GB caps it in absolute
# terms; 0 = keep nothing), emptied by `release_scratch()` and whenever an allocation fails -- in DeviceBuffer or inside the
# library (a plan call that returns CWT_ENOMEM runs the same hook and is retried once, _hip._locked).
_POOL_LOCK = threading.RLock()     # re-entrant: a cyclic GC that runs while the lock is held may finalise a DeviceCoherence -> _pool_give
_POOL: dict = {}                  # (library, device, nbytes) -> [DeviceBuffer]
_POOL_HELD = [0]
_POOL_TOTALS: dict = {}           # (library, device) -> total device memory in bytes


def _pool_limit(lib=None, device=0):
    """Bytes the pool may hold on this device."""
    try:
        frac = float(os.environ.get("PYCWT_AMD_SCRATCH_POOL_FRACTION", "0.25"))
        cap_gb = os.environ.get("PYCWT_AMD_SCRATCH_POOL_GB")
        cap = None if cap_gb is None else int(float(cap_gb) * 2 ** 30)
    except ValueError:
        return 0
    total = None
    if lib is not None:
        key = (id(lib), device)
        total = _POOL_TOTALS.get(key)
        if total is None:
            try:
                total = _POOL_TOTALS[key] = lib.device_memory(device)[1]
            except Exception:               # (a library without the entry point: fall back to the absolute cap alone)
                total = None
    limit = int(frac * total) if total else (cap if cap is not None else 0)
    if cap is not None:
        limit = min(limit, cap)
    return max(0, limit)


def release_scratch():
    """Free the device memory kept from earlier calls (see PYCWT_AMD_SCRATCH_POOL_FRACTION / _GB)."""
    with _POOL_LOCK:
        bufs = [b for v in _POOL.values() for b in v]
        _POOL.clear()
        _POOL_HELD[0] = 0
    for b in bufs:
        b.free()


_hip.on_allocation_failure.append(release_scratch)


def _pool_take(lib, device, nbytes):
    with _POOL_LOCK:
        v = _POOL.get((id(lib), device, int(nbytes)))
        if v:
            _POOL_HELD[0] -= int(nbytes)
            return v.pop()
    return None


def _pool_give(bufs):
    if not bufs:
        return
    live = [b for b in bufs if b.ptr]
    if not live:
        return
    # nothing queued on ANY stream (the caller's own kernels and the plans' side streams included) may still use the buffers
    # when their next owner gets them: an explicit device-wide wait (round 5 leaned on the one hipFree implies)
    try:
        live[0].lib.device_synchronize(live[0].device)
    except Exception:
        for b in live:                                     # cannot vouch for the buffers: give them back to the driver
            b.free()
        return
    drop = []
    with _POOL_LOCK:                                       # accounting and insertion in ONE critical section
        for b in live:
            if _POOL_HELD[0] + b.nbytes <= _pool_limit(b.lib, b.device):
                _POOL_HELD[0] += b.nbytes
                _POOL.setdefault((id(b.lib), b.device, b.nbytes), []).append(b)
            else:
                drop.append(b)
    for b in drop:
        b.free()


class _Scratch:
    """Device buffers of one wct evaluation, freed together."""

    def __init__(self, device):
        self.device, self.bufs = device, []

    def new(self, nbytes):
        lib = _hip.load()
        b = _pool_take(lib, self.device, nbytes)
        if b is None:
            b = _hip.DeviceBuffer(nbytes, self.device, lib)      # (an allocation that fails empties the pool and tries again)
        self.bufs.append(b)
        return b

    def named(self, key, nbytes):
        """A buffer that survives until free(): the Monte-Carlo loop re-uses its ~10 work matrices per draw
        instead of allocating and freeing tens of GB every time."""
        cache = self.__dict__.setdefault("cache", {})
        b = cache.get(key)
        if b is None or b.nbytes < nbytes:
            b = cache[key] = self.new(nbytes)
        return b

    def free(self):
        bufs, self.bufs = self.bufs, []
        _pool_give(bufs)


def _smooth_on_device(plan, mother, T, rows, n, dt, dj, sj, spec, tmp, out):
    """mothers.py:61-104 on device: per-row FFT -> Gaussian of width s/dt -> inverse FFT -> boxcar over
    scales.  T, tmp, out: rows x n complex; spec: rows x nfft complex."""
    from .helpers import rect
    plan.fft_rows(T.ptr, True, rows, n, n, spec.ptr)
    a = (np.asarray(sj) / dt) * (2 * np.pi / plan.nfft)             # exp(-0.5*(s/dt)^2*k^2), k = 2*pi*fftfreq
    plan.filter_rows(spec.ptr, plan.nfft, _hip.DOG, 0.0, a, 1.0, tmp.ptr, n, n)
    win = rect(int(np.round(mother.deltaj0 / dj * 2)), normalize=True)
    plan.boxcar_scales(tmp.ptr, rows, n, n, win, out.ptr)


def _coherence_on_device(x1, x2, dt, dj, sj, mother, precision, device, want_angle=True, consume=None,
                         pool=None, auto=True, n0=None, keep=False):
    """|S12|^2/(S1 S2) and arg(W1 conj W2) for two equally long series, all on the GPU; only the two
    real result matrices cross PCIe.  With `consume(plan, r2_buffer, rows, n0)` the coherence stays on the
    device and is handed to that callback instead (Monte-Carlo histogram)."""
    on_device = isinstance(x1, _hip.DeviceBuffer)           # (surrogates made on the GPU: `n0` says how long they are)
    n0 = len(x1) if n0 is None else n0
    N = _next_pow2(n0)
    rows = len(sj)
    kind, param = _device_id(mother)
    plan = _plan(N, precision, device, rows)
    es = np.dtype(plan.real).itemsize
    sc = _Scratch(device) if pool is None else pool
    names = iter(range(100))
    alloc = (lambda nbytes: sc.new(nbytes)) if pool is None else (lambda nbytes: sc.named(next(names), nbytes))
    try:
        with plan.lock:
            xd, xh = alloc(n0 * es), alloc(N * 2 * es)
            W1, W2 = alloc(rows * n0 * 2 * es), alloc(rows * n0 * 2 * es)
            target = _auto(plan) if plan.nfft > 4096 and auto else 0.0
            if on_device:
                if target:
                    tols = []
                    for x in (x1, x2):
                        plan.forward_fft(x.ptr, n0, xh.ptr)
                        tols.append(plan.auto_tolerance(xh.ptr, target))
                    plan.set_tolerance(min(tols))
                for x, W in ((x1, W1), (x2, W2)):                 # finite by construction: the signal path (overlap-save rows) applies
                    plan.transform(x.ptr, n0, kind, param, dt, sj, xh.ptr, W.ptr, n0, n0)
            elif target and np.isfinite(x1).all() and np.isfinite(x2).all():
                # automatic accuracy: ONE tolerance for the pair, the tighter of the two spectra's (a red series paired with
                # a white one, `surrogates='ar1'` with unlike coefficients); it also stays for the later draws of a
                # Monte-Carlo loop (auto=False there), whose series come from the same two processes
                tols = []
                for x in (x1, x2):
                    xd.upload(plan, np.ascontiguousarray(x, dtype=plan.real))
                    plan.forward_fft(xd.ptr, n0, xh.ptr)
                    tols.append(plan.auto_tolerance(xh.ptr, target))
                plan.set_tolerance(min(tols))
            for x, W in (() if on_device else ((x1, W1), (x2, W2))):
                xh_ = np.ascontiguousarray(x, dtype=plan.real)
                xd.upload(plan, xh_)
                _transform(plan, xh_, xd.ptr, n0, kind, param, dt, sj, xh.ptr, W.ptr, auto=False)
            P, Cx, ang = alloc(rows * n0 * 2 * es), alloc(rows * n0 * 2 * es), alloc(rows * n0 * es)
            plan.wct_products(W1.ptr, W2.ptr, sj, n0, n0, P.ptr, Cx.ptr, ang.ptr)
            spec = alloc(rows * N * 2 * es)
            tmp, S, S12 = W1, W2, alloc(rows * n0 * 2 * es)             # W1/W2 are dead after the products
            _smooth_on_device(plan, mother, P, rows, n0, dt, dj, sj, spec, tmp, S)
            _smooth_on_device(plan, mother, Cx, rows, n0, dt, dj, sj, spec, tmp, S12)
            plan.wct_coherence(S.ptr, S12.ptr, rows, n0, n0, P.ptr)     # result (reals) re-uses P's storage
            if consume is not None:
                consume(plan, P, rows, n0)
                plan.sync()
                return None, None
            if keep:                                            # device-resident results: (plan, coherence, angle, scratch to free)
                plan.sync()
                return plan, P, ang, sc
            wct_ = P.download(plan, (rows, n0), plan.real).astype(np.float64, copy=False)
            awct = ang.download(plan, (rows, n0), plan.real).astype(np.float64, copy=False) if want_angle else None
            return wct_, awct
    finally:
        if pool is None and not keep:
            sc.free()


class DeviceCoherence:
    """Result of `wct_device`: the coherence R^2 (rows x n0 reals) and the phase angle stay in GPU memory; `.wct()` /
    `.angle()` download what the caller asks for, `.close()` frees the device memory.  (`wct(sig=False)` at two 2^20-point
    series is 58 ms of which ~48 are the download of two 77 x 2^20 matrices; on the device the call is ~10 ms.)"""

    def __init__(self, plan, r2, ang, scratch, shape, coi, freq):
        self._plan, self._r2, self._ang, self._sc = plan, r2, ang, scratch
        self.shape, self.coi, self.freq = shape, coi, freq

    @property
    def wct_ptr(self):
        return self._r2.ptr

    @property
    def angle_ptr(self):
        return self._ang.ptr

    def wct(self):
        return self._r2.download(self._plan, self.shape, self._plan.real).astype(np.float64, copy=False)

    def angle(self):
        return self._ang.download(self._plan, self.shape, self._plan.real).astype(np.float64, copy=False)

    def close(self):
        if self._sc is not None:
            self._sc.free()
            self._sc = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def wct_device(y1, y2, dt, dj=1 / 12, s0=-1, J=-1, wavelet="morlet", normalize=True, *, precision=None, device=0):
    """`wct(..., sig=False)` with the results left on the GPU (a `DeviceCoherence`): both transforms, the smoothing of
    wavelet.py:499-514 and the coherence ratio as in `wct`, no result matrix crosses PCIe until `.wct()` / `.angle()` is
    called.  For pipelines that consume the coherence on the device (thresholding, averaging, the Monte-Carlo histogram)."""
    mother = _check_parameter_wavelet(wavelet)
    if not hasattr(mother, "deltaj0") or mother.deltaj0 == -1 or not isinstance(mother, Morlet):
        raise AttributeError("wct needs a mother with a smoothing operator (Morlet), as in the reference")
    precision = _default_precision() if precision is None else int(precision)
    y1, y2 = np.asarray(y1), np.asarray(y2)
    if s0 == -1:
        s0 = 2 * dt / mother.flambda()
    if J == -1:
        J = int(np.round(np.log2(y1.size * dt / s0) / dj))
    sj = s0 * 2 ** (np.arange(0, J + 1) * dj)
    plan, r2, ang, sc = _coherence_on_device(_normalised(y1, normalize), _normalised(y2, normalize), dt, dj, sj, mother,
                                             precision, device, keep=True)
    return DeviceCoherence(plan, r2, ang, sc, (sj.size, y1.size), _coi(mother, y1.size, dt), 1 / (mother.flambda() * sj))


def wct(y1, y2, dt, dj=1 / 12, s0=-1, J=-1, sig=True, significance_level=0.95, wavelet="morlet",
        normalize=True, *, precision=None, device=0, **kwargs):
    """Wavelet coherence (wavelet.py:422-528).  Returns (WCT, aWCT, coi, freq, sig).

    Both transforms, the three smoothings (two of them packed into one complex pass) and the
    coherence ratio run on the GPU without W ever leaving the device.  `sig=True` runs the Monte-Carlo
    significance of `wct_significance` (300 surrogate pairs by default, cached on disk like the
    reference); pass `sig=False` to skip it.

    NOTE on the significance levels: like the reference's, they are those of WHITE surrogates -- the reference's
    `rednoise` applies its AR(1) filter along the wrong axis (helpers.py:170) and this package reproduces that seed
    for seed (`helpers.rednoise`).  `wct_significance(..., surrogates="ar1")` gives the levels of true AR(1) noise.
    The default cache file carries the reference's own name and format (`wct_sig_*_<Mother>.gz`), so caches written by
    pycwt are served and vice versa; the AR(1) variant caches under `..._ar1.gz`.
    """
    mother = _check_parameter_wavelet(wavelet)
    if not hasattr(mother, "deltaj0") or mother.deltaj0 == -1 or not isinstance(mother, Morlet):
        raise AttributeError("wct needs a mother with a smoothing operator (Morlet), as in the reference")
    precision = _default_precision() if precision is None else int(precision)
    y1, y2 = np.asarray(y1), np.asarray(y2)
    if s0 == -1:
        s0 = 2 * dt / mother.flambda()
    if J == -1:
        J = int(np.round(np.log2(y1.size * dt / s0) / dj))
    sj = s0 * 2 ** (np.arange(0, J + 1) * dj)
    freq = 1 / (mother.flambda() * sj)
    n0 = y1.size
    coi = _coi(mother, n0, dt)
    WCT, aWCT = _coherence_on_device(_normalised(y1, normalize), _normalised(y2, normalize), dt, dj, sj,
                                     mother, precision, device)
    if sig:
        a1, a2 = ar1(y1)[0], ar1(y2)[0]
        sig = wct_significance(a1, a2, dt=dt, dj=dj, s0=s0, J=J, significance_level=significance_level,
                               wavelet=mother, precision=precision, device=device, **kwargs)
    else:
        sig = np.asarray([0])
    return WCT, aWCT, coi, freq, sig


_MC_BINS = 1000


def _mc_setup(mother, dt, dj, s0, J):
    """Geometry of the Monte-Carlo surrogates (wavelet.py:591-606): series length 6*s_max/dt, scales, the part
    of every row outside the cone of influence, and the last scale that has any.

    The reference builds the rows x N mask `period <= coi` (1.5 GB at BASELINE config 5); the COI is a
    symmetric triangle, so that mask is one interval [lo_j, hi_j) per row, found here by a binary search on
    the rising half of `coi` with the same floating-point comparison."""
    N = int(np.ceil(s0 * (2 ** (J * dj)) / dt * 6))
    sj = s0 * 2 ** (np.arange(0, J + 1) * dj)
    coi = _coi(mother, N, dt)
    half = (N + 1) // 2
    lo = np.searchsorted(coi[:half], mother.flambda() * sj, side="left").astype(np.int64)
    hi = N - lo
    empty = lo >= half
    lo[empty] = 0
    hi[empty] = 0
    rows_with_data = hi > lo
    return N, sj, (lo, hi), rows_with_data, find(rows_with_data)[-1]


def _mc_histogram(draws, al1, al2, dt, dj, sj, N, outside, maxscale, mother, precision, device, progress=False,
                  ar1_surrogates=False, rng="numpy", seed=0, first_draw=0):
    """Per-scale histograms (1000 bins on [0, 1)) of the coherence of `draws` AR(1) surrogate pairs, taken
    outside the COI (wavelet.py:609-630).  Coherence AND histogram run on the GPU (`cwt_coherence_histogram`):
    per draw only the two surrogate series go up, and the rows x 1000 counters come down once at the end."""
    rows = sj.size
    lo, hi = (np.array(v, dtype=np.int64) for v in outside)        # [lo_s, hi_s): row s outside the COI
    lo[maxscale:] = 0                                               # rows >= maxscale are not counted (:625)
    hi[maxscale:] = 0
    max_span = int((hi - lo).max()) if rows else 0
    sc = _Scratch(device)
    it = range(draws)
    if progress:
        try:
            from tqdm import tqdm
            it = tqdm(it)
        except ImportError:
            pass
    try:
        plan0 = _plan(_next_pow2(N), precision, device, rows)
        lo_d, hi_d, hist_d = sc.new(rows * 8), sc.new(rows * 8), sc.new(rows * _MC_BINS * 8)
        lo_d.upload(plan0, lo)
        hi_d.upload(plan0, hi)
        hist_d.upload(plan0, np.zeros((rows, _MC_BINS), dtype=np.uint64))

        def count(plan, r2, nrows, n0):
            plan.coherence_histogram(r2.ptr, n0, nrows, lo_d.ptr, hi_d.ptr, max_span, _MC_BINS, hist_d.ptr)

        if rng == "device":
            # surrogates made on the GPU (cwt_random_normal / cwt_ar1_filter): series 2 k and 2 k + 1 of draw k, nothing
            # crosses PCIe and no host thread draws 2 N normal deviates per iteration
            es = np.dtype(plan0.real).itemsize
            tau = [int(np.ceil(-2 / np.log(np.abs(g)))) if (ar1_surrogates and g != 0) else 0 for g in (al1, al2)]
            x = [sc.new(N * es), sc.new(N * es)]
            e = sc.new((N + max(tau)) * es) if max(tau) else None
            for i in it:
                k = first_draw + i
                for w, (g, t) in enumerate(zip((al1, al2), tau)):
                    if t:
                        plan0.random_normal(seed, 2 * k + w, N + t, 1.0, e.ptr)
                        plan0.ar1_filter(e.ptr, t, N, g, x[w].ptr)
                    else:
                        plan0.random_normal(seed, 2 * k + w, N, 1.0, x[w].ptr)
                _coherence_on_device(x[0], x[1], dt, dj, sj, mother, precision, device, want_angle=False, consume=count,
                                     pool=sc, auto=(i == 0), n0=N)
            return hist_d.download(plan0, (rows, _MC_BINS), np.uint64).astype(np.float64)
        # the next surrogate pair is drawn on a helper thread while the GPU works on the current one (one
        # worker: the pairs still come from the global NumPy generator in the reference's order)
        from concurrent.futures import ThreadPoolExecutor

        def pair():
            return rednoise(N, al1, 1, ar1=ar1_surrogates), rednoise(N, al2, 1, ar1=ar1_surrogates)

        with ThreadPoolExecutor(max_workers=1) as pool:
            nxt = pool.submit(pair) if draws > 0 else None
            for i in it:
                n1, n2 = nxt.result()
                nxt = pool.submit(pair) if i + 1 < draws else None
                # (the first pair measures the dynamic range of its spectra for the accuracy target; the other 299 are
                # draws of the same process: no second look, no host synchronisation per transform)
                _coherence_on_device(n1, n2, dt, dj, sj, mother, precision, device, want_angle=False,
                                     consume=count, pool=sc, auto=(i == 0))
        return hist_d.download(plan0, (rows, _MC_BINS), np.uint64).astype(np.float64)
    finally:
        sc.free()


def _mc_percentiles(hist, rows_with_data, maxscale, significance_level):
    """wavelet.py:632-640: the `significance_level` quantile of every scale's histogram; scales inside
    the COI everywhere stay 0, scales beyond the last resolvable one NaN."""
    sig = np.zeros(hist.shape[0])
    sig[rows_with_data] = np.nan
    centres = (np.arange(_MC_BINS) + 0.5) / _MC_BINS
    for s in range(maxscale):
        sel = hist[s] > 0
        cum = hist[s, sel].cumsum()
        sig[s] = np.interp(significance_level, (cum - 0.5) / cum[-1], centres[sel])
    return sig


def _mc_cache_path(al1, al2, dt, dj, s0, J, mother, tag=""):
    with np.errstate(invalid="ignore", divide="ignore"):      # |4*al| > 1 gives nan, as in the reference
        aa = np.round(np.arctanh(np.array([al1, al2]) * 4))
    aa = np.abs(aa) + 0.5 * (aa < 0)
    return os.path.join(get_cache_dir(), "wct_sig_{:0.5f}_{:0.5f}_{:0.5f}_{:0.5f}_{:d}_{}{}.gz".format(
        aa[0], aa[1], dj, s0 / dt, int(J), mother.name, tag))


def wct_significance(al1, al2, dt, dj, s0, J, significance_level=0.95, wavelet="morlet", mc_count=300,
                     progress=True, cache=True, *, precision=None, device=0, surrogates="reference", rng="numpy", seed=None):
    """Monte-Carlo significance of the coherence (wavelet.py:531-647): `mc_count` pairs of surrogate series,
    coherence of each pair on the GPU, per-scale histogram of the values outside the cone of influence,
    `significance_level` percentile.  Scales that never leave the COI get NaN.

    Seed for seed with the reference: the surrogates come from the global NumPy generator in the reference's order
    -- one series before the loop (wavelet.py:594, drawn there only to size the arrays), then two per draw
    (:612-613) -- so `np.random.seed(k)` pins the result to the reference's within one histogram bin
    (tests/golden/mc_significance.npz).  `surrogates="reference"` (default) draws what the reference draws, which is
    white noise (see `helpers.rednoise`); `surrogates="ar1"` draws the AR(1) processes of lag-1 correlation al1, al2
    that the method calls for, and caches under a different file name.  Results are cached in the reference's file
    format under `get_cache_dir()`.  `pycwt_amd.parallel.wct_significance_sharded` splits the draws over the GPUs of
    a node.

    `rng="device"` (opt-in): the surrogates are made on the GPU (Philox4x32-10 + Box-Muller, `cwt_random_normal`; AR(1)
    filtering by `cwt_ar1_filter` for `surrogates="ar1"`) instead of by NumPy on one host thread, which is 0.10 s of a 0.13 s
    iteration at two 2^20-point series.  Same distributions, another generator: the levels agree with the NumPy path within
    the Monte-Carlo error, not seed for seed; `seed` (default: drawn from NumPy's global generator, so `np.random.seed` still
    pins the result) names the sequence.  Cached under `..._devrng.gz` (`..._devrng_seed<k>.gz` for an explicit seed)."""
    if surrogates not in ("reference", "ar1"):
        raise ValueError("surrogates must be 'reference' or 'ar1'")
    if rng not in ("numpy", "device"):
        raise ValueError("rng must be 'numpy' or 'device'")
    true_ar1 = surrogates == "ar1"
    mother = _check_parameter_wavelet(wavelet)
    precision = _default_precision() if precision is None else int(precision)
    tag = ("_ar1" if true_ar1 else "") + ("_devrng" if rng == "device" else "")
    if rng == "device" and seed is not None:
        tag += f"_seed{int(seed)}"          # an explicit seed names ANOTHER result: it must not return the first seed's cached levels
    path = _mc_cache_path(al1, al2, dt, dj, s0, J, mother, tag) if cache else None
    if cache and os.path.exists(path):
        return np.loadtxt(path, unpack=True)
    N, sj, outside, rows_with_data, maxscale = _mc_setup(mother, dt, dj, s0, J)
    if rng == "device":
        if seed is None:
            seed = int(np.random.randint(0, 2 ** 31 - 1)) * (2 ** 31) + int(np.random.randint(0, 2 ** 31 - 1))
        hist = _mc_histogram(mc_count, al1, al2, dt, dj, sj, N, outside, maxscale, mother, precision, device,
                             progress, ar1_surrogates=true_ar1, rng="device", seed=int(seed))
    else:
        rednoise(N, al1, 1, ar1=true_ar1)              # wavelet.py:594: consumed from the RNG before the loop
        hist = _mc_histogram(mc_count, al1, al2, dt, dj, sj, N, outside, maxscale, mother, precision, device,
                             progress, ar1_surrogates=true_ar1)
    sig95 = _mc_percentiles(hist, rows_with_data, maxscale, significance_level)
    if cache:
        np.savetxt(path, sig95)
    return sig95
