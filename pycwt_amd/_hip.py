"""ctypes binding of libcwt_hip.so (the C ABI declared in include/cwt_hip.h).

The product opens exactly one library: ``pycwt_amd/libcwt_hip.so``, built for gfx950 by
``pycwt_amd/_build.py``.  There is no CPU fallback: if the library is missing, cannot be loaded,
or no GPU is visible, the import / call fails loudly.
"""
from __future__ import annotations

import ctypes as C
import functools
import os
import sys
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIBRARY = os.path.join(_HERE, "libcwt_hip.so")

MORLET, PAUL, DOG = 0, 1, 2

# every symbol include/cwt_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("cwt_backend", C.c_char_p, []),
    ("cwt_build_id", C.c_char_p, []),
    ("cwt_last_error", C.c_char_p, []),
    ("cwt_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("cwt_plan_create", C.c_int, [C.POINTER(_P), C.c_int, C.c_int64, C.c_int, C.c_int]),
    ("cwt_plan_destroy", C.c_int, [_P]),
    ("cwt_plan_set_stream", C.c_int, [_P, _P]),
    ("cwt_plan_set_option", C.c_int, [_P, C.c_char_p, C.c_int64]),
    ("cwt_plan_sync", C.c_int, [_P]),
    ("cwt_device_memory", C.c_int, [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    ("cwt_device_synchronize", C.c_int, [C.c_int]),
    ("cwt_malloc", C.c_int, [C.c_int, C.POINTER(_P), C.c_size_t]),
    ("cwt_free", C.c_int, [C.c_int, _P]),
    ("cwt_memcpy_h2d", C.c_int, [_P, _P, _P, C.c_size_t]),
    ("cwt_memcpy_d2h", C.c_int, [_P, _P, _P, C.c_size_t]),
    ("cwt_host_malloc", C.c_int, [C.POINTER(_P), C.c_size_t]),
    ("cwt_host_free", C.c_int, [_P]),
    ("cwt_forward_fft", C.c_int, [_P, _P, C.c_int64, _P]),
    ("cwt_transform_rows", C.c_int, [_P, _P, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double),
                                     C.c_int, _P, C.c_int64, C.c_int64]),
    ("cwt_transform", C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double),
                                C.c_int, _P, _P, C.c_int64, C.c_int64]),
    ("cwt_forward_fft_n", C.c_int, [_P, _P, C.c_int64, _P]),
    ("cwt_transform_rows_n", C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double),
                                       C.c_int, _P, C.c_int64]),
    ("cwt_transform_rows_batch", C.c_int, [_P, _P, C.c_int, C.c_int64, C.c_int, C.c_double, C.c_double,
                                           C.POINTER(C.c_double), C.c_int, _P, C.c_int64, C.c_int64]),
    ("cwt_transform_batch", C.c_int, [_P, _P, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_double, C.c_double,
                                      C.POINTER(C.c_double), C.c_int, _P, _P, C.c_int64, C.c_int64]),
    ("cwt_transform_rows_table", C.c_int, [_P, _P, _P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, _P,
                                           C.c_int64, C.c_int64]),
    ("cwt_fft_rows", C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int64, C.c_int64, _P]),
    ("cwt_filter_rows", C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_double, C.POINTER(C.c_double),
                                  C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, _P, C.c_int64, C.c_int64]),
    ("cwt_boxcar_scales", C.c_int, [_P, _P, C.c_int, C.c_int64, C.c_int64, C.POINTER(C.c_double), C.c_int, _P]),
    ("cwt_cross_spectrum", C.c_int, [_P, _P, _P, C.c_int, C.c_int64, C.c_int64, _P]),
    ("cwt_wct_products", C.c_int, [_P, _P, _P, C.POINTER(C.c_double), C.c_int, C.c_int64, C.c_int64, _P, _P, _P]),
    ("cwt_wct_coherence", C.c_int, [_P, _P, _P, C.c_int, C.c_int64, C.c_int64, _P]),
    ("cwt_icwt_reduce", C.c_int, [_P, _P, C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_double),
                                  C.c_double, _P]),
    ("cwt_reduce_scales", C.c_int, [_P, _P, C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_double), C.c_int,
                                    C.c_double, _P]),
    ("cwt_time_mean_power", C.c_int, [_P, _P, C.c_int64, C.c_int64, C.c_int, _P]),
    ("cwt_coherence_histogram", C.c_int, [_P, _P, C.c_int64, C.c_int, _P, _P, C.c_int64, C.c_int, _P]),
    ("cwt_execute_host", C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_double, C.c_double,
                                   C.POINTER(C.c_double), C.c_int, _P, _P]),
    ("cwt_plan_timings", C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                   C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("cwt_plan_row_classes", C.c_int, [_P, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]),
    ("cwt_plan_classify", C.c_int, [_P, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double), C.c_int, C.c_int64,
                                    C.c_int, C.POINTER(C.c_int)]),
    ("cwt_plan_last_split", C.c_int, [_P, C.POINTER(C.c_int)]),
    ("cwt_plan_last_split8", C.c_int, [_P, C.POINTER(C.c_int)]),
    ("cwt_plan_balanced_shards", C.c_int, [_P, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double), C.c_int, C.c_int64,
                                           C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("cwt_shard_codes", C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_int),
                                  C.POINTER(C.c_int)]),
    ("cwt_shard_cost", C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_double)]),
    ("cwt_plan_set_tolerance", C.c_int, [_P, C.c_double]),
    ("cwt_plan_get_tolerance", C.c_int, [_P, C.POINTER(C.c_double)]),
    ("cwt_plan_set_auto_tolerance", C.c_int, [_P, C.c_double]),
    ("cwt_spectrum_range", C.c_int, [_P, _P, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("cwt_plan_auto_tolerance", C.c_int, [_P, _P, C.c_double, C.POINTER(C.c_double)]),
    ("cwt_random_normal", C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_int64, C.c_double, _P]),
    ("cwt_ar1_filter", C.c_int, [_P, _P, C.c_int64, C.c_int64, C.c_double, _P]),
]


class HipError(RuntimeError):
    """A C-ABI call returned a negative status (`code`: CWT_EINVAL -1, CWT_EHIP -2, CWT_ENOMEM -3, CWT_ENODEV -4)."""

    def __init__(self, message, code=None):
        super().__init__(message)
        self.code = code


class Library:
    """A loaded libcwt_hip.so with typed entry points."""

    def __init__(self, path: str = DEFAULT_LIBRARY):
        if not os.path.exists(path):
            raise ImportError(
                f"{path} not found: the HIP extension has not been built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or python -m pycwt_amd._build). "
                "pycwt_amd has no CPU fallback.")
        self.path = path
        self.dll = C.CDLL(path)
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(self.dll, name)      # AttributeError if the symbol is not exported
            fn.restype = restype
            fn.argtypes = argtypes
            setattr(self, name, fn)
        self.pinned = PinnedPool(self)

    def backend(self) -> str:
        return self.cwt_backend().decode()

    def build_id(self) -> str:
        """Identity of the sources this binary was built from (`_build.source_id()` of its tree at build time)."""
        return self.cwt_build_id().decode()

    def check(self, rc: int):
        if rc != 0:
            raise HipError(f"libcwt_hip error {rc}: {self.cwt_last_error().decode()}", rc)

    def device_memory(self, device: int = 0):
        """(free, total) bytes of device memory."""
        f, t = C.c_size_t(0), C.c_size_t(0)
        self.check(self.cwt_device_memory(device, C.byref(f), C.byref(t)))
        return f.value, t.value

    def device_synchronize(self, device: int = 0):
        """Everything queued on any stream of the device has finished on return."""
        self.check(self.cwt_device_synchronize(device))

    def device_count(self) -> int:
        n = C.c_int(0)
        self.check(self.cwt_device_count(C.byref(n)))
        return n.value


class _PinnedSlot:
    """Returns a page-locked buffer to its pool when the last array (or view) on it is gone."""
    __slots__ = ("pool", "ptr", "size")

    def __init__(self, pool, ptr, size):
        self.pool, self.ptr, self.size = pool, ptr, size

    def __del__(self):
        try:
            self.pool._release(self.ptr, self.size)
        except Exception:           # interpreter shutdown: the process is going away with its mappings
            pass


class PinnedPool:
    """NumPy result arrays in page-locked host memory (cwt_host_malloc).  To the caller they are ordinary arrays; to
    cwt_execute_host they are buffers the kernels of a short transform write over PCIe themselves, which removes the
    staging copy of W -- 25 of the ~85 us of the reference's canonical 504-point call.  Buffers go back to the pool when
    the array and every view of it are garbage; at most LIMIT bytes are ever pinned, then (or for results above ONE)
    `empty` returns None and the caller uses pageable memory."""
    LIMIT = 256 << 20
    ONE = 16 << 20
    GRAIN = 1 << 16

    def __init__(self, lib):
        self.lib = lib
        self.free = {}
        self.total = 0
        # re-entrant: _release runs from __del__, and a garbage collection that starts while this thread holds the lock
        # (any allocation inside it can trigger one) may finalise another pinned array on the same thread
        self.lock = threading.RLock()

    def empty(self, shape, dtype):
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        if nbytes == 0 or nbytes > self.ONE:
            return None
        size = (nbytes + self.GRAIN - 1) & ~(self.GRAIN - 1)
        with self.lock:
            stack = self.free.setdefault(size, [])            # (the free list exists before any release needs it)
            ptr = stack.pop() if stack else None
            if ptr is None:
                if self.total + size > self.LIMIT:
                    return None
                self.total += size
        if ptr is None:
            p = _P()
            if self.lib.cwt_host_malloc(C.byref(p), size) != 0 or not p.value:
                with self.lock:
                    self.total -= size
                return None
            ptr = p.value
        buf = (C.c_char * nbytes).from_address(ptr)
        buf._slot = _PinnedSlot(self, ptr, size)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def _release(self, ptr, size):
        with self.lock:
            self.free.setdefault(size, []).append(ptr)


_default = None


def _one_hip_runtime():
    """PyTorch-ROCm wheels carry their own libamdhip64 / libhsa-runtime64 under the system's SONAMEs.  Whichever copy is
    loaded first serves the whole process: with the system runtime first, torch (built against its own) finds no GPU --
    `torch.cuda.is_available()` is False and tensors silently land on the CPU.  So when torch is installed, let it load
    its runtime before libcwt_hip.so resolves the same SONAMEs (the library itself is torch-free; a C host is not
    affected).  PYCWT_AMD_NO_TORCH_PRELOAD=1 skips this."""
    if "torch" in sys.modules or os.environ.get("PYCWT_AMD_NO_TORCH_PRELOAD"):
        return
    import importlib.util
    if importlib.util.find_spec("torch") is not None:
        try:
            import torch  # noqa: F401
        except Exception:       # a broken torch install must not take the transform down with it
            pass


def load() -> Library:
    """The product library (HIP, gfx950).  Raises if it is missing."""
    global _default
    if _default is None:
        _one_hip_runtime()
        _check_provenance()
        _default = Library(DEFAULT_LIBRARY)
    return _default


def _check_provenance():
    """A library built from other sources than this tree's would be used -- and benchmarked -- silently: compare the build id
    embedded in the file with the tree's and rebuild on a mismatch (refuse where there is no compiler;
    PYCWT_AMD_ALLOW_STALE=1 uses the file as it is).  Installed copies without the sources are taken as they are."""
    from . import _build
    if not os.path.exists(DEFAULT_LIBRARY) or os.environ.get("PYCWT_AMD_ALLOW_STALE"):
        return
    if not all(os.path.exists(d) for d in _build.DEPS):
        return
    have, want = _build.library_id(DEFAULT_LIBRARY), _build.source_id()
    if have == want:
        return
    try:
        _build.build()
    except Exception as e:
        raise ImportError(f"{DEFAULT_LIBRARY} was built from other sources than this tree (library {have}, tree {want}) and cannot be "
                          f"rebuilt here ({e}).  PYCWT_AMD_ALLOW_STALE=1 loads it anyway.") from e


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _locked(method):
    """Serialise the calls of one plan: the C plan is not thread-safe (shared row table, workspaces, stream) and
    ctypes drops the GIL during a call.  Re-entrant, so a caller may hold `plan.lock` across a whole
    upload -> transform -> download sequence while the methods take it again."""
    @functools.wraps(method)
    def wrapper(self, *a, **kw):
        with self.lock:
            try:
                return method(self, *a, **kw)
            except HipError as e:
                # an allocation INSIDE the library failed (scratch of a transform, the buffers of cwt_execute_host, a row table):
                # give back what the host side keeps (wavelet.release_scratch) and try once more, as DeviceBuffer does
                if e.code != -3 or not on_allocation_failure:
                    raise
            for release in list(on_allocation_failure):
                release()
            return method(self, *a, **kw)
    return wrapper


class Plan:
    """One (device, nfft, precision) plan; thin RAII wrapper over the C ABI.  Every method holds `self.lock`
    (an RLock) for the duration of its C call; multi-call sequences that must not interleave with another
    thread's (everything in pycwt_amd.wavelet) hold it around the sequence."""

    def __init__(self, nfft: int, precision: int = 64, max_rows: int = 1024, device: int = 0,
                 lib: Library | None = None, options: dict | None = None):
        self.lib = lib or load()
        self.nfft = int(nfft)
        self.precision = int(precision)
        self.device = device
        self.max_rows = int(max_rows)
        self.real = np.float64 if precision == 64 else np.float32
        self.cplx = np.complex128 if precision == 64 else np.complex64
        self.lock = threading.RLock()
        h = _P()
        self.lib.check(self.lib.cwt_plan_create(C.byref(h), device, self.nfft, self.precision, self.max_rows))
        self.h = h
        for k, v in (options or {}).items():
            if k == "tolerance":
                self.set_tolerance(v)
            elif k == "auto_tolerance":
                self.set_auto_tolerance(v)
            else:
                self.set_option(k, v)

    def close(self):
        lock = getattr(self, "lock", None)
        if lock is None:                     # __init__ failed before the handle existed
            return
        with lock:
            h, self.h = getattr(self, "h", None), None
        if h:
            try:
                self.lib.cwt_plan_destroy(h)
            except Exception:       # interpreter shutdown: the library may already be gone
                pass

    __del__ = close

    @_locked
    def set_option(self, key: str, value: int):
        self.lib.check(self.lib.cwt_plan_set_option(self.h, key.encode(), int(value)))

    @_locked
    def set_tolerance(self, rel_tol: float):
        """Accuracy target per row of W (max|dW| / max|W|); 0 = the precision's default (see cwt_plan_set_tolerance)."""
        self.lib.check(self.lib.cwt_plan_set_tolerance(self.h, float(rel_tol)))

    @_locked
    def set_auto_tolerance(self, target: float):
        """execute_host() derives each call's tolerance from `target` and the dynamic range of the call's spectrum
        (cwt_plan_set_auto_tolerance); 0 = off."""
        self.lib.check(self.lib.cwt_plan_set_auto_tolerance(self.h, float(target)))

    @_locked
    def spectrum_range(self, xhat_dev: int, n: int):
        """(max|xhat|, rms|xhat|, rms of the quietest 3/4-octave stretch) of a device-resident spectrum."""
        mx, rms, fl = C.c_double(0), C.c_double(0), C.c_double(0)
        self.lib.check(self.lib.cwt_spectrum_range(self.h, _P(xhat_dev), n, C.byref(mx), C.byref(rms), C.byref(fl)))
        return mx.value, rms.value, fl.value

    @_locked
    def auto_tolerance(self, xhat_dev: int, target: float) -> float:
        """The filter-relative tolerance that holds `target` for the device-resident spectrum of THIS call
        (cwt_plan_auto_tolerance: one formula for the C host path and the Python shim)."""
        v = C.c_double(0)
        self.lib.check(self.lib.cwt_plan_auto_tolerance(self.h, _P(xhat_dev), float(target), C.byref(v)))
        return v.value

    @_locked
    def random_normal(self, seed: int, offset: int, n: int, scale: float, out_dev: int):
        """out_dev[0:n] = scale * N(0, 1) made on the device (cwt_random_normal: Philox4x32-10, reproducible per
        (seed, offset, index))."""
        self.lib.check(self.lib.cwt_random_normal(self.h, int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), int(n),
                                                  float(scale), _P(out_dev)))

    @_locked
    def ar1_filter(self, e_dev: int, tau: int, n: int, g: float, out_dev: int):
        """out_dev[0:n] = lfilter([1, 0], [1, -g], e)[tau:] for e_dev[0:tau + n] (cwt_ar1_filter)."""
        self.lib.check(self.lib.cwt_ar1_filter(self.h, _P(e_dev), int(tau), int(n), float(g), _P(out_dev)))

    @_locked
    def tolerance(self) -> float:
        v = C.c_double(0)
        self.lib.check(self.lib.cwt_plan_get_tolerance(self.h, C.byref(v)))
        return v.value

    @_locked
    def set_stream(self, stream_handle: int):
        self.lib.check(self.lib.cwt_plan_set_stream(self.h, _P(stream_handle)))

    @_locked
    def sync(self):
        self.lib.check(self.lib.cwt_plan_sync(self.h))

    # -- device-resident entry points (raw device addresses as ints) --
    @_locked
    def forward_fft(self, x_dev: int, n0: int, xhat_dev: int):
        self.lib.check(self.lib.cwt_forward_fft(self.h, _P(x_dev), n0, _P(xhat_dev)))

    @_locked
    def transform_rows(self, xhat_dev: int, mother: int, param: float, dt: float, scales, W_dev: int,
                       ldw: int, ncols: int):
        s = np.ascontiguousarray(scales, dtype=np.float64)
        self.lib.check(self.lib.cwt_transform_rows(self.h, _P(xhat_dev), mother, float(param), float(dt),
                                                   _dptr(s), s.size, _P(W_dev), ldw, ncols))

    @_locked
    def transform(self, x_dev: int, n0: int, mother: int, param: float, dt: float, scales, xhat_dev, W_dev: int,
                  ldw: int, ncols: int):
        """forward_fft + transform_rows in one call; the library may use the signal itself (overlap-save rows).
        xhat_dev = None: the spectrum is not wanted (computed into plan scratch only if some row needs it)."""
        s = np.ascontiguousarray(scales, dtype=np.float64)
        self.lib.check(self.lib.cwt_transform(self.h, _P(x_dev), n0, mother, float(param), float(dt), _dptr(s), s.size,
                                              _P(xhat_dev) if xhat_dev else None, _P(W_dev), ldw, ncols))

    @_locked
    def forward_fft_n(self, x_dev: int, n0: int, xhat_dev: int):
        """Forward transform at length n0 (not a power of two; this plan's nfft >= 2*n0 - 1)."""
        self.lib.check(self.lib.cwt_forward_fft_n(self.h, _P(x_dev), n0, _P(xhat_dev)))

    @_locked
    def transform_rows_n(self, xhat_dev: int, n0: int, mother: int, param: float, dt: float, scales, W_dev: int,
                         ldw: int):
        s = np.ascontiguousarray(scales, dtype=np.float64)
        self.lib.check(self.lib.cwt_transform_rows_n(self.h, _P(xhat_dev), n0, mother, float(param), float(dt),
                                                     _dptr(s), s.size, _P(W_dev), ldw))

    @_locked
    def transform_rows_batch(self, xhat_dev: int, nbatch: int, xhat_ld: int, mother: int, param: float,
                             dt: float, scales, W_dev: int, ldw: int, ncols: int):
        s = np.ascontiguousarray(scales, dtype=np.float64)
        self.lib.check(self.lib.cwt_transform_rows_batch(self.h, _P(xhat_dev), nbatch, xhat_ld, mother,
                                                         float(param), float(dt), _dptr(s), s.size, _P(W_dev),
                                                         ldw, ncols))

    @_locked
    def transform_batch(self, x_dev: int, nbatch: int, x_ld: int, n0: int, mother: int, param: float, dt: float,
                        scales, xhat_dev: int, W_dev: int, ldw: int, ncols: int):
        """Forward transforms + rows of a batch of signals (cwt_transform_batch): xhat_dev nbatch x nfft complex."""
        s = np.ascontiguousarray(scales, dtype=np.float64)
        self.lib.check(self.lib.cwt_transform_batch(self.h, _P(x_dev), nbatch, x_ld, n0, mother, float(param),
                                                    float(dt), _dptr(s), s.size, _P(xhat_dev), _P(W_dev), ldw, ncols))

    @_locked
    def transform_rows_table(self, xhat_dev: int, table_dev: int, k_lo, nband, W_dev: int, ldw: int, ncols: int):
        k = np.ascontiguousarray(k_lo, dtype=np.int32)
        b = np.ascontiguousarray(nband, dtype=np.int32)
        self.lib.check(self.lib.cwt_transform_rows_table(
            self.h, _P(xhat_dev), _P(table_dev), k.ctypes.data_as(C.POINTER(C.c_int)),
            b.ctypes.data_as(C.POINTER(C.c_int)), k.size, _P(W_dev), ldw, ncols))

    @_locked
    def fft_rows(self, in_dev: int, in_complex: bool, nrows: int, in_ld: int, ncols_in: int, spec_dev: int):
        self.lib.check(self.lib.cwt_fft_rows(self.h, _P(in_dev), int(in_complex), nrows, in_ld, ncols_in,
                                             _P(spec_dev)))

    @_locked
    def filter_rows(self, spec_dev: int, spec_ld: int, mother: int, param: float, a, amp, W_dev: int, ldw: int,
                    ncols: int):
        a = np.ascontiguousarray(a, dtype=np.float64)
        amp = np.asarray(amp, dtype=np.complex128) * np.ones(a.size)
        ar, ai = np.ascontiguousarray(amp.real), np.ascontiguousarray(amp.imag)
        self.lib.check(self.lib.cwt_filter_rows(self.h, _P(spec_dev), spec_ld, mother, float(param), _dptr(a),
                                                _dptr(ar), _dptr(ai), a.size, _P(W_dev), ldw, ncols))

    @_locked
    def boxcar_scales(self, in_dev: int, nrows: int, ld: int, ncols: int, win, out_dev: int):
        w = np.ascontiguousarray(win, dtype=np.float64)
        self.lib.check(self.lib.cwt_boxcar_scales(self.h, _P(in_dev), nrows, ld, ncols, _dptr(w), w.size, _P(out_dev)))

    @_locked
    def wct_products(self, W1_dev: int, W2_dev: int, scales, ld: int, ncols: int, P_dev: int, C_dev: int,
                     angle_dev: int):
        s = np.ascontiguousarray(scales, dtype=np.float64)
        self.lib.check(self.lib.cwt_wct_products(self.h, _P(W1_dev), _P(W2_dev), _dptr(s), s.size, ld, ncols,
                                                 _P(P_dev), _P(C_dev), _P(angle_dev)))

    @_locked
    def wct_coherence(self, S_dev: int, S12_dev: int, nrows: int, ld: int, ncols: int, out_dev: int):
        self.lib.check(self.lib.cwt_wct_coherence(self.h, _P(S_dev), _P(S12_dev), nrows, ld, ncols, _P(out_dev)))

    @_locked
    def icwt_reduce(self, W_dev: int, ldw: int, ncols: int, scales, coeff: float, out_dev: int):
        s = np.ascontiguousarray(scales, dtype=np.float64)
        self.lib.check(self.lib.cwt_icwt_reduce(self.h, _P(W_dev), ldw, ncols, s.size, _dptr(s),
                                                float(coeff), _P(out_dev)))

    @_locked
    def reduce_scales(self, W_dev: int, ldw: int, ncols: int, weights, power: bool, coeff: float, out_dev: int):
        w = np.ascontiguousarray(weights, dtype=np.float64)
        self.lib.check(self.lib.cwt_reduce_scales(self.h, _P(W_dev), ldw, ncols, w.size, _dptr(w), int(power),
                                                  float(coeff), _P(out_dev)))

    @_locked
    def time_mean_power(self, W_dev: int, ldw: int, ncols: int, nrows: int, out_dev: int):
        self.lib.check(self.lib.cwt_time_mean_power(self.h, _P(W_dev), ldw, ncols, nrows, _P(out_dev)))

    @_locked
    def cross_spectrum(self, W1_dev: int, W2_dev: int, nrows: int, ld: int, ncols: int, out_dev: int):
        self.lib.check(self.lib.cwt_cross_spectrum(self.h, _P(W1_dev), _P(W2_dev), nrows, ld, ncols, _P(out_dev)))

    @_locked
    def coherence_histogram(self, r2_dev: int, ld: int, nrows: int, lo_dev: int, hi_dev: int, max_span: int,
                            nbins: int, hist_dev: int):
        self.lib.check(self.lib.cwt_coherence_histogram(self.h, _P(r2_dev), ld, nrows, _P(lo_dev), _P(hi_dev),
                                                        int(max_span), nbins, _P(hist_dev)))

    # -- host convenience --
    @_locked
    def execute_host(self, x, mother: int, param: float, dt: float, scales, want_W=True, want_xhat=True):
        x = np.ascontiguousarray(x, dtype=self.real)
        s = np.ascontiguousarray(scales, dtype=np.float64)
        n0 = x.size
        W = None
        if want_W:
            if self.nfft <= 4096:       # one kernel writes W over PCIe: into page-locked memory when the pool has some
                W = self.lib.pinned.empty((s.size, n0), self.cplx)
            if W is None:
                W = np.empty((s.size, n0), dtype=self.cplx)
        xhat = np.empty(self.nfft, dtype=self.cplx) if want_xhat else None
        rc = self.lib.cwt_execute_host(
            self.h, x.ctypes.data, n0, mother, param, dt, _dptr(s), s.size,
            W.ctypes.data if want_W else None, xhat.ctypes.data if want_xhat else None)
        if rc:
            self.lib.check(rc)
        return W, xhat

    @_locked
    def timings(self):
        cap = 16
        names = (C.c_char_p * cap)()
        ms = (C.c_double * cap)()
        cnt = (C.c_int * cap)()
        n = C.c_int(0)
        self.lib.check(self.lib.cwt_plan_timings(self.h, cap, names, ms, cnt, C.byref(n)))
        return {names[i].decode(): (ms[i], cnt[i]) for i in range(min(n.value, cap))}

    _KINDS = ("single_wg", "narrow", "narrow_k2048", "two_pass")

    @_locked
    def row_classes(self):
        """Kernel class of every row of the last transform call, as labels like 'narrow/K1024/t3',
        'narrow_k2048/t4', 'two_pass/full', 'two_pass/c64', 'ols/K512' (see cwt_plan_row_classes)."""
        n = C.c_int(0)
        self.lib.check(self.lib.cwt_plan_row_classes(self.h, None, 0, C.byref(n)))
        codes = (C.c_int * max(n.value, 1))()
        self.lib.check(self.lib.cwt_plan_row_classes(self.h, codes, n.value, C.byref(n)))
        return self._labels(codes[:n.value])

    @_locked
    def classify(self, mother: int, param: float, dt: float, scales, ncols: int, with_signal: bool = True):
        """The labels `row_classes()` would report after transform() (with_signal) / transform_rows() with these
        arguments; nothing is launched."""
        s = np.ascontiguousarray(scales, dtype=np.float64)
        codes = (C.c_int * max(s.size, 1))()
        self.lib.check(self.lib.cwt_plan_classify(self.h, mother, float(param), float(dt), _dptr(s), s.size, ncols,
                                                  int(with_signal), codes))
        return self._labels(codes[:s.size])

    @_locked
    def balanced_shards(self, mother: int, param: float, dt: float, scales, ncols: int, world: int):
        """Contiguous cost-balanced shards of the scale grid for `world` ranks (cwt_plan_balanced_shards): list of index
        arrays, identical on every rank."""
        s = np.ascontiguousarray(scales, dtype=np.float64)
        first, count = (C.c_int * world)(), (C.c_int * world)()
        self.lib.check(self.lib.cwt_plan_balanced_shards(self.h, mother, float(param), float(dt), _dptr(s), s.size, ncols,
                                                         world, first, count))
        return [np.arange(first[r], first[r] + count[r]) for r in range(world)]

    @staticmethod
    def codes_of(labels):
        """Inverse of `_labels`: row-class codes of label strings."""
        out = []
        for lab in labels:
            parts = lab.split("/")
            if parts[0] == "single_wg":
                out.append(0)
            elif parts[0] == "two_pass":
                out.append(30000 + (0 if parts[1] == "full" else int(np.log2(int(parts[1][1:])))) * 100 + 1)
            elif parts[0] == "narrow_k2048":
                out.append(20000 + 1100 + int(parts[1][1:]))
            elif parts[0] == "poly":
                out.append(70000 + int(np.log2(int(parts[1][1:]))) * 100 + int(parts[2][1:]))
            elif parts[0] == "aols":
                out.append(60000 + int(np.log2(int(parts[1][1:]))) * 100 + 1)
            elif parts[0].startswith("ols"):
                terms = int(parts[0][3:]) if len(parts[0]) > 3 else 1
                kind = 5 if parts[-1] == "half" else 4
                out.append(kind * 10000 + int(np.log2(int(parts[1][1:]))) * 100 + terms)
            else:
                terms = int(parts[2][1:]) if len(parts) > 2 else 1
                out.append(10000 + int(np.log2(int(parts[1][1:]))) * 100 + terms)
        return out

    @staticmethod
    def _labels(codes):
        out = []
        for c in codes:
            kind, logk, terms = c // 10000, (c // 100) % 100, c % 100
            if kind == 0:
                out.append("single_wg")
            elif kind == 3:
                out.append("two_pass/" + ("full" if logk == 0 else f"c{1 << logk}"))
            elif kind == 2:
                out.append(f"narrow_k2048/t{terms}")
            elif kind == 4:
                out.append(("ols/K" if terms == 1 else f"ols{terms}/K") + str(1 << logk))
            elif kind == 5:                                  # overlap-save row on half-size workgroup tiles
                out.append(f"ols/K{1 << logk}/half")
            elif kind == 7:                                  # band-limited row in polynomial form: K' intervals, degree d
                out.append(f"poly/K{1 << logk}/d{terms}")
            elif kind == 6:                                  # row clipped at Nyquist: overlap-save on the band-passed signal
                out.append(f"aols/P{1 << logk}")
            else:
                out.append(f"narrow/K{1 << logk}" + (f"/t{terms}" if terms > 1 else ""))
        return out

    @_locked
    def last_split(self):
        c = (C.c_int * 8)()
        self.lib.check(self.lib.cwt_plan_last_split8(self.h, c))
        return {"small": c[0], "narrow": c[1] + c[3] + c[4], "two_pass": c[2], "narrow_k2048": c[3], "narrow_many": c[4],
                "ols": c[5], "aols": c[6], "poly": c[7]}


on_allocation_failure: list = []      # callables that give device memory back; run once before an allocation is retried


class DeviceBuffer:
    """Device memory owned through cwt_malloc / cwt_free (used by the NumPy-only host path)."""

    def __init__(self, nbytes: int, device: int = 0, lib: Library | None = None):
        self.lib = lib or load()
        self.device = device
        self.nbytes = int(nbytes)
        p = _P()
        try:
            self.lib.check(self.lib.cwt_malloc(device, C.byref(p), self.nbytes))
        except HipError:
            for release in list(on_allocation_failure):       # (the shim's pool of kept work matrices, wavelet.release_scratch)
                release()
            self.lib.check(self.lib.cwt_malloc(device, C.byref(p), self.nbytes))
        self.ptr = p.value

    def free(self):
        ptr, self.ptr = getattr(self, "ptr", None), None
        if ptr:
            try:
                self.lib.cwt_free(self.device, _P(ptr))
            except Exception:       # interpreter shutdown
                pass

    __del__ = free

    def upload(self, plan: Plan, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        with plan.lock:
            self.lib.check(self.lib.cwt_memcpy_h2d(plan.h, _P(self.ptr), arr.ctypes.data_as(_P), arr.nbytes))

    def download(self, plan: Plan, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        with plan.lock:
            self.lib.check(self.lib.cwt_memcpy_d2h(plan.h, out.ctypes.data_as(_P), _P(self.ptr), out.nbytes))
        return out
