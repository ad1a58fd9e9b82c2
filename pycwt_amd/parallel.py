"""Scale-sharded CWT across the GPUs of one node (one process per GPU, torch.distributed).

The rows (scales) of W are independent once the signal is known (pycwt/wavelet.py:102-106 is an
outer product followed by independent inverse FFTs), so the only exchange of the path is ONE
broadcast per transform: rank 0 owns the signal and broadcasts it (8 MiB at N = 2^20 fp64; backend
"nccl" = RCCL over xGMI); every rank then runs the forward FFT itself (1/rows of its work, cheaper
than moving the 16 MiB spectrum) and computes its rows into a device-resident shard.  The default shards are
contiguous runs of scales of equal ESTIMATED COST (`balanced_shards`: the library classifies every row first,
`Plan.classify`); `partition="interleaved"` gives rows j = rank, rank + G, ... instead.  W is never gathered (4 GiB would dwarf the compute); `icwt_sharded` reduces
per-rank partial column sums with one `reduce`.

torch is plumbing here: device memory, the current stream and the process group.  The compute goes
through the C ABI (`_hip.Plan`) via `HipEngine`; tests substitute a CPU engine to exercise the
sharding / collective logic under the gloo backend.
"""
from __future__ import annotations

import ctypes as C
import functools

import numpy as np

from . import _hip
from .wavelet import _check_parameter_wavelet, _device_id, _next_pow2


def shard_rows(nrows: int, world: int, rank: int) -> np.ndarray:
    """Indices of the rows owned by `rank` (interleaved)."""
    return np.arange(rank, nrows, world)


# The cost model and the search live in the C library (cwt_shard_codes / cwt_shard_cost / cwt_plan_balanced_shards, so that
# a host in any language gets the same shards): a fixed part per kernel class present (launch ramp and tail, block spectra)
# + a per-row part, fitted to the per-class launch durations of bench.py (profiles/r03_per_class.txt, r03_shards.txt).
def _lib(lib=None):
    return lib or _hip._default or _hip.load()


def _chunk_rows(precision: int, nfft: int) -> int:
    """Rows per two-pass launch pair of a default plan: the intermediate must fit 192 MiB (Infinity Cache)."""
    return max(1, (192 << 20) // (int(nfft) * (16 if precision == 64 else 8)))


def shard_cost(labels, precision: int = 64, nfft: int = 1 << 20, lib=None) -> float:
    """The model's estimate (microseconds per step) for a rank that owns the rows with these class labels."""
    codes = _hip.Plan.codes_of(labels)
    arr = (C.c_int * max(len(codes), 1))(*codes)
    out = C.c_double(0)
    L = _lib(lib)
    L.check(L.cwt_shard_cost(arr, len(codes), precision, float(nfft) / float(1 << 20), _chunk_rows(precision, nfft), C.byref(out)))
    return out.value


def balanced_shards(labels, world: int, precision: int = 64, nfft: int = 1 << 20, lib=None):
    """Cuts the scale grid (rows in scale order, `labels` = their kernel classes from `Plan.classify`) into `world`
    CONTIGUOUS shards of equal estimated cost (cwt_shard_codes).  Against interleaving (row j -> rank j mod G) a rank
    then runs few kernel classes with many rows each instead of every class with a handful -- at 8 ranks the
    interleaved share is 2 two-pass rows, 2 K = 2048 rows, 10 overlap-save rows ..., all launch-latency bound -- and
    ranks whose rows are all overlap-save never need the forward FFT.  Returns a list of index arrays (possibly empty);
    cached per (labels, world, precision, length)."""
    return [np.array(s) for s in _balanced_shards(tuple(labels), world, precision, int(nfft), _lib(lib))]


@functools.lru_cache(maxsize=64)
def _balanced_shards(labels, world, precision, nfft, lib):
    codes = _hip.Plan.codes_of(labels)
    arr = (C.c_int * max(len(codes), 1))(*codes)
    first, count = (C.c_int * world)(), (C.c_int * world)()
    lib.check(lib.cwt_shard_codes(arr, len(codes), precision, float(nfft) / float(1 << 20), _chunk_rows(precision, nfft),
                                  world, first, count))
    return [np.arange(first[r], first[r] + count[r]) for r in range(world)]


_engines: dict = {}       # default engines of cwt_sharded, one per (length, precision, device): keeps the row-table cache


class HipEngine:
    """Runs the hot path on this rank's GPU through libcwt_hip.so on torch's current stream."""

    def __init__(self, nfft: int, precision: int, max_rows: int, device_index: int, on_torch_stream: bool = True,
                 options=None, lib=None):
        import torch
        self.torch = torch
        if not on_torch_stream and (lib or _hip.load()).backend().startswith("hip"):
            # tensors that live on the host would hand host pointers to the kernels: a GPU memory fault, not an error
            # (on_torch_stream=False is for the test suite's CPU emulation of the library)
            raise RuntimeError("HipEngine needs tensors on a GPU and torch sees none (torch.cuda.is_available() is False: "
                               "was another HIP runtime loaded before torch was imported?)")
        self.plan = _hip.Plan(nfft, precision, max_rows=max_rows, device=device_index, lib=lib, options=options)
        if on_torch_stream:               # tensors that live on a GPU: queue behind torch's work on its stream
            self.plan.set_stream(torch.cuda.current_stream(device_index).cuda_stream)

    def forward(self, x, n0, xhat):
        """x: (n0,) or (batch, n0) reals -> xhat: (N,) or (batch, N) spectra."""
        if x.dim() == 1:
            self.plan.forward_fft(x.data_ptr(), n0, xhat.data_ptr())
        else:
            self.plan.fft_rows(x.data_ptr(), False, x.shape[0], x.shape[1], n0, xhat.data_ptr())

    def rows(self, xhat, kind, param, dt, sj, W, ncols):
        """W: (rows, n0) for one signal, (batch, rows, n0) for a batch."""
        if xhat.dim() == 1:
            self.plan.transform_rows(xhat.data_ptr(), kind, param, dt, sj, W.data_ptr(), W.shape[-1], ncols)
        else:
            self.plan.transform_rows_batch(xhat.data_ptr(), xhat.shape[0], xhat.shape[1], kind, param, dt, sj,
                                           W.data_ptr(), W.shape[-1], ncols)

    def transform(self, x, n0, xhat, kind, param, dt, sj, W, ncols):
        """forward + rows; for one signal in ONE C call (cwt_transform), which lets the library use the signal itself
        for time-compact rows (overlap-save).  xhat = None: spectrum not wanted (plan scratch, skipped if unused)."""
        if x.dim() == 1:
            self.plan.transform(x.data_ptr(), n0, kind, param, dt, sj, None if xhat is None else xhat.data_ptr(),
                                W.data_ptr(), W.shape[-1], ncols)
        else:                                             # a batch: cwt_transform_batch (spectra to xhat, required)
            if xhat is None:
                xhat = self.torch.empty((x.shape[0], self.plan.nfft), dtype=W.dtype, device=W.device)
            self.plan.transform_batch(x.data_ptr(), x.shape[0], x.shape[1], n0, kind, param, dt, sj, xhat.data_ptr(),
                                      W.data_ptr(), W.shape[-1], ncols)

    def classify(self, kind, param, dt, sj, ncols):
        return self.plan.classify(kind, param, dt, sj, ncols, True)

    def icwt_partial(self, W, sj, out):
        self.plan.icwt_reduce(W.data_ptr(), W.shape[1], W.shape[1], sj, 1.0, out.data_ptr())


_shard_cache: dict = {}   # (engine, mother, grid, world, partition) -> this rank's rows: the classification of a grid is host work


def _broadcast_shape(signal, rank, src, world, group, device, dist, torch):
    """The signal's shape on every rank: ONE small tensor broadcast (3 x int64), only when the callers did not pass `shape=`."""
    meta = torch.zeros(3, dtype=torch.int64, device=device)
    if rank == src:
        shp = tuple(signal.shape) if hasattr(signal, "shape") else np.shape(signal)
        meta[0] = len(shp)
        for i, v in enumerate(shp):
            meta[1 + i] = int(v)
    if world > 1:
        dist.broadcast(meta, src=src, group=group)
    m = [int(v) for v in meta.tolist()]
    return tuple(m[1:1 + m[0]])


def cwt_sharded(signal, dt, dj=1 / 12, s0=-1, J=-1, wavelet="morlet", freqs=None, *, group=None,
                precision=64, device=None, engine=None, src=0, partition="balanced", shape=None, assume_finite=False):
    """Scale-sharded `cwt`.  Call on every rank of `group`; only rank `src` needs `signal` (a sequence / NumPy array, or a
    torch tensor that already lives on this rank's device: no host copy then).

    `signal` may be 2-D (batch x n0): then every rank transforms all signals for its scales and
    `W_local` is (batch, len(rows_local), n0).

    Returns `(W_local, rows_local, sj, freqs, coi)`: `W_local` is a device tensor
    (len(rows_local) x n0, complex) holding rows `rows_local` of the full transform; `sj`, `freqs`,
    `coi` describe the full transform exactly as `pycwt.cwt` returns them (after the Paul NaN-row
    rule; read-only views of cached arrays: copy before writing).

    Collectives: ONE broadcast of the signal -- the exchange of the path -- when every rank passes `shape=` (the signal's
    shape, `(n0,)` or `(batch, n0)`); without it one more broadcast of three integers tells the other ranks.  No host
    synchronisation when `assume_finite=True`: by default the broadcast signal is checked for NaN / inf on the device and
    the answer read back (a NaN sample makes every row NaN in the reference, wavelet.py:91, :111-115; the block-wise row
    forms would confine it, so such signals take the spectrum-only entry points) -- a device-to-host round trip that costs
    about as much as a rank's share of the transform at 8 GPUs.

    `rows_local` is a contiguous run of scales of about equal estimated cost (`partition="balanced"`, one signal,
    an engine that can classify the whole grid) or `rank, rank + world, ...` (`partition="interleaved"`, batches of
    signals, an engine sized for its own share only, engines without `classify`): use the returned indices, do not
    assume either.  Without `engine=` one HipEngine per (length, precision, device) is created and kept, so that
    repeated calls re-use its classified row tables.
    """
    import torch
    import torch.distributed as dist
    from .wavelet import _geometry

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mother = _check_parameter_wavelet(wavelet)
    real_t = torch.float64 if precision == 64 else torch.float32
    cplx_t = torch.complex128 if precision == 64 else torch.complex64
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")

    if shape is None:
        shape = _broadcast_shape(signal, rank, src, world, group, device, dist, torch)
    shape = tuple(int(v) for v in shape)             # (n0,) or (batch, n0): BASELINE config 4
    n0 = shape[-1]
    if rank == src and torch.is_tensor(signal) and signal.device == device and signal.dtype == real_t and tuple(signal.shape) == shape \
            and signal.is_contiguous():
        x = signal                                   # already where the kernels read it
    else:
        x = torch.empty(shape, dtype=real_t, device=device)
        if rank == src:
            x.copy_(signal if torch.is_tensor(signal) else torch.as_tensor(np.asarray(signal), dtype=real_t))
    if world > 1:
        dist.broadcast(x, src=src, group=group)            # the one exchange of the path

    # host scalars exactly as wavelet.py:75-88 / :111-115 / :120-121 (cached per grid: they do not depend on the samples)
    N, sj, freqs, coi, _, bad = _geometry(mother, n0, dt, dj, s0, J, freqs, True)
    # a NaN / inf sample makes every row NaN in the reference (wavelet.py:91), which then keeps all rows (:111-115); the
    # block-wise rows of cwt_transform would confine the damage, so such signals go through the spectrum-only entry points
    # (as pycwt_amd.cwt does).  Every rank holds the broadcast signal and decides alike.
    finite = True if assume_finite else bool(torch.isfinite(x).all().item())
    if bad is not None and not bad.all() and finite:
        sj, freqs = sj[~bad], np.asarray(freqs)[~bad]

    kind, param = _device_id(mother)
    nbatch = shape[0] if len(shape) == 2 else 1
    if engine is None:
        cap = sj.size if (partition == "balanced" and nbatch == 1) else -(-sj.size // world) * nbatch
        key = (N, precision, device.type, device.index or 0)
        engine = _engines.get(key)
        if engine is None or engine.plan.max_rows < cap or not engine.plan.h:
            engine = _engines[key] = HipEngine(N, precision, max(1, cap), device.index or 0, device.type == "cuda")
        elif device.type == "cuda":
            engine.plan.set_stream(torch.cuda.current_stream(device.index or 0).cuda_stream)
    # balanced shards need the classification of the WHOLE grid: an engine sized for its own share cannot give it
    can_classify = hasattr(engine, "classify") and getattr(getattr(engine, "plan", None), "max_rows", sj.size) >= sj.size
    if partition == "balanced" and nbatch == 1 and world > 1 and can_classify:
        ckey = None
        try:
            ckey = (id(engine), kind, param, n0, dt, sj.tobytes(), world, rank,
                    getattr(getattr(engine, "plan", None), "tolerance", lambda: 0)())
            mine = _shard_cache.get(ckey)
        except TypeError:
            mine = None
        if mine is None:
            mine = (engine.plan.balanced_shards(kind, param, dt, sj, n0, world)[rank] if hasattr(engine, "plan")
                    else balanced_shards(engine.classify(kind, param, dt, sj, n0), world, precision, N)[rank])
            if ckey is not None:
                if len(_shard_cache) >= 64:
                    _shard_cache.clear()
                _shard_cache[ckey] = mine
    else:
        mine = shard_rows(sj.size, world, rank)
    W = torch.empty(shape[:-1] + (mine.size, n0), dtype=cplx_t, device=device)
    if hasattr(engine, "transform") and finite:
        if mine.size:                                     # one signal: the spectrum stays inside the library
            xhat = torch.empty(shape[:-1] + (N,), dtype=cplx_t, device=device) if nbatch > 1 else None
            engine.transform(x, n0, xhat, kind, param, dt, np.ascontiguousarray(sj[mine]), W, n0)
    else:
        xhat = torch.empty(shape[:-1] + (N,), dtype=cplx_t, device=device)
        engine.forward(x, n0, xhat)
        if mine.size:
            engine.rows(xhat, kind, param, dt, np.ascontiguousarray(sj[mine]), W, n0)
    return W, mine, sj, freqs, coi


def icwt_sharded(W_local, sj_local, dt, dj=1 / 12, wavelet="morlet", *, group=None, engine=None, dst=0):
    """TC98 eq. 11 over row shards: per-rank partial sums, one `reduce` to rank `dst`.

    Returns the reconstruction on rank `dst` (NumPy, dtype as `pycwt.icwt`), None elsewhere.
    """
    import torch
    import torch.distributed as dist

    mother = _check_parameter_wavelet(wavelet)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    real_t = torch.float64 if W_local.dtype == torch.complex128 else torch.float32
    part = torch.zeros(W_local.shape[1], dtype=real_t, device=W_local.device)
    if W_local.shape[0]:
        if engine is None:
            n = W_local.shape[1]
            engine = HipEngine(_next_pow2(max(n, 2)), 64 if real_t == torch.float64 else 32, W_local.shape[0],
                               W_local.device.index or 0, W_local.device.type == "cuda")
        engine.icwt_partial(W_local, np.ascontiguousarray(sj_local, dtype=np.float64), part)
    if world > 1:
        dist.reduce(part, dst=dst, op=dist.ReduceOp.SUM, group=group)
    if rank != dst:
        return None
    total = part.cpu().numpy().astype(np.float64)
    return dj * np.sqrt(dt) / (mother.cdelta * mother.psi(0)) * total


def wct_significance_sharded(al1, al2, dt, dj, s0, J, significance_level=0.95, wavelet="morlet", mc_count=300,
                             *, group=None, precision=64, device_index=None, seed=None, rng="numpy"):
    """Monte-Carlo coherence significance with the surrogate draws split over the ranks:
    every rank simulates ~mc_count/G AR(1) pairs on its GPU, the per-scale histograms (rows x 1000) are
    summed with ONE all-reduce, and every rank evaluates the same percentiles.  Returns the array of
    `pycwt.wct_significance` (no disk cache here).  `rng="device"`: the surrogates are made on the GPUs
    (`cwt_random_normal`): rank r takes a CONTIGUOUS block of draws, [first_r, first_r + count_r), of ONE Philox sequence named
    by `seed`, so the result does not depend on the number of ranks.  `seed=None`: as `wct_significance` does, a seed is drawn
    from NumPy's global generator -- on rank 0, and broadcast (one more small collective), so that every rank names the same
    sequence."""
    import torch
    import torch.distributed as dist
    from . import wavelet as _w

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mother = _check_parameter_wavelet(wavelet)
    if device_index is None:
        device_index = torch.cuda.current_device() if torch.cuda.is_available() else 0
    if seed is not None:
        np.random.seed(seed + rank)                       # independent surrogates per rank
    N, sj, outside, rows_with_data, maxscale = _w._mc_setup(mother, dt, dj, s0, J)
    mine = len(range(rank, mc_count, world))
    if rng == "device":
        if seed is None:
            box = [int(np.random.randint(0, 2 ** 31 - 1)) * (2 ** 31) + int(np.random.randint(0, 2 ** 31 - 1))]
            if world > 1:
                dist.broadcast_object_list(box, src=0, group=group)
            seed = box[0]
        # contiguous blocks of draws per rank: [first, first + mine) of the one sequence
        first = sum(len(range(r, mc_count, world)) for r in range(rank))
        hist = _w._mc_histogram(mine, al1, al2, dt, dj, sj, N, outside, maxscale, mother, precision, device_index,
                                rng="device", seed=int(seed), first_draw=first)
    else:
        hist = _w._mc_histogram(mine, al1, al2, dt, dj, sj, N, outside, maxscale, mother, precision, device_index)
    if world > 1:
        backend = dist.get_backend(group)
        t = torch.from_numpy(hist)
        if backend == "nccl":
            t = t.to(torch.device("cuda", device_index))
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        hist = t.cpu().numpy()
    return _w._mc_percentiles(hist, rows_with_data, maxscale, significance_level)
