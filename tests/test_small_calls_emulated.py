"""Short transforms (one workgroup per row: the reference's canonical 504-point call, sample/simple_sample.py:58): the
kernels read the signal from / write W into page-locked host memory, no copy operations (cwt_execute_host,
cwt_host_malloc), and the result arrays of the shim come from a pool of such buffers.
CPU emulation of the real kernels; the GPU repeat is tests/test_gpu_parity.py::test_short_calls_on_gpu."""
import ctypes as C
import gc

import numpy as np
import pytest

import pycwt_amd
from conftest import load_golden, row_errors
from oracle import cwt_oracle as orc
from pycwt_amd import _hip
from test_kernels_emulated import grid


def call(lib, prec, kind, param, x, sj, N, **opts):
    plan = _hip.Plan(N, prec, max_rows=len(sj), lib=lib, options=opts)
    W, xhat = plan.execute_host(x, kind, param, 0.25, sj)
    classes = plan.row_classes()
    plan.close()
    return W, xhat, classes


@pytest.mark.parametrize("prec,tol", [(64, 2e-14), (32, 2e-5)])
@pytest.mark.parametrize("kind,param", [(orc.MORLET, 6), (orc.PAUL, 4), (orc.DOG, 2), (orc.DOG, 5)])
@pytest.mark.parametrize("n0", [16, 100, 504, 1000, 4096])
def test_direct_and_staged_calls_agree_with_each_other_and_the_oracle(emu_library, prec, tol, kind, param, n0):
    N = 1 << int(np.ceil(np.log2(n0)))
    x = np.random.default_rng(n0).standard_normal(n0)
    m = orc.Mother(kind, param)
    sj = grid(n0, 0.25, m, 23)
    fused = call(emu_library, prec, kind, param, x, sj, N)
    staged = call(emu_library, prec, kind, param, x, sj, N, host_direct=0)
    assert set(fused[2]) == {"single_wg"}
    np.testing.assert_array_equal(fused[0], staged[0])              # the same kernels on the same values
    np.testing.assert_array_equal(fused[1], staged[1])
    ref = orc.cwt_rows(x, 0.25, sj, m, N=N)[:, :n0]
    per_row, l2 = row_errors(fused[0], ref)
    assert per_row.max() < tol and l2 < tol
    xref = np.fft.fft(x, N)
    assert np.abs(fused[1] - xref).max() <= (1e-14 if prec == 64 else 1e-5) * np.abs(xref).max()


def test_result_in_a_page_locked_buffer_of_the_library(emu_library):
    """cwt_execute_host writes a W_host from cwt_host_malloc in place; one that is not goes through the staging buffer.
    Both give the same bytes."""
    lib = emu_library
    g = load_golden("nino3_simple")
    x, sj = np.ascontiguousarray(g["x"]), np.ascontiguousarray(g["sj"])
    plan = _hip.Plan(512, 64, max_rows=len(sj), lib=lib)
    nbytes = sj.size * x.size * 16
    p = C.c_void_p()
    lib.check(lib.cwt_host_malloc(C.byref(p), nbytes))
    try:
        Wp = np.frombuffer((C.c_char * nbytes).from_address(p.value), dtype=np.complex128).reshape(sj.size, x.size)
        Wp[...] = 0
        xh = np.empty(512, dtype=np.complex128)
        lib.check(lib.cwt_execute_host(plan.h, x.ctypes.data, x.size, orc.MORLET, 6.0, 0.25,
                                       sj.ctypes.data_as(C.POINTER(C.c_double)), sj.size, p.value, xh.ctypes.data))
        W = np.empty_like(Wp)
        lib.check(lib.cwt_execute_host(plan.h, x.ctypes.data, x.size, orc.MORLET, 6.0, 0.25,
                                       sj.ctypes.data_as(C.POINTER(C.c_double)), sj.size, W.ctypes.data, None))
        np.testing.assert_array_equal(W, Wp)
        per_row, _ = row_errors(W, g["W"])
        assert per_row.max() < 1e-13
        del Wp
    finally:
        lib.check(lib.cwt_host_free(p))
    assert lib.cwt_host_free(p) != 0 and b"cwt_host_malloc" in lib.cwt_last_error()      # not (any more) one of ours
    assert lib.cwt_host_free(None) == 0
    assert lib.cwt_host_malloc(None, 16) != 0 and lib.cwt_host_malloc(C.byref(p), 0) != 0
    plan.close()


def test_pool_of_result_arrays(emu_library):
    pool = _hip.PinnedPool(emu_library)
    a = pool.empty((97, 504), np.complex128)
    assert a.shape == (97, 504) and a.dtype == np.complex128 and a.flags.writeable and a.flags.c_contiguous
    addr = a.ctypes.data
    a[...] = 1 + 2j
    view = a[3:5, :7]
    del a
    gc.collect()
    b = pool.empty((97, 504), np.complex128)           # the first buffer is still held by the view
    assert b.ctypes.data != addr
    assert (view == 1 + 2j).all()
    del view
    gc.collect()
    c = pool.empty((90, 504), np.complex128)           # same size class: the released buffer comes back
    assert c.ctypes.data == addr
    assert pool.empty((1 << 20, 2), np.complex128) is None          # above the per-array limit: pageable memory
    assert pool.empty((0, 5), np.complex128) is None
    pool.LIMIT = pool.total                              # nothing more may be pinned
    assert pool.empty((97, 504), np.complex128) is None
    del b
    gc.collect()
    assert pool.empty((97, 504), np.complex128) is not None         # ... but released buffers are still handed out


def test_shim_results_are_independent_arrays(emulated):
    """Results of consecutive calls live in different buffers for as long as the caller holds them."""
    g = load_golden("nino3_simple")
    outs = [pycwt_amd.cwt(g["x"] * k, 0.25, 1 / 12, 0.5, 84, "morlet")[0] for k in (1.0, 2.0, 3.0)]
    for k, W in zip((1.0, 2.0, 3.0), outs):
        per_row, _ = row_errors(W, k * g["W"])
        assert per_row.max() < 1e-12
    assert len({W.ctypes.data for W in outs}) == 3
    W = outs[0]
    W *= 2                                               # writable, like the reference's
    assert W.dtype == np.complex128 and W.shape == g["W"].shape
