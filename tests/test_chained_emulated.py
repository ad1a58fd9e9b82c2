"""cwt_plan_set_input_stream: the chained schedule of cwt_transform (preparation not behind the plan's stream) -- the LOGIC on
the CPU emulation, where streams execute at launch: which calls take it, that they give the bits of the ordinary schedule, that
other entry points in between are safe.  The hazards between streams are what tests/test_gpu_parity.py::test_chained_* check."""
import numpy as np
import pytest

from conftest import row_errors
from oracle import cwt_oracle as orc
from pycwt_amd import _hip
from test_kernels_emulated import grid

N = 1 << 16


def buffers(plan, lib, x, rows):
    es = 8 if plan.precision == 64 else 4
    xd = _hip.DeviceBuffer(x.size * es, lib=lib)
    xh = _hip.DeviceBuffer(plan.nfft * 2 * es, lib=lib)
    W = _hip.DeviceBuffer(rows * x.size * 2 * es, lib=lib)
    xd.upload(plan, np.ascontiguousarray(x, dtype=plan.real))
    return xd, xh, W


@pytest.mark.parametrize("kind,param,prec", [(orc.MORLET, 6, 64), (orc.DOG, 2, 32)])
def test_chained_calls_give_the_bits_of_ordinary_calls(emu_library, monkeypatch, kind, param, prec):
    monkeypatch.delenv("CWT_TOLERANCE", raising=False)
    m = orc.Mother(kind, param)
    n0 = N - 37
    sj = grid(n0, 1.0, m, 40)
    opts = {"ols_min_logn": 15, "poly_min_logn": 15, "tolerance": 1e-9 if prec == 64 else 3e-5}
    plan = _hip.Plan(N, prec, max_rows=len(sj), lib=emu_library, options=opts)
    plan.set_input_stream(0, True)
    ref_plan = _hip.Plan(N, prec, max_rows=len(sj), lib=emu_library, options=opts)
    signals = [np.random.default_rng(s).standard_normal(n0) for s in range(4)]
    bufs = [buffers(plan, emu_library, x, len(sj)) for x in signals]
    for i, (xd, xh, W) in enumerate(bufs):
        plan.transform(xd.ptr, n0, kind, param, 1.0, sj, xh.ptr if i % 2 == 0 else None, W.ptr, n0, n0)
        if i == 1:                                         # another entry point in between: the next chained call waits for it
            tmp = _hip.DeviceBuffer(n0 * (8 if prec == 64 else 4), lib=emu_library)
            plan.icwt_reduce(W.ptr, n0, n0, sj, 1.0, tmp.ptr)
            tmp.free()
    assert plan.chained_calls() == len(signals), plan.row_classes()
    for i, (x, (xd, xh, W)) in enumerate(zip(signals, bufs)):
        got = W.download(plan, (len(sj), n0), plan.cplx)
        rx, rh, rW = buffers(ref_plan, emu_library, x, len(sj))
        ref_plan.transform(rx.ptr, n0, kind, param, 1.0, sj, rh.ptr, rW.ptr, n0, n0)
        assert np.array_equal(got, rW.download(ref_plan, (len(sj), n0), ref_plan.cplx))
        if i % 2 == 0:
            assert np.array_equal(xh.download(plan, (N,), plan.cplx), rh.download(ref_plan, (N,), ref_plan.cplx))
        for b in (rx, rh, rW):
            b.free()
    ref = orc.cwt_rows(signals[-1], 1.0, sj, m, N=N)[:, :n0]
    assert row_errors(got, ref)[0].max() < (1e-8 if prec == 64 else 1e-4)
    assert ref_plan.chained_calls() == 0
    # a grid with other row forms ignores the setting; switching it off goes back to the ordinary schedule
    plan.set_option("poly", 0)
    xd, xh, W = bufs[0]
    plan.transform(xd.ptr, n0, kind, param, 1.0, sj, xh.ptr, W.ptr, n0, n0)
    assert plan.chained_calls() == len(signals)
    plan.set_option("poly", 1)
    plan.set_input_stream(0, False)
    plan.transform(xd.ptr, n0, kind, param, 1.0, sj, xh.ptr, W.ptr, n0, n0)
    assert plan.chained_calls() == len(signals)
    for tr in bufs:
        for b in tr:
            b.free()
    plan.close()
    ref_plan.close()
