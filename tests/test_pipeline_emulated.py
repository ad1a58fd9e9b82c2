"""Pipelined mode of cwt_transform (option "pipeline", cwt_plan_join): the LOGIC of the lanes -- which scratch a call uses,
which calls take the path, that every other entry point joins -- on the CPU emulation, where streams execute at launch (the
overlap itself, i.e. the hazards between streams, is what tests/test_gpu_parity.py::test_pipelined_* check on the GPU)."""
import numpy as np
import pytest

from conftest import row_errors
from oracle import cwt_oracle as orc
from pycwt_amd import _hip
from test_kernels_emulated import grid

N = 1 << 16


def device_transform(plan, lib, x, kind, param, sj, want_xhat=True):
    es = 8 if plan.precision == 64 else 4
    xd = _hip.DeviceBuffer(x.size * es, lib=lib)
    xh = _hip.DeviceBuffer(plan.nfft * 2 * es, lib=lib)
    W = _hip.DeviceBuffer(len(sj) * x.size * 2 * es, lib=lib)
    xd.upload(plan, np.ascontiguousarray(x, dtype=plan.real))
    return xd, xh, W


@pytest.mark.parametrize("kind,param,prec", [(orc.MORLET, 6, 64), (orc.DOG, 2, 32), (orc.PAUL, 4, 64)])
def test_pipelined_calls_give_the_rows_of_ordinary_calls(emu_library, monkeypatch, kind, param, prec):
    monkeypatch.delenv("CWT_TOLERANCE", raising=False)
    m = orc.Mother(kind, param)
    n0 = N - 37
    sj = grid(n0, 1.0, m, 40)
    opts = {"ols_min_logn": 15, "poly_min_logn": 15, "tolerance": 1e-9 if prec == 64 else 3e-5}
    plan = _hip.Plan(N, prec, max_rows=len(sj), lib=emu_library, options=dict(opts, pipeline=1))
    ref_plan = _hip.Plan(N, prec, max_rows=len(sj), lib=emu_library, options=opts)
    es = 8 if prec == 64 else 4
    signals = [np.random.default_rng(s).standard_normal(n0) for s in range(5)]
    bufs = [device_transform(plan, emu_library, x, kind, param, sj) for x in signals]
    for xd, xh, W in bufs:
        plan.transform(xd.ptr, n0, kind, param, 1.0, sj, xh.ptr, W.ptr, n0, n0)
    labels = plan.row_classes()
    if all(c.split("/")[0] in ("poly", "ols", "aols") for c in labels):
        assert plan.pipelined_calls() == len(signals)          # every call took the pipelined path
    plan.join()
    for x, (xd, xh, W) in zip(signals, bufs):
        got = W.download(plan, (len(sj), n0), plan.cplx)
        spec = xh.download(plan, (N,), plan.cplx)
        rx, rh, rW = device_transform(ref_plan, emu_library, x, kind, param, sj)
        ref_plan.transform(rx.ptr, n0, kind, param, 1.0, sj, rh.ptr, rW.ptr, n0, n0)
        want = rW.download(ref_plan, (len(sj), n0), ref_plan.cplx)
        assert np.array_equal(got, want)                        # same kernels, same inputs: the same bits
        assert np.array_equal(spec, rh.download(ref_plan, (N,), ref_plan.cplx))
        for b in (rx, rh, rW):
            b.free()
    keep = ~orc.dropped_rows(sj, 1.0, m)
    ref = orc.cwt_rows(signals[-1], 1.0, sj, m, N=N)[:, :n0]
    assert row_errors(got[keep], ref[keep])[0].max() < (1e-8 if prec == 64 else 1e-4)
    for tr in bufs:
        for b in tr:
            b.free()
    plan.close()
    ref_plan.close()


def test_other_entry_points_join_and_other_tables_leave_the_path(emu_library, monkeypatch):
    monkeypatch.delenv("CWT_TOLERANCE", raising=False)
    m = orc.Mother(orc.MORLET, 6)
    n0 = N
    sj = grid(n0, 1.0, m, 40)
    plan = _hip.Plan(N, 64, max_rows=64, lib=emu_library,
                     options={"ols_min_logn": 15, "poly_min_logn": 15, "tolerance": 1e-9, "pipeline": 1})
    x = np.random.default_rng(5).standard_normal(n0)
    xd, xh, W = device_transform(plan, emu_library, x, orc.MORLET, 6, sj)
    plan.transform(xd.ptr, n0, orc.MORLET, 6, 1.0, sj, xh.ptr, W.ptr, n0, n0)
    assert plan.pipelined_calls() == 1
    # icwt_reduce is "another entry point": it joins and sees the complete W
    out = _hip.DeviceBuffer(n0 * 8, lib=emu_library)
    plan.icwt_reduce(W.ptr, n0, n0, sj, 1.0, out.ptr)
    got = out.download(plan, (n0,), np.float64)
    Wh = W.download(plan, (len(sj), n0), np.complex128)
    np.testing.assert_allclose(got, (Wh.real / np.sqrt(sj)[:, None]).sum(axis=0), rtol=1e-10, atol=1e-12)
    # a grid with single-workgroup / two-pass rows does not take the path, and still computes
    plan.set_option("poly", 0)
    plan.set_option("ols", 0)
    plan.set_option("aols", 0)
    plan.transform(xd.ptr, n0, orc.MORLET, 6, 1.0, sj, xh.ptr, W.ptr, n0, n0)
    assert plan.pipelined_calls() == 1
    ref = orc.cwt_rows(x, 1.0, sj, m, N=N)
    assert row_errors(W.download(plan, (len(sj), n0), np.complex128), ref)[0].max() < 1e-8
    # the input-stream setter is accepted and switched off again
    plan.set_input_stream(0, True)
    plan.set_input_stream(0, False)
    for b in (xd, xh, W, out):
        b.free()
    plan.close()
