"""Surrogate series made on the device (cwt_random_normal, cwt_ar1_filter; pycwt/wavelet.py:609-613, helpers.py:146-173), on
the CPU emulation: the generator against a NumPy restatement of Philox4x32-10 + Box-Muller (itself checked on the Random123
known-answer vectors), the statistics of the deviates, the AR(1) filter against scipy.signal.lfilter."""
import numpy as np
import pytest
from scipy.signal import lfilter

from pycwt_amd import _hip

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def philox4x32_10(counter, key):
    """Vectorised Philox4x32-10: counter (..., 4) uint32, key (2,) ints -> (..., 4) uint32."""
    c = [counter[..., i].astype(np.uint64) for i in range(4)]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    mask = np.uint64(0xFFFFFFFF)
    for r in range(10):
        p0, p1 = np.uint64(M0) * c[0], np.uint64(M1) * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        if r < 9:
            k0, k1 = (k0 + np.uint64(W0)) & mask, (k1 + np.uint64(W1)) & mask
    return np.stack(c, axis=-1).astype(np.uint32)


def test_numpy_philox_on_the_random123_known_answers():
    """Random123 kat_vectors, philox4x32 10 rounds."""
    z = philox4x32_10(np.zeros((1, 4), np.uint32), (0, 0))[0]
    assert [hex(v) for v in z] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    f = philox4x32_10(np.full((1, 4), 0xFFFFFFFF, np.uint32), (0xFFFFFFFF, 0xFFFFFFFF))[0]
    assert [hex(v) for v in f] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    p = philox4x32_10(np.array([[0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344]], np.uint32), (0xA4093822, 0x299F31D0))[0]
    assert [hex(v) for v in p] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def expected_normals(seed, offset, n):
    t = np.arange((n + 1) // 2, dtype=np.uint64)
    ctr = np.stack([t & np.uint64(0xFFFFFFFF), t >> np.uint64(32), np.full_like(t, offset & 0xFFFFFFFF), np.full_like(t, offset >> 32)],
                   axis=-1).astype(np.uint32)
    r = philox4x32_10(ctr, (seed & 0xFFFFFFFF, seed >> 32)).astype(np.uint64)
    u1 = (((r[:, 0] << np.uint64(21)) ^ (r[:, 1] >> np.uint64(11))).astype(np.float64) + 1.0) / 9007199254740992.0
    u2 = ((r[:, 2] << np.uint64(21)) ^ (r[:, 3] >> np.uint64(11))).astype(np.float64) / 9007199254740992.0
    rad, ang = np.sqrt(-2.0 * np.log(u1)), 2 * np.pi * u2
    return np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=-1).reshape(-1)[:n]


@pytest.mark.parametrize("prec", [64, 32])
def test_random_normal_is_philox_box_muller_and_reproducible(emu_library, prec):
    plan = _hip.Plan(1 << 12, prec, max_rows=4, lib=emu_library)
    real = np.float64 if prec == 64 else np.float32
    for seed, offset, n in ((0, 0, 9), (0x1234567890ABCDEF, 7, 4097), (42, (1 << 40) + 3, 1000)):
        buf = _hip.DeviceBuffer(n * real().itemsize, lib=emu_library)
        plan.random_normal(seed, offset, n, 1.5, buf.ptr)
        got = buf.download(plan, (n,), real)
        want = 1.5 * expected_normals(seed, offset, n)
        np.testing.assert_allclose(got, want.astype(real), rtol=1e-12 if prec == 64 else 2e-6, atol=1e-300)
        plan.random_normal(seed, offset, n, 1.5, buf.ptr)
        assert np.array_equal(got, buf.download(plan, (n,), real))             # the same call, the same bits
        buf.free()
    plan.close()


def test_random_normal_statistics(emu_library):
    plan = _hip.Plan(1 << 12, 64, max_rows=4, lib=emu_library)
    n = 1 << 18
    a, b = _hip.DeviceBuffer(n * 8, lib=emu_library), _hip.DeviceBuffer(n * 8, lib=emu_library)
    plan.random_normal(2024, 0, n, 1.0, a.ptr)
    plan.random_normal(2024, 1, n, 1.0, b.ptr)
    x, y = a.download(plan, (n,), np.float64), b.download(plan, (n,), np.float64)
    se = 1 / np.sqrt(n)
    for z in (x, y):
        assert abs(z.mean()) < 4 * se and abs(z.var() - 1) < 4 * np.sqrt(2) * se
        assert abs(((z - z.mean()) ** 4).mean() / z.var() ** 2 - 3) < 4 * np.sqrt(24) * se       # kurtosis of a normal
        assert abs(np.corrcoef(z[:-1], z[1:])[0, 1]) < 4 * se                                      # neighbours (cos / sin of one draw)
        assert abs(np.corrcoef(z[:-2], z[2:])[0, 1]) < 4 * se
    assert abs(np.corrcoef(x, y)[0, 1]) < 4 * se                                                   # two offsets: two series
    # the tails are there: a Box-Muller on 24-bit uniforms would stop at 5.9 sigma, and a broken one much earlier
    assert 4.0 < np.abs(x).max() < 6.5
    from scipy.stats import kstest
    assert kstest(x[:50000], "norm").pvalue > 1e-3
    a.free(); b.free()
    plan.close()


@pytest.mark.parametrize("prec,g,n,tau", [(64, 0.7, 5000, 6), (64, 0.99, 70001, 200), (64, -0.5, 300, 3), (32, 0.9, 20000, 19), (64, 0.9999, 3000, 0)])
def test_ar1_filter_against_lfilter(emu_library, prec, g, n, tau):
    """y = lfilter([1, 0], [1, -g], e, axis=0)[tau:] -- helpers.py:170 as it is meant.  g = 0.99 at n = 70001: segments start
    from a truncated history (g^warm <= 1e-17); g = 0.9999: every segment runs from e[0]."""
    real = np.float64 if prec == 64 else np.float32
    plan = _hip.Plan(1 << 12, prec, max_rows=4, lib=emu_library)
    e = np.random.default_rng(5).standard_normal(n + tau).astype(real)
    ed, yd = _hip.DeviceBuffer(e.nbytes, lib=emu_library), _hip.DeviceBuffer(n * e.itemsize, lib=emu_library)
    ed.upload(plan, e)
    plan.ar1_filter(ed.ptr, tau, n, g, yd.ptr)
    got = yd.download(plan, (n,), real)
    want = lfilter([1, 0], [1, -g], e.astype(np.float64))[tau:]
    assert np.abs(got - want).max() <= (1e-12 if prec == 64 else 2e-6) * np.abs(want).max()
    with pytest.raises(_hip.HipError):
        plan.ar1_filter(ed.ptr, tau, n, 1.0, yd.ptr)
    ed.free(); yd.free()
    plan.close()
