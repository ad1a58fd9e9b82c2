"""pytest configuration: registers the ``gpu`` marker and shared fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The engine's default accuracy target is round-off (every truncation of the fast forms below the arithmetic's rounding),
# which is what most of the suite checks the kernels' ARITHMETIC at.  The shim's automatic mode (pycwt_amd.set_tolerance
# "auto": 1e-9 / 3e-5 relative to every row's peak, tightened by the dynamic range of the signal's spectrum) and the
# targets bench.py times are exercised by the tests that say so (test_tolerance_*, test_shim_*, the "bench" rows of
# test_every_row_of_the_bench_workloads_against_the_oracle); tests of the shim's arithmetic switch it to round-off.


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: takes more than ~20 s on CPU")


@pytest.fixture(autouse=True)
def shim_at_round_off():
    """pycwt_amd's default accuracy mode is "auto" (1e-9 / 3e-5 relative to every row's peak).  The suite compares the shim
    with fixtures at 1e-12: it runs the shim at the engine's round-off default; the tests of the automatic mode switch it
    on themselves (pycwt_amd.set_tolerance("auto"))."""
    from pycwt_amd import wavelet
    keep = wavelet._tolerance
    wavelet._tolerance = None
    yield
    wavelet._tolerance = keep


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def row_errors(W, Wref):
    """Parity metric of SURVEY.md 8c: per-row max|dW| / max|Wref| and global rel-L2."""
    W = np.asarray(W)
    Wref = np.asarray(Wref)
    num = np.abs(W - Wref).max(axis=-1)
    den = np.abs(Wref).max(axis=-1)
    per_row = num / np.where(den == 0, 1, den)
    l2 = np.linalg.norm((W - Wref).ravel()) / max(np.linalg.norm(Wref.ravel()), 1e-300)
    return per_row, l2


@pytest.fixture(scope="session")
def emu_library():
    """libcwt_emu.so: the kernel sources compiled against the CPU stand-in for HIP (tests/emu)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    from pycwt_amd import _hip
    if os.environ.get("CWT_EMU_LIBRARY"):         # the sanitizer build, in a process started with build_emu.sanitizer_env()
        return _hip.Library(os.environ["CWT_EMU_LIBRARY"])
    return _hip.Library(build_emu.build())


@pytest.fixture()
def emulated(emu_library, monkeypatch):
    """Route the Python shim to the emulated library for one test (test-only monkeypatch; the
    product resolves pycwt_amd/libcwt_hip.so and nothing else)."""
    from pycwt_amd import _hip, wavelet
    monkeypatch.setattr(_hip, "_default", emu_library)
    monkeypatch.setattr(wavelet, "_plans", {})
    yield emu_library
    for p in wavelet._plans.values():
        p.close()


@pytest.fixture(scope="session")
def hip_library():
    """The product library on a real GPU (gpu-marked tests only)."""
    from pycwt_amd import _build, _hip
    _build.build()
    lib = _hip.load()
    assert lib.backend() == "hip-gfx950"
    if lib.device_count() < 1:
        pytest.fail("gpu test selected but no GPU is visible")
    return lib
