"""The C-ABI library builds for gfx950, loads on a box without a GPU, and exports every symbol
include/cwt_hip.h declares (no compute calls here)."""
import os
import re

import pytest

from conftest import ROOT
from pycwt_amd import _build, _hip


def header_functions():
    text = open(os.path.join(ROOT, "include", "cwt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cwt_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert header_functions() == sorted(name for name, _, _ in _hip.SYMBOLS)


def test_product_library_builds_loads_and_exports_all_symbols():
    path = _build.build()
    assert path.endswith("pycwt_amd/libcwt_hip.so") and os.path.exists(path)
    lib = _hip.Library(path)            # getattr on every symbol; AttributeError if one is missing
    assert lib.backend() == "hip-gfx950"
    assert lib.cwt_last_error() is not None


def test_library_carries_the_identity_of_its_sources(tmp_path, monkeypatch):
    """`cwt_build_id()` = `_build.source_id()` of the tree the library was built from; a library built from other sources is
    detected from the file alone (no dlopen), rebuilt by `_build.ensure()` / `_hip.load()`, or refused where it cannot be."""
    path = _build.build()
    assert _build.up_to_date() and _build.library_id(path) == _build.source_id() == _hip.Library(path).build_id()
    assert _build.source_id(["-DX=1"]) != _build.source_id()             # a -D variant is another binary
    # a stale copy: the same file with another id inside
    blob = open(path, "rb").read()
    i = blob.find(_build.ID_MARKER) + len(_build.ID_MARKER)
    stale = tmp_path / "libcwt_hip.so"
    stale.write_bytes(blob[:i] + b"0123456789abcdef" + blob[i + 16:])
    assert _build.library_id(str(stale)) == "0123456789abcdef"
    monkeypatch.setattr(_build, "OUT", str(stale))
    monkeypatch.setattr(_hip, "DEFAULT_LIBRARY", str(stale))
    assert not _build.up_to_date()

    def no_compiler(*a, **k):
        raise RuntimeError("hipcc not found")
    monkeypatch.setattr(_build, "build", no_compiler)
    with pytest.raises(ImportError, match="built from other sources"):
        _hip._check_provenance()
    with pytest.raises(RuntimeError, match="built from other sources"):
        _build.ensure(0)
    monkeypatch.setenv("PYCWT_AMD_ALLOW_STALE", "1")
    _hip._check_provenance()                                              # the override: used as it is
    assert _build.ensure(0) == str(stale)


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(ImportError, match="no CPU fallback"):
        _hip.Library(str(tmp_path / "libcwt_hip.so"))


def test_argument_validation_without_gpu(emu_library):
    import ctypes as C
    lib = emu_library
    h = C.c_void_p()
    assert lib.cwt_plan_create(C.byref(h), 0, 1000, 64, 4) < 0          # not a power of two
    assert b"power of two" in lib.cwt_last_error()
    assert lib.cwt_plan_create(C.byref(h), 0, 1024, 16, 4) < 0          # bad precision
    assert lib.cwt_plan_create(C.byref(h), 0, 1024, 64, 0) < 0          # bad max_rows
    assert lib.cwt_plan_create(C.byref(h), 0, 1024, 64, 4) == 0
    assert lib.cwt_plan_set_option(h, b"nonsense", 1) < 0
    assert lib.cwt_plan_set_option(h, b"lmax", 100) < 0
    assert lib.cwt_forward_fft(h, None, 10, None) < 0
    assert lib.cwt_plan_destroy(h) == 0


def test_torch_runtime_is_loaded_before_the_library():
    """PyTorch-ROCm wheels bundle their own HIP runtime under the system's SONAMEs; the one loaded first serves the
    process, and with the system's first torch sees no GPU.  `_hip.load()` therefore imports torch (if installed)
    before it opens libcwt_hip.so; PYCWT_AMD_NO_TORCH_PRELOAD=1 skips that."""
    import subprocess
    import sys
    pytest.importorskip("torch")          # nothing to order on a machine without torch
    code = ("import sys; sys.path.insert(0, %r); from pycwt_amd import _hip; assert 'torch' not in sys.modules; "
            "_hip._one_hip_runtime(); print('torch' in sys.modules)" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout.strip()
    assert out == "True"
    env = dict(os.environ, PYCWT_AMD_NO_TORCH_PRELOAD="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True, env=env).stdout.strip()
    assert out == "False"
