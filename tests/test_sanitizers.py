"""SURVEY.md 5 / VERDICT r05 7: the host logic (plan_host.cpp: the index arithmetic that sizes every launch and scratch buffer)
and the kernels' own indexing on the CPU stand-in, under -fsanitize=address,undefined.

The instrumented build of the unmodified sources takes ~6 minutes and the two kernel suites ~6 more, so this runs when asked
for: CWT_RUN_SANITIZERS=1 python -m pytest tests/test_sanitizers.py   (once per round; the record of the last run is
profiles/r06_sanitizer.txt)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.slow
@pytest.mark.skipif(not os.environ.get("CWT_RUN_SANITIZERS"), reason="set CWT_RUN_SANITIZERS=1 (12 minutes: instrumented build + two kernel suites)")
def test_kernel_suites_under_address_and_undefined_behaviour_sanitizers():
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    build_emu.build(asan=True)
    env = build_emu.sanitizer_env()
    env["PYTHONPATH"] = ROOT
    env.pop("CWT_RUN_SANITIZERS", None)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_kernels_randomized.py", "tests/test_new_forms_emulated.py",
                        "tests/test_c_host.py", "-x", "-q", "-n", "6", "-m", "not gpu"], cwd=ROOT, env=env, capture_output=True, text=True)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
