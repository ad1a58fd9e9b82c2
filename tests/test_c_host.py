"""A host written in plain C (tests/c_host/cwt_host.c: C99, gcc, only include/cwt_hip.h) drives the library through the
C ABI and checks W against the closed form of the transform of a cosine: the boundary works for a
compiled-language caller, without Python, torch or HIP headers on the caller's side.

CPU: linked against the CPU emulation of the kernels (tests/emu); GPU: against pycwt_amd/libcwt_hip.so."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_host", "cwt_host.c")


def build_host(tmp_path, libfile):
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    libdir, name = os.path.split(libfile)
    exe = str(tmp_path / "cwt_host")
    # -l:<file> links the shared object by its file name (libcwt_hip.so / libcwt_emu.so alike)
    cmd = [gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
           "-L", libdir, f"-l:{name}", "-lm", f"-Wl,-rpath,{libdir}"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def run_host(exe, logn, prec):
    env = dict(os.environ)
    env.pop("CWT_TOLERANCE", None)            # the C host runs the library's default accuracy targets
    r = subprocess.run([exe, str(logn), str(prec)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout
    return r.stdout


@pytest.mark.parametrize("logn,prec", [(10, 64), (14, 64), (16, 32), (18, 64)])
def test_c_host_on_the_emulated_kernels(emu_library, tmp_path, logn, prec):
    out = run_host(build_host(tmp_path, emu_library.path), logn, prec)
    assert "cpu-emulation" in out


@pytest.mark.gpu
@pytest.mark.parametrize("logn,prec", [(12, 64), (16, 32), (20, 64), (20, 32)])
def test_c_host_on_the_gpu(hip_library, tmp_path, logn, prec):
    out = run_host(build_host(tmp_path, hip_library.path), logn, prec)
    assert "hip-gfx950" in out
