"""Host wall time of the drop-in call at N = 2^20 x 77 rows (1.29 GB of W crossing PCIe into a fresh NumPy array)
for several CWT_COPY_THREADS settings (one process each: the copier reads the variable once per plan).
python tests/perf/d2h_bench.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r"""
import sys, time, numpy as np
sys.path.insert(0, %r)
import pycwt_amd
x = np.random.default_rng(0).standard_normal(1 << 20)
f = lambda: pycwt_amd.cwt(x, 0.25, 0.25, wavelet="morlet")
f(); ts = []
for _ in range(5):
    t = time.perf_counter(); W = f()[0]; ts.append(time.perf_counter() - t)
print("%%6.1f ms  (%%.1f GB/s of W)" %% (min(ts) * 1e3, W.nbytes / min(ts) / 1e9))
""" % ROOT
for t in sys.argv[1:] or ["1", "8", "16", "32", "64", "128"]:
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CWT_COPY_THREADS=t), capture_output=True, text=True)
    print(f"CWT_COPY_THREADS={t:4s}", out.stdout.strip() or out.stderr[-400:])
