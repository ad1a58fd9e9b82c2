"""Host-visible latency of the drop-in call pycwt_amd.cwt() (NumPy in, NumPy out, PCIe included) next to the CPU
oracle, for short and medium series.  Times the call that produces the result; releasing the previous result
(munmap of up to 1.3 GB of W, 60-80 ms at 2^20 x 77 -- the caller pays that for the reference's W too) happens outside
the clock and is printed separately.   python tests/perf/latency_bench.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pycwt_amd
from oracle import cwt_oracle as orc


def best(f, reps):
    r = f()
    ts, frees = [], []
    for _ in range(reps):
        t = time.perf_counter(); del r; frees.append(time.perf_counter() - t)
        t = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t)
    return min(ts), min(frees)


for n0, dj in ((504, 1 / 12), (4096, 1 / 12), (65536, 1 / 12), (1 << 20, 0.25)):
    x = np.random.default_rng(0).standard_normal(n0)
    g, gf = best(lambda: pycwt_amd.cwt(x, 0.25, dj, wavelet="morlet"), 200 if n0 <= 4096 else 5)   # short calls: enough of them for the clock
    W = pycwt_amd.cwt(x, 0.25, dj, wavelet="morlet")[0]
    rows, nbytes = W.shape[0], W.nbytes
    del W
    c, _ = best(lambda: orc.cwt(x, 0.25, dj, wavelet="morlet"), 2 if n0 > 100000 else 5)
    print(f"n0={n0:8d} rows={rows:4d}  pycwt_amd.cwt {g*1e3:9.2f} ms ({nbytes / g / 1e9:5.1f} GB/s of W; freeing it "
          f"{gf*1e3:6.2f} ms)   oracle (1 core) {c*1e3:10.2f} ms   x{c/g:6.1f}")
