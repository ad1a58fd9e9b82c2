import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from pycwt_amd import _hip
for logn in (16, 20, 23):
    n = 1 << logn
    z = (np.random.default_rng(1).standard_normal(n) + 1j * np.random.default_rng(2).standard_normal(n))
    plan = _hip.Plan(n, 64, max_rows=4)
    buf = _hip.DeviceBuffer(z.nbytes); buf.upload(plan, z)
    plan.spectrum_range(buf.ptr, n)
    t = time.perf_counter()
    for _ in range(50): r = plan.spectrum_range(buf.ptr, n)
    dt = (time.perf_counter() - t) / 50
    a = np.abs(z)
    print(f"N=2^{logn}: {dt*1e6:.0f} us per call incl. sync; max ok {abs(r[0]-a.max())<1e-12*a.max()}, rms ok {abs(r[1]-np.sqrt((a**2).mean()))<1e-12}")
    buf.free(); plan.close()
