"""Repeated fits of one shape with pauses between them, for a rocprofv3 --kernel-trace timeline of the SLOWEST one:
    python tools/stall_fit.py n p blocks K method fits"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix

n, p, blocks, K = (int(x) for x in sys.argv[1:5])
method, fits = sys.argv[5], int(sys.argv[6])
ctx = mbpls_amd.get_context()
Ts, Y = planted(n, 1, 32, 0)
offs = [int(round(b * p / blocks)) for b in range(blocks + 1)]
X = DeviceMatrix(ctx, n, p, offs)
X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0)
ctx.sync()
keep = []
for i in range(fits):
    m = MBPLS(n_components=K, method=method)
    keep.append(m)
    t0 = time.perf_counter(); m.fit(X, Y); ctx.sync()
    print(f"fit {i}: {1e3 * (time.perf_counter() - t0):.1f} ms  {({k: round(1e3 * v, 1) for k, v in m.fit_timings_.items()})}", flush=True)
    time.sleep(0.15)
