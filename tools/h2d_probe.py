"""Host->device copy latency right after a fit (tall data): python tools/h2d_probe.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mbpls_amd
from mbpls_amd import MBPLS, DeviceMatrix
from bench import planted
ctx = mbpls_amd.get_context()
a = np.random.rand(1_000_000)
d = ctx.empty(1_000_000)
d.set(a); ctx.sync()
n, p = 1_000_000, 2000
Ts, Y = planted(n, 1, 32, 0)
X = DeviceMatrix(ctx, n, p)
X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0)
ctx.sync()
m = MBPLS(n_components=3, method="NIPALS")
out = []
for _ in range(8):
    t0 = time.perf_counter(); m.fit(X, Y); t1 = time.perf_counter(); ctx.sync()
    t2 = time.perf_counter(); d.set(a); t3 = time.perf_counter(); ctx.sync()
    out.append(f"fit {1e3*(t1-t0):.1f} set {1e3*(t3-t2):.2f}")
print(os.environ.get("PROBE_TAG", ""), " | ".join(out))
