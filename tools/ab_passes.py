"""Kernel averages of a few warm fits (HIP-event profile), no checks: python tools/ab_passes.py [p] [K]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix

n = 4096
p = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = mbpls_amd.get_context()
Ts, Y = planted(n, 1, 32, 0)
X = DeviceMatrix(ctx, n, p)
X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0)
ctx.sync()
m = MBPLS(n_components=K)
for _ in range(3):
    m.fit(X, Y)
ctx.profile(True); ctx.profile_reset()
import time
t0 = time.perf_counter()
for _ in range(10):
    m.fit(X, Y)
dt = (time.perf_counter() - t0) / 10
pr = ctx.profile_get()
print(f"fit {1e3 * dt:.3f} ms", {k: round(v["avg_ms"], 4) for k, v in pr.items()})
