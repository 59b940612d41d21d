"""nipals_iterations_ of deep components at full C2 size (lazy deflation): python tools/iters_deep.py [p] [K ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix

n = 4096
p = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
Ks = [int(k) for k in sys.argv[2:]] or [20, 30, 38]
ctx = mbpls_amd.get_context()
Ts, Y = planted(n, 1, 32, 0)
X = DeviceMatrix(ctx, n, p)
X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0)
ctx.sync()
for K in Ks:
    m = MBPLS(n_components=K)
    m.fit(X, Y)
    t0 = time.perf_counter(); m.fit(X, Y); dt = time.perf_counter() - t0
    it = list(m.nipals_iterations_)
    print(f"K={K}: fit {1e3 * dt:.1f} ms, passes {1 + sum(it) + K - 1}, iterations {it}", flush=True)
