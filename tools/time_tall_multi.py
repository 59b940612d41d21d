import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import mbpls_amd
from mbpls_amd import MBPLS, DeviceMatrix
from bench import planted
ctx = mbpls_amd.get_context()
n, p, q = 1_000_000, 2000, 3
Ts, Y = planted(n, q, 32, 0)
X = DeviceMatrix(ctx, n, p)
X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0)
ctx.sync()
m = MBPLS(n_components=3, method="NIPALS")
for _ in range(2):
    t0 = time.perf_counter(); m.fit(X, Y); t1 = time.perf_counter()
    print(f"tall NIPALS q=3: {1e3*(t1-t0):.1f} ms iters {m.nipals_iterations_}", {k: round(1e3*v,1) for k,v in m.fit_timings_.items()})
