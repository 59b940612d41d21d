import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mbpls_amd
from mbpls_amd import MBPLS, DeviceMatrix
from bench import planted
n, p, K = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 10
ctx = mbpls_amd.get_context()
Ts, Y = planted(n, 1, 32, 0)
X = DeviceMatrix(ctx, n, p)
X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0)
ctx.sync()
m = MBPLS(n_components=K)
for prof in (False, True, False):
    ctx.profile(prof)
    for i in range(4):
        t0 = time.perf_counter(); m.fit(X, Y); ctx.sync(); t1 = time.perf_counter()
        print(f"profile={prof} fit {i}: {1e3*(t1-t0):.1f} ms", flush=True)
print(ctx.profile_get())
ctx.profile(False)
for i in range(3):
    t0 = time.perf_counter(); m.fit(X, Y); t1 = time.perf_counter()
    print({k: round(1e3 * v, 1) for k, v in m.fit_timings_.items()})
