"""Host-side time breakdown of one device-resident fit (cProfile) + raw malloc/free/copy costs."""
import cProfile, pstats, sys, time, os
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
t0 = time.perf_counter(); W = DeviceMatrix(ctx, n, p); ctx.sync(); t1 = time.perf_counter()
W.close(); ctx.sync(); t2 = time.perf_counter()
print(f"DeviceMatrix create {1e3*(t1-t0):.1f} ms, close {1e3*(t2-t1):.1f} ms")
a = ctx.empty((p, K), zero=True); ctx.sync()
t0 = time.perf_counter(); h = a.get(); t1 = time.perf_counter()
print(f"D2H p x K ({a.nbytes/1e6:.0f} MB) {1e3*(t1-t0):.1f} ms = {a.nbytes/1e9/(t1-t0):.1f} GB/s")
t0 = time.perf_counter(); a.set(h); ctx.sync(); t1 = time.perf_counter()
print(f"H2D p x K {1e3*(t1-t0):.1f} ms")
m = MBPLS(n_components=K)
m.fit(X, Y)
pr = cProfile.Profile(); pr.enable(); t0 = time.perf_counter(); m.fit(X, Y); t1 = time.perf_counter(); pr.disable()
print(f"fit {1e3*(t1-t0):.1f} ms")
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
