"""cProfile of one warm fit: python tools/profile_fit.py n p [method] [K]  (host-side hot spots of a device-resident fit)"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbpls_amd
from mbpls_amd import MBPLS, DeviceMatrix
from bench import planted

n, p = int(sys.argv[1]), int(sys.argv[2])
method = sys.argv[3] if len(sys.argv) > 3 else "NIPALS"
K = int(sys.argv[4]) if len(sys.argv) > 4 else 3
ctx = mbpls_amd.get_context()
Ts, Y = planted(n, 1, 32, 0)
X = DeviceMatrix(ctx, n, p)
X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0)
ctx.sync()
m = MBPLS(n_components=K, method=method)
for _ in range(3):
    m.fit(X, Y)
pr = cProfile.Profile()
pr.enable()
m.fit(X, Y)
pr.disable()
print({k: round(1e3 * v, 1) for k, v in m.fit_timings_.items()})
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
