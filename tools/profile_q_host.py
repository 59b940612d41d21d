"""cProfile of one warm several-response fit on a resident matrix (by own time): what the host does between the passes.
python tools/profile_q_host.py [n] [p] [K] [q] [method] [blocks] [top]"""
import cProfile
import os
import pstats
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter("ignore")
import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix

arg = lambda i, d: type(d)(sys.argv[i]) if len(sys.argv) > i else d
n, p, K, q, method, B, top = arg(1, 4096), arg(2, 1_000_000), arg(3, 10), arg(4, 10), arg(5, "NIPALS"), arg(6, 1), arg(7, 30)
ctx = mbpls_amd.get_context()
Ts, Y = planted(n, q, 32, 0)
offs = [int(round(p * b / B)) for b in range(B + 1)]
X = DeviceMatrix(ctx, n, p, offs)
X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0)
ctx.sync()
for i in range(3):
    t0 = time.perf_counter()
    m = MBPLS(n_components=K, method=method).fit(X, Y)
    print(f"fit {i}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
pr = cProfile.Profile()
pr.enable()
m = MBPLS(n_components=K, method=method).fit(X, Y)
pr.disable()
print(m.fit_plan_.get("solver"))
print({k: round(1e3 * v, 2) for k, v in m.fit_timings_.items() if v > 2e-4})
pstats.Stats(pr).sort_stats("tottime").print_stats(top)
