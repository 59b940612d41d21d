"""cProfile of one warm fit (host side): where a host-driven solver spends its time between the passes.
    python tools/profile_fit.py --n 20000 --p 1000 --q 10 --K 20 --blocks 2 [--method NIPALS] [--top 35]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=20000)
ap.add_argument("--p", type=int, default=1000)
ap.add_argument("--q", type=int, default=10)
ap.add_argument("--K", type=int, default=20)
ap.add_argument("--blocks", type=int, default=2)
ap.add_argument("--method", default="NIPALS")
ap.add_argument("--top", type=int, default=35)
ap.add_argument("--calc-all", action="store_true")
a = ap.parse_args()
import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix

ctx = mbpls_amd.get_context()
Ts, Y = planted(a.n, a.q, 32, 0)
offs = [int(round(b * a.p / a.blocks)) for b in range(a.blocks + 1)]
X = DeviceMatrix(ctx, a.n, a.p, offs)
X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0)
ctx.sync()
mk = lambda: MBPLS(n_components=a.K, method=a.method, calc_all=a.calc_all)
for _ in range(3):
    t = time.perf_counter(); m = mk().fit(X, Y); print(f"fit {1e3 * (time.perf_counter() - t):.2f} ms", m.fit_plan_.get("solver"))
pr = cProfile.Profile()
m = mk()
pr.runcall(m.fit, X, Y)
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(a.top)
