"""Latency of fits at the sizes the reference's own examples use (hundreds of samples, thousands of features), from
host numpy arrays, every method; beside the CPU oracle's lean mode.   python tools/time_small.py"""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

warnings.simplefilter("ignore")
from mbpls_amd import MBPLS
from oracle import mbpls_oracle as orc

for n, p, q, K in ((150, 3000, 1, 5), (150, 3000, 3, 5), (1000, 20000, 1, 8), (40, 200, 1, 2)):
    X, Y = orc.make_latent(n, p, q, seed=1)
    Xs = [np.ascontiguousarray(X[:, :p // 2]), np.ascontiguousarray(X[:, p // 2:])]
    line = f"n={n} p={p} q={q} K={K}:"
    for method in ("NIPALS", "UNIPALS", "KERNEL", "SIMPLS"):
        m = MBPLS(n_components=K, method=method)
        m.fit(Xs, Y)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); m.fit(Xs, Y); ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); orc.fit(Xs, Y, n_components=K, method=method); tc = time.perf_counter() - t0
        line += f"  {method} {1e3 * sorted(ts)[3]:.2f} ms (cpu lean {1e3 * tc:.1f})"
    print(line, flush=True)
    ph = {k: round(1e3 * v, 2) for k, v in m.fit_timings_.items()}
print("last fit phases (SIMPLS):", ph)
m = MBPLS(n_components=5).fit(Xs, Y)
print("NIPALS phases:", {k: round(1e3 * v, 2) for k, v in m.fit_timings_.items()})
