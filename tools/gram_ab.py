"""Gram kernel A/B: operand loads in front of the MFMAs of a step (0) or between them (1): python tools/gram_ab.py [p] [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mbpls_amd
from mbpls_amd import DeviceMatrix
p = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
ctx = mbpls_amd.get_context()
X = DeviceMatrix(ctx, n, p)
X.fill_latent(ctx.from_host(np.random.RandomState(0).randn(n, 8)), 8, 0.5, 1, 0)
G = ctx.empty((n, n))
ref = None
for rep in range(2):
    for il in (0, 1):
        ctx.set_tunable("gram_interleave", il)
        X.gram(G, n); ctx.sync()
        ts = []
        for _ in range(3):
            ctx.sync(); t0 = time.perf_counter(); X.gram(G, n); ctx.sync(); ts.append(time.perf_counter() - t0)
        g = G.get()
        if ref is None:
            ref = g
        t = min(ts)
        print(f"n={n} p={p} interleave={il}: {1e3 * t:.1f} ms  {n * (n + 1.0) * p / t / 1e12:.2f} TFLOP/s  "
              f"(max rel. diff vs first {float(np.max(np.abs(g - ref)) / np.max(np.abs(ref))):.1e})", flush=True)
