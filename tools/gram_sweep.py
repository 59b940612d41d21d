"""Gram (X X^T, f64 MFMA) rate vs p at fixed n: python tools/gram_sweep.py [n]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import mbpls_amd
from mbpls_amd import DeviceMatrix

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ctx = mbpls_amd.get_context()
ldg = n
G = ctx.empty((ldg, n))
if len(sys.argv) > 2:
    ctx.set_tunable("gram_slice_features", int(sys.argv[2]))
for p in ((2_000_000,) if len(sys.argv) > 2 else (2048, 8192, 32768, 131072, 524288, 2_000_000)):
    X = DeviceMatrix(ctx, n, p)
    dT = ctx.from_host(np.random.RandomState(0).randn(n, 8))
    X.fill_latent(dT, 8, 0.5, 1, 0)
    ctx.sync()
    X.gram(G, ldg)
    ctx.sync()
    ctx.profile(True); ctx.profile_reset()
    reps = 3 if p < 1_000_000 else 1
    for _ in range(reps):
        X.gram(G, ldg)
    ctx.sync()
    pr = ctx.profile_get()["gram"]; ctx.profile(False)
    tf = float(n) * (n + 1) * p / (pr["avg_ms"] * 1e-3) / 1e12
    print(f"n={n} p={p}: {pr['avg_ms']:.2f} ms  {tf:.1f} TFLOP/s  ({tf / 78.6:.1%} of 78.6)", flush=True)
    X.close(); dT.free()
