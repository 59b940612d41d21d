"""The statistics pass (+ first sweep) alone, N launches, for rocprofv3 --pmc runs: python tools/stats_one.py n p [launches]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import mbpls_amd
from mbpls_amd import DeviceMatrix, _capi

n, p = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ctx = mbpls_amd.get_context()
rng = np.random.RandomState(0)
v = rng.standard_normal(n); v -= v.mean()
dv = ctx.from_host(v)
dT = ctx.from_host(rng.standard_normal((n, 8)))
X = DeviceMatrix(ctx, n, p)
X.fill_latent(dT, 8, 0.5, 1, 0)
mean, scale, rs, r2, g = (ctx.empty(p) for _ in range(5))
t, s1, s2, s3 = ctx.empty(n), ctx.empty(1), ctx.empty(1), ctx.empty(1)
ctx.profile(True); ctx.profile_reset()
for _ in range(reps):
    X.standardize_stats_sweep(_capi.SCALE_FIT, mean, scale, rs, r2, s1, s2, dv, 1.0, g, t, s3)
    X.sweep(dv, 1.0, g, t, s3)
ctx.sync()
pr = ctx.profile_get()
print(n, p, {k: (round(v["avg_ms"], 3), round(v["GBps"])) for k, v in pr.items()})
