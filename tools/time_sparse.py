"""Timing of the sparse_data=True path (missing values): per-kernel GB/s from the HIP-event profile.
    python tools/time_sparse.py [p] [frac_missing] [rows_per_thread]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import mbpls_amd
from mbpls_amd import MBPLS, DeviceMatrix

n, K = 4096, 5
p = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
pb = 50_000
rng = np.random.RandomState(0)
ts = rng.randn(n, 4)
blk = ts.dot(rng.randn(4, pb)) + 0.5 * rng.randn(n, pb)
Y = ts[:, :1] + 0.1 * rng.randn(n, 1)
blk[rng.rand(n, pb) < frac] = np.nan
ctx = mbpls_amd.get_context()
if len(sys.argv) > 3:
    ctx.set_tunable("rows_per_thread", int(sys.argv[3]))
X = DeviceMatrix(ctx, n, p)
for o in range(0, p, pb):
    X.upload(blk[:, :min(pb, p - o)], o)
ctx.sync()
for sparse in (True, False) if frac == 0.0 else (True,):
    m = MBPLS(n_components=K, sparse_data=sparse)
    for rep in range(3):
        ctx.profile(rep == 2)
        ctx.profile_reset()
        t0 = time.perf_counter(); m.fit(X, Y); ctx.sync(); t1 = time.perf_counter()
        print(f"sparse={sparse} fit {rep}: {1e3 * (t1 - t0):.1f} ms  iterations {m.nipals_iterations_}", flush=True)
    print({k: round(1e3 * v, 1) for k, v in m.fit_timings_.items()})
    prof = ctx.profile_get()
    ctx.profile(False)
    print(prof)
