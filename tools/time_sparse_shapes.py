"""Missing-value passes across sample counts (1 GB matrices, 5 % NaN): python tools/time_sparse_shapes.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import mbpls_amd
from mbpls_amd import MBPLS, DeviceMatrix

ctx = mbpls_amd.get_context()
rng = np.random.default_rng(0)
for n in (64, 128, 256, 512, 1024, 4096):
    p = int(1e9 / 8 / n)
    T = rng.standard_normal((n, 4))
    X = T @ rng.standard_normal((4, p)) + 0.3 * rng.standard_normal((n, p))
    Y = T[:, :1] + 0.1 * rng.standard_normal((n, 1))
    X[rng.random((n, p)) < 0.05] = np.nan
    dX = DeviceMatrix(ctx, n, p)
    dX.upload(X, 0)
    m = MBPLS(n_components=3, method="NIPALS", sparse_data=True)
    for rep in range(3):
        ctx.profile(rep == 2)
        ctx.profile_reset()
        t0 = time.perf_counter(); m.fit(dX, Y); ctx.sync(); t1 = time.perf_counter()
    prof = ctx.profile_get()
    ctx.profile(False)
    print(f"n={n:5d} p={p:8d} sparse NIPALS: fit {1e3 * (t1 - t0):7.1f} ms iters {m.nipals_iterations_}  " +
          "  ".join(f"{k}: {v['avg_ms']:.2f} ms {v['GBps']:.0f} GB/s x{v['launches']}" for k, v in prof.items()),
          flush=True)
    dX.close()
