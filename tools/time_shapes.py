"""Pass bandwidth across sample counts at a fixed matrix size (default 8 GB): python tools/time_shapes.py [GB] [method] [responses] [n,n,...]
Few samples x very many features is the everyday shape of multiblock PLS (spectra, omics)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import mbpls_amd
from mbpls_amd import MBPLS, DeviceMatrix
from bench import planted

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
method = sys.argv[2] if len(sys.argv) > 2 else "NIPALS"
q = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ctx = mbpls_amd.get_context()
sizes = (40, 64, 100, 128, 200, 256, 300, 500, 1000, 1024, 1200, 1500, 1800, 2048, 2500, 3000, 3500, 4096, 5000, 6000, 8192, 12000)
if len(sys.argv) > 4:
    sizes = tuple(int(x) for x in sys.argv[4].split(","))
for n in sizes:
    p = int(gb * 1e9 / 8 / n)
    Ts, Y = planted(n, q, 16, 0)
    X = DeviceMatrix(ctx, n, p, [0, p // 3, p])
    X.fill_latent(ctx.from_host(Ts), 16, 0.5, 1234, 0)
    ctx.sync()
    keep = []
    for rep in range(3):
        ctx.profile(rep == 2)
        ctx.profile_reset()
        m = MBPLS(n_components=5, method=method)      # a fresh estimator per fit (as bench.py), released after the timing
        keep.append(m)
        t0 = time.perf_counter(); m.fit(X, Y); ctx.sync(); t1 = time.perf_counter()
    prof = ctx.profile_get()
    ctx.profile(False)
    print(f"n={n:6d} p={p:9d} {method} q={q}: fit {1e3 * (t1 - t0):7.1f} ms  " +
          "  ".join(f"{k}: {v['avg_ms']:.2f} ms {v['GBps']:.0f} GB/s x{v['launches']}" for k, v in prof.items()),
          flush=True)
    X.close()
    del X, keep, m
