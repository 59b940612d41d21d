"""Which fit of a sequence is slow, and does the cyclic GC have a hand in it?  python tools/outlier_probe.py [gc_off]"""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix

mode = sys.argv[1] if len(sys.argv) > 1 else "default"
ctx = mbpls_amd.get_context()
n, p, K = 4096, 1_000_000, 10
Ts, Y = planted(n, 1, 32, 0)
X = DeviceMatrix(ctx, n, p, [0, p])
X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0)
ctx.sync()
if mode == "gc_off":
    gc.disable()
mdl = MBPLS(n_components=K)
rows = []
for i in range(14):
    if mode == "gc_freeze" and i == 2:
        gc.collect(); gc.freeze()
    c0 = gc.get_count()
    t0 = time.perf_counter()
    mdl.fit(X, Y)
    dt = time.perf_counter() - t0
    rows.append((round(1e3 * dt, 1), c0, round(1e3 * mdl.fit_timings_.get("wait_kernel+d2h_small", 0), 1)))
print(mode)
for r in rows:
    print(r)
