"""Register-resident (spilling above n = 8192) vs streamed-column passes: python tools/time_long.py n p"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import mbpls_amd
from mbpls_amd import MBPLS, DeviceMatrix
from bench import planted

n, p = int(sys.argv[1]), int(sys.argv[2])
ctx = mbpls_amd.get_context()
Ts, Y = planted(n, 1, 32, 0)
X = DeviceMatrix(ctx, n, p)
X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0)
ctx.sync()
for thr in ([65536, 0] if n <= 65536 else [8192]):
    ctx.set_tunable("long_columns_from", thr)
    for method in ("NIPALS", "UNIPALS"):
        m = MBPLS(n_components=3, method=method)
        for rep in range(3):
            ctx.profile(rep == 2)
            ctx.profile_reset()
            t0 = time.perf_counter(); m.fit(X, Y); ctx.sync(); t1 = time.perf_counter()
        prof = ctx.profile_get()
        ctx.profile(False)
        print(f"n={n} p={p} long_columns_from={thr} {method}: fit {1e3 * (t1 - t0):.1f} ms  " +
              "  ".join(f"{k}: {v['avg_ms']:.2f} ms {v['GBps']:.0f} GB/s x{v['launches']}" for k, v in prof.items()),
              flush=True)
        print("    host phases (ms):", {k: round(1e3 * v, 1) for k, v in m.fit_timings_.items()}, flush=True)
