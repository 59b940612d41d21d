"""Long columns (n > 8192): two-phase streamed passes vs clusters of workgroups (one read of X per pass):
python tools/time_long.py n p [K]"""
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
K = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for thr in (-1, 0, 1, 2, 3):
    ctx.set_tunable("cluster_passes", 0 if thr < 0 else 1)
    ctx.set_tunable("cluster_exchange", min(max(thr, 0), 2))        # 3: shared exchange wave WITH prefetch
    ctx.set_tunable("cluster_prefetch", 1 if thr == 3 else 0)
    for method in ("NIPALS", "UNIPALS"):
        m = MBPLS(n_components=K, method=method)
        for rep in range(3):
            ctx.profile(rep == 2)
            ctx.profile_reset()
            t0 = time.perf_counter(); m.fit(X, Y); ctx.sync(); t1 = time.perf_counter()
        prof = ctx.profile_get()
        ctx.profile(False)
        print(f"n={n} p={p} cluster_exchange={thr} {method}: fit {1e3 * (t1 - t0):.1f} ms  " +
              "  ".join(f"{k}: {v['avg_ms']:.2f} ms {v['GBps']:.0f} GB/s x{v['launches']}" for k, v in prof.items()),
              flush=True)
        print("    host phases (ms):", {k: round(1e3 * v, 1) for k, v in m.fit_timings_.items()}, "timeouts", ctx.cluster_status(), flush=True)
