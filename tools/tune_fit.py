"""C2 fit (default never-writes-X path) under the launch tunables `prefetch` and `chunks_per_cu`.
    python tools/tune_fit.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix
n, p, K = 4096, 1_000_000, 10
ctx = mbpls_amd.get_context()
Ts, Y = planted(n, 1, 32, 0)
dT = ctx.from_host(Ts)
for cpc in (2, 3, 4, 6, 8):
    ctx.set_tunable("chunks_per_cu", cpc)
    X = DeviceMatrix(ctx, n, p)
    X.fill_latent(dT, 32, 0.5, 1234, 0)
    ctx.sync()
    m = MBPLS(n_components=K)
    for D in (1, 2, 3):
        ctx.set_tunable("prefetch", D)
        m.fit(X, Y); ctx.sync()
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); m.fit(X, Y); ctx.sync(); ts.append(1e3 * (time.perf_counter() - t0))
        ctx.profile(True); ctx.profile_reset(); m.fit(X, Y); ctx.sync(); pr = ctx.profile_get(); ctx.profile(False)
        print(f"chunks/CU {cpc} prefetch {D}: median {sorted(ts)[1]:.1f} ms  " + "  ".join(f"{k} {v['avg_ms']:.2f}" for k, v in pr.items()), flush=True)
    X.close()
