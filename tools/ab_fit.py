"""Same-box A/B of the C2 fit: implicit standardisation (never writes X) vs the stored standardised copy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix
n, p, K = 4096, 1_000_000, 10
ctx = mbpls_amd.get_context()
Ts, Y = planted(n, 1, 32, 0)
X = DeviceMatrix(ctx, n, p)
X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0)
ctx.sync()
m = MBPLS(n_components=K)
for rnd in range(3):
    for mode in ("1", "0"):
        os.environ["MBPLS_IMPLICIT_STD"] = mode
        m.fit(X, Y); ctx.sync()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); m.fit(X, Y); ctx.sync(); ts.append(1e3 * (time.perf_counter() - t0))
        ctx.profile(True); ctx.profile_reset(); m.fit(X, Y); ctx.sync(); pr = ctx.profile_get(); ctx.profile(False)
        print(f"implicit={mode}: median {sorted(ts)[2]:.1f} ms  " + "  ".join(f"{k} {v['avg_ms']:.2f}" for k, v in pr.items()), flush=True)
