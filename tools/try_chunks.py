"""chunks_per_cu of the register-resident passes: fit wall, pass and (from rocprof elsewhere) reduce times"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix
n, K = 4096, 10
ctx = mbpls_amd.get_context()
Ts, Y = planted(n, 1, 32, 0)
dT = ctx.from_host(Ts)
for p in (125_000, 1_000_000):
    for cpc in (4, 2, 1, 3):
        ctx.set_tunable("chunks_per_cu", cpc)
        X = DeviceMatrix(ctx, n, p, [0, p])
        X.fill_latent(dT, 32, 0.5, 1234, 0)
        ctx.sync()
        for _ in range(2):
            MBPLS(n_components=K).fit(X, Y)
        walls = []
        for _ in range(6):
            ctx.sync(); t = time.perf_counter(); MBPLS(n_components=K).fit(X, Y); walls.append(time.perf_counter() - t)
        ctx.profile(True); ctx.profile_reset()
        for _ in range(3):
            MBPLS(n_components=K).fit(X, Y)
        pr = ctx.profile_get(); ctx.profile(False)
        print(json.dumps({"p": p, "chunks_per_cu": cpc, "fit_ms": round(1e3 * float(np.median(walls)), 3),
                          "passes_ms": round(sum(v["total_ms"] for v in pr.values()) / 3, 3),
                          "avg": {k: round(v["avg_ms"], 4) for k, v in pr.items()}}), flush=True)
        X.close()
