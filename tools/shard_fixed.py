"""The shard-sized fit of bench.py's `shard_proxy` leg by itself (this problem / 8 through the RCCL communicator with one
rank): fit wall, sum of passes, fixed cost -- for iterating on what a fit spends outside its passes.
    python tools/shard_fixed.py [--local]        (--local: LocalComm instead of the one-rank RCCL communicator)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mbpls_amd
from bench import planted, shard_proxy
from mbpls_amd import MBPLS, DeviceMatrix

n, p, K = 4096, 125_000, 10
ctx = mbpls_amd.get_context()
if "--cpc" in sys.argv:
    ctx.set_tunable("chunks_per_cu", int(sys.argv[sys.argv.index("--cpc") + 1]))
Ts, Y = planted(n, 1, 32, 0)
dT = ctx.from_host(Ts)
if "--with-big" in sys.argv:          # as in bench.py's side-leg process: the full matrix resident, two fits of it first
    Xbig = DeviceMatrix(ctx, n, 1_000_000, [0, 1_000_000])
    Xbig.fill_latent(dT, 32, 0.5, 1234, 0)
    ctx.sync()
    big = MBPLS(n_components=K)
    for _ in range(2):
        big.fit(Xbig, Y)
if "--local" in sys.argv:
    X = DeviceMatrix(ctx, n, p, [0, p])
    X.fill_latent(dT, 32, 0.5, 1234, 0)
    ctx.sync()
    for _ in range(3):
        MBPLS(n_components=K).fit(X, Y)
    for prof in (False, False, True):
        nosync = "--nosync" in sys.argv
        ctx.profile(prof); ctx.profile_reset()
        walls, keep = [], []
        for _ in range(10):
            if not nosync:
                ctx.sync()
            t1 = time.perf_counter()
            keep.append(MBPLS(n_components=K).fit(X, Y))
            walls.append(time.perf_counter() - t1)
        out = {"profile": prof, "fit_wall_ms": round(1e3 * float(np.median(walls)), 3), "all": [round(1e3 * w, 2) for w in walls]}
        if prof:
            pr = ctx.profile_get()
            passes = sum(v["total_ms"] for v in pr.values()) / 10
            out.update(sum_of_passes_ms=round(passes, 3), fixed_ms=round(out["fit_wall_ms"] - passes, 3))
        ctx.profile(False)
        print(json.dumps(out))
else:
    print(json.dumps(shard_proxy(ctx, mbpls_amd, dT, Y, 32, n, p, K, 0.106)))
