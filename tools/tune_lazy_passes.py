"""The two passes of a never-writes-X fit -- statistics + first sweep, loadings + lazy deflation + next sweep (k = 3
earlier components) -- against chunks per CU / rows per thread / columns in flight, at several sample counts.
    python tools/tune_lazy_passes.py [GB] n [n ...]       (TUNE_R / TUNE_D / TUNE_C: comma lists; 0 = the default rule)"""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import mbpls_amd
from mbpls_amd import DeviceMatrix, _capi

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
ns = [int(a) for a in sys.argv[2:]] or [2048, 5000, 8192]
Rs = [int(x) for x in os.environ.get("TUNE_R", "0,8,16").split(",")]
Ds = [int(x) for x in os.environ.get("TUNE_D", "2").split(",")]
Cs = [int(x) for x in os.environ.get("TUNE_C", "4,8").split(",")]
ctx = mbpls_amd.get_context()
rng = np.random.RandomState(0)
KP = 3
for n in ns:
    p = int(gb * 1e9 / 8 / n)
    v = rng.standard_normal(n); v -= v.mean()
    ts = rng.standard_normal(n); ts -= ts.mean(); ts /= np.linalg.norm(ts)
    dv, dts = ctx.from_host(v), ctx.from_host(ts)
    dT = ctx.from_host(rng.standard_normal((n, 8)))
    for cpc in Cs:
        ctx.set_tunable("chunks_per_cu", cpc)
        X = DeviceMatrix(ctx, n, p)
        X.fill_latent(dT, 8, 0.5, 1, 0)
        mean, scale, rs, r2, g, pv = (ctx.empty(p) for _ in range(6))
        P = ctx.from_host(0.01 * rng.standard_normal((p, KP + 1)))
        t, s1, s2, s3, s4 = ctx.empty(n), ctx.empty(1), ctx.empty(1), ctx.empty(1), ctx.empty(1)
        cu, ce, dd = ctx.from_host(rng.standard_normal(40)), ctx.from_host(rng.standard_normal(40)), ctx.empty(40)
        for R, D in itertools.product(Rs, Ds):
            ctx.set_tunable("rows_per_thread", R); ctx.set_tunable("prefetch", D)
            try:
                X.standardize_stats_sweep(_capi.SCALE_FIT, mean, scale, rs, r2, s1, s2, dv, 1.0, g, t, s3); ctx.sync()
            except _capi.HipError as e:
                print("skip", n, cpc, R, D, str(e)[:60]); continue
            ctx.profile(True); ctx.profile_reset()
            for _ in range(5):
                X.standardize_stats_sweep(_capi.SCALE_FIT, mean, scale, rs, r2, s1, s2, dv, 1.0, g, t, s3)
                X.loadings_sweep_lazy(dts, 1.0, pv, s4, dv, 1.0, g, t, s3, P, p, KP, cu, ce, dd, r2=r2)
                X.sweep_lazy(dv, 1.0, g, t, s3, P, p, KP, cu, dd, r2=r2)
            pr = ctx.profile_get(); ctx.profile(False)
            print(f"n={n:6d} chunks/CU {cpc} rows/thread {R:2d} in flight {D}: statistics {pr['standardize_sweep']['GBps']:5.0f}  "
                  f"loadings+sweep {pr['deflate_sweep']['GBps']:5.0f}  sweep {pr['sweep']['GBps']:5.0f} GB/s", flush=True)
        ctx.set_tunable("rows_per_thread", 0); ctx.set_tunable("prefetch", 2)
        X.close()
        for a in (mean, scale, rs, r2, g, pv, P, t, s1, s2, s3, s4, cu, ce, dd):
            a.free()
    ctx.set_tunable("chunks_per_cu", 4)
