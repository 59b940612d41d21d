"""Statistics-only first pass (n=4096, p=1e6) against rows per thread / columns in flight / chunks per CU.
    python tools/tune_stats_pass.py"""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mbpls_amd
from mbpls_amd import DeviceMatrix, _capi

n, p = 4096, 1_000_000
ctx = mbpls_amd.get_context()
rng = np.random.RandomState(0)
v = rng.standard_normal(n); v -= v.mean()
dv = ctx.from_host(v)
dT = ctx.from_host(rng.standard_normal((n, 8)))
for cpc in (4, 8, 2):
    ctx.set_tunable("chunks_per_cu", cpc)
    X = DeviceMatrix(ctx, n, p)
    X.fill_latent(dT, 8, 0.5, 1, 0)
    mean, scale, rs, r2, g = (ctx.empty(p) for _ in range(5))
    t, s1, s2, s3 = ctx.empty(n), ctx.empty(1), ctx.empty(1), ctx.empty(1)
    for R, D in itertools.product((0, 8, 16), (2, 1, 3)):
        ctx.set_tunable("rows_per_thread", R); ctx.set_tunable("prefetch", D)
        try:
            X.standardize_stats_sweep(_capi.SCALE_FIT, mean, scale, rs, r2, s1, s2, dv, 1.0, g, t, s3); ctx.sync()
        except _capi.HipError as e:
            print("skip", cpc, R, D, str(e)[:60]); continue
        ctx.profile(True); ctx.profile_reset()
        for _ in range(6):
            X.standardize_stats_sweep(_capi.SCALE_FIT, mean, scale, rs, r2, s1, s2, dv, 1.0, g, t, s3)
            X.sweep(dv, 1.0, g, t, s3)
        pr = ctx.profile_get(); ctx.profile(False)
        print(f"chunks/CU {cpc} rows/thread {R:2d} in flight {D}: statistics {pr['standardize_sweep']['avg_ms']:.3f} ms "
              f"{pr['standardize_sweep']['GBps']:.0f} GB/s   sweep {pr['sweep']['avg_ms']:.3f} ms", flush=True)
    ctx.set_tunable("rows_per_thread", 0); ctx.set_tunable("prefetch", 2)
    X.close()
