"""Sweep the launch tunables of the fused column-pass kernels and print achieved GB/s (HIP events).

    python tools/tune_colpass.py [n] [p]
"""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import mbpls_amd
from mbpls_amd import DeviceMatrix, _capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
p = int(sys.argv[2]) if len(sys.argv) > 2 else 250_000
reps = 5
ctx = mbpls_amd.get_context()
rng = np.random.RandomState(0)
v = rng.standard_normal(n)
ts = rng.standard_normal(n); ts /= np.linalg.norm(ts)
dv, dts = ctx.from_host(v), ctx.from_host(ts)
dT = ctx.from_host(rng.standard_normal((n, 8)))
results = []
Rs = [int(x) for x in os.environ.get("TUNE_R", "4,8,16").split(",")]
Ds = [int(x) for x in os.environ.get("TUNE_D", "1,2,3").split(",")]
Cs = [int(x) for x in os.environ.get("TUNE_C", "2,4,8").split(",")]
for cpc in Cs:
    ctx.set_tunable("chunks_per_cu", cpc)
    X = DeviceMatrix(ctx, n, p)
    X.fill_latent(dT, 8, 0.5, 1, 0)
    W = DeviceMatrix(ctx, n, p)
    B = 1
    d_mean, d_scale = ctx.empty(p), ctx.empty(p)
    d_s = [ctx.empty(B) for _ in range(3)]
    d_g, d_t, d_p = ctx.empty(p), ctx.empty(n * B), ctx.empty(p)
    for R, D in itertools.product(Rs, Ds):
        ctx.set_tunable("rows_per_thread", R)
        ctx.set_tunable("prefetch", D)
        try:
            X.standardize_sweep(W, _capi.SCALE_FIT, d_mean, d_scale, d_s[0], dv, 1.0, d_g, d_t, d_s[1])
            ctx.sync()
        except _capi.HipError as e:
            print("skip", cpc, R, D, e)
            continue
        ctx.profile(True); ctx.profile_reset()
        for _ in range(reps):
            X.standardize_sweep(W, _capi.SCALE_FIT, d_mean, d_scale, d_s[0], dv, 1.0, d_g, d_t, d_s[1])
            W.sweep(dv, 1.0, d_g, d_t, d_s[1])
            W.deflate_sweep(dts, 1.0, True, d_p, d_s[2], dv, 1.0, d_g, d_t, d_s[1])
            W.deflate_sweep(dts, 1.0, False, d_p, d_s[2])
        pr = ctx.profile_get(); ctx.profile(False)
        row = (cpc, R, D, pr["standardize_sweep"]["GBps"], pr["sweep"]["GBps"], pr["deflate_sweep"]["GBps"],
               pr["deflate"]["GBps"])
        results.append(row)
        print("chunks/CU %d  R %2d  D %d | standardize %6.0f  sweep %6.0f  deflate_sweep %6.0f  loadings %6.0f GB/s"
              % row, flush=True)
    X.close(); W.close()
best = max(results, key=lambda r: r[5])
print("best deflate_sweep:", best)
best = max(results, key=lambda r: r[4])
print("best sweep:", best)
