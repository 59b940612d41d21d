import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mbpls_amd
from mbpls_amd import MBPLS, DeviceMatrix, _capi
from bench import planted
n, p, K = 4096, 1_000_000, 10
ctx = mbpls_amd.get_context()
Ts, Y = planted(n, 1, 32, 0)
X = DeviceMatrix(ctx, n, p)
X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0)
ctx.sync()
m = MBPLS(n_components=K)
m.fit(X, Y); m.fit(X, Y)
def T(tag, f):
    t0 = time.perf_counter(); r = f(); ctx.sync(); print(f"  {tag:28s} {1e3*(time.perf_counter()-t0):8.2f} ms", flush=True); return r
for rep in range(2):
    print("manual sequence", rep)
    W = T("DeviceMatrix create", lambda: DeviceMatrix(ctx, n, p))
    s1 = T("slot alloc+memset", lambda: ctx.empty((p, K), zero=True))
    s2 = T("slot alloc+memset", lambda: ctx.empty((p, K), zero=True))
    mean = T("mean/scale alloc", lambda: (ctx.empty(p), ctx.empty(p)))
    v = T("put vec", lambda: ctx.from_host(Y[:, 0]))
    xs = ctx.empty(n + 3)
    T("standardize_sweep", lambda: X.standardize_sweep(W, _capi.SCALE_FIT, mean[0], mean[1], xs.ptr + 8 * (n + 2), v, 1.0, s1.col(0), xs.ptr, xs.ptr + 8 * n))
    T("standardize_sweep again", lambda: X.standardize_sweep(W, _capi.SCALE_FIT, mean[0], mean[1], xs.ptr + 8 * (n + 2), v, 1.0, s1.col(0), xs.ptr, xs.ptr + 8 * n))
    T("sweep", lambda: W.sweep(v, 1.0, s1.col(0), xs.ptr, xs.ptr + 8 * n))
    T("full fit", lambda: m.fit(X, Y))
    T("standardize_sweep after fit", lambda: X.standardize_sweep(W, _capi.SCALE_FIT, mean[0], mean[1], xs.ptr + 8 * (n + 2), v, 1.0, s1.col(0), xs.ptr, xs.ptr + 8 * n))
    T("standardize_sweep again", lambda: X.standardize_sweep(W, _capi.SCALE_FIT, mean[0], mean[1], xs.ptr + 8 * (n + 2), v, 1.0, s1.col(0), xs.ptr, xs.ptr + 8 * n))
    W.close(); s1.free(); s2.free()
