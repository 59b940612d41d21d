"""Durations of every host<->device copy and allocation inside one warm fit (tall data): python tools/h2d_probe.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mbpls_amd
from mbpls_amd import MBPLS, DeviceMatrix, _capi
from bench import planted
ctx = mbpls_amd.get_context()
n, p = 1_000_000, 2000
Ts, Y = planted(n, 1, 32, 0)
X = DeviceMatrix(ctx, n, p)
X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0)
ctx.sync()
m = MBPLS(n_components=3, method=sys.argv[1] if len(sys.argv) > 1 else "NIPALS")
for _ in range(4):
    m.fit(X, Y)
log = []
t_start = [0.0]
def wrap(cls, name):
    f = getattr(cls, name)
    def g(self, *a, **k):
        t0 = time.perf_counter(); r = f(self, *a, **k); t1 = time.perf_counter()
        nb = getattr(self, "nbytes", 0) if name in ("set", "get") else (a[0] if a else 0)
        log.append(f"{1e3*(t0-t_start[0]):7.2f} {name:10s} {nb:>10} B {1e3*(t1-t0):7.2f} ms")
        return r
    setattr(cls, name, g)
wrap(_capi.DeviceArray, "set"); wrap(_capi.DeviceArray, "get")
wrap(type(ctx), "_raw_alloc"); wrap(type(ctx), "_raw_free")
t_start[0] = time.perf_counter()
m.fit(X, Y)
print("fit", round(1e3 * (time.perf_counter() - t_start[0]), 1), {k: round(1e3*v,1) for k,v in m.fit_timings_.items()})
print("\n".join(l for l in log if float(l.split()[-2]) > 0.3))
