import sys, time, os, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mbpls_amd
from mbpls_amd import MBPLS, DeviceMatrix, engine
from bench import planted
n, p, K = 4096, 1_000_000, 10
ctx = mbpls_amd.get_context()
Ts, Y = planted(n, 1, 32, 0)
X = DeviceMatrix(ctx, n, p)
X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0)
ctx.sync()
log = []
def wrap(cls, name):
    f = getattr(cls, name)
    def g(self, *a, **k):
        t0 = time.perf_counter(); r = f(self, *a, **k); t1 = time.perf_counter()
        log.append((name, t0, t1)); return r
    setattr(cls, name, g)
for nm in ("prepare", "sweep", "deflate_sweep", "fetch", "close", "load_device"):
    wrap(engine.HipEngine, nm)
from mbpls_amd import _capi
wrap(_capi.DeviceMatrix, "standardize_sweep")
wrap(_capi.DeviceArray, "get")
wrap(_capi.DeviceArray, "set")
wrap(_capi.Context, "_raw_alloc")
wrap(_capi.Context, "_raw_free")
gc.callbacks.append(lambda phase, info: log.append(("gc-%s-gen%d" % (phase, info["generation"]), time.perf_counter(), time.perf_counter())))
m = MBPLS(n_components=K)
for i in range(3):
    log.clear()
    t0 = time.perf_counter(); m.fit(X, Y); t1 = time.perf_counter()
    print(f"fit {i}: {1e3*(t1-t0):.1f} ms")
    prev = t0
    for nm, a, b in log:
        if (b - a) < 1e-3 and (a - prev) < 1e-3 and nm in ("get", "set", "_raw_alloc", "_raw_free"):
            prev = max(prev, b); continue
        print(f"   +{1e3*(a-prev):7.2f} gap | {nm:14s} {1e3*(b-a):7.2f} ms")
        prev = b
    print(f"   +{1e3*(t1-prev):7.2f} tail")
