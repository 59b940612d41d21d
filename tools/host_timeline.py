"""Host-side timeline of ONE warm fit (microseconds from the start of fit): which Python / ctypes calls the head and
the tail of a device-resident fit consist of.   python tools/host_timeline.py [p] [K]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix, _capi, engine, solvers, mbpls as mb

n = 4096
p = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = mbpls_amd.get_context()
Ts, Y = planted(n, 1, 32, 0)
X = DeviceMatrix(ctx, n, p)
X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0)
ctx.sync()
log, depth = [], [0]


def wrap(obj, name, label=None):
    f = getattr(obj, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        depth[0] += 1
        try:
            return f(*a, **k)
        finally:
            depth[0] -= 1
            log.append((t0, time.perf_counter(), depth[0], label or name))
    setattr(obj, name, g)


for nm in ("load_device", "_set_matrix", "slot", "prepare", "_prepare_implicit", "nipals_resident", "use_lazy_deflation",
           "fetch_async", "cross", "right_mul", "host_array", "stream_stats", "wait_stats", "close", "run_deferred",
           "_res_sweep", "_res_loadings", "deferred_first_pass"):
    wrap(engine.HipEngine, nm)
for nm in ("__init__", "__exit__", "step", "stream_out"):
    wrap(engine._ResidentQueue, nm, "rq." + nm)
wrap(solvers, "_finalise_device")
wrap(solvers, "_global_sizes")
wrap(mb, "check_array")
wrap(mb.MBPLS, "_store")
wrap(mb.MBPLS, "_engine")
wrap(_capi.DeviceArray, "set", "DeviceArray.set")
wrap(_capi.DeviceArray, "get", "DeviceArray.get")
import numpy as np
wrap(np.linalg, "pinv")
m = MBPLS(n_components=K)
for i in range(4):
    log.clear()
    t0 = time.perf_counter(); m.fit(X, Y); t1 = time.perf_counter()
print(f"fit: {1e6 * (t1 - t0):.0f} us")
skip = {"_res_sweep", "_res_loadings", "rq.step", "rq.stream_out"}
for a, b, d, nm in sorted(log):
    if nm in skip and not (a - t0 < 700e-6):
        continue
    print(f"{1e6 * (a - t0):9.0f} {1e6 * (b - a):8.0f} us  {'  ' * d}{nm}")
