"""Where the host spends a several-response fit (warm): totals per engine call.
python tools/host_profile_q.py [p] [K] [q] [method] [blocks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix, _capi, engine, solvers, mbpls as mb

n = 4096
p = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
q = int(sys.argv[3]) if len(sys.argv) > 3 else 10
method = sys.argv[4] if len(sys.argv) > 4 else "NIPALS"
B = int(sys.argv[5]) if len(sys.argv) > 5 else 2
ctx = mbpls_amd.get_context()
Ts, Y = planted(n, q, 32, 0)
offs = [int(round(p * b / B)) for b in range(B + 1)]
X = DeviceMatrix(ctx, n, p, offs)
X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0)
ctx.sync()
log, depth = [], [0]


def wrap(obj, name, label=None):
    f = getattr(obj, name)

    def g(*a, **k):
        t0 = time.perf_counter(); depth[0] += 1
        try:
            return f(*a, **k)
        finally:
            depth[0] -= 1
            log.append((t0, time.perf_counter(), depth[0], label or name))
    setattr(obj, name, g)


for nm in ("load_device", "slot", "prepare", "prepare_stats", "xt_v", "x_w_slot", "x_w", "deflate_sweep", "deflate",
           "nipals_subspace_loop", "fetch", "fetch_async", "cross", "right_mul", "host_array", "close", "sync",
           "unit_columns", "scale_rows", "run_deferred", "wait_stats"):
    if hasattr(engine.HipEngine, nm):
        wrap(engine.HipEngine, nm)
wrap(solvers, "_finalise_device"); wrap(mb.MBPLS, "_store"); wrap(np.linalg, "pinv"); wrap(np.linalg, "svd")
wrap(_capi.DeviceArray, "set", "DeviceArray.set"); wrap(_capi.DeviceArray, "get", "DeviceArray.get")
m = MBPLS(n_components=K, method=method)
for i in range(3):
    log.clear()
    t0 = time.perf_counter(); m.fit(X, Y); t1 = time.perf_counter()
print(f"{method} q={q} p={p} K={K} B={B}: fit {1e3 * (t1 - t0):.1f} ms")
tot = {}
for a, b, d, nm in log:
    c = tot.setdefault((d, nm), [0, 0.0]); c[0] += 1; c[1] += b - a
top = sum(v[1] for (d, nm), v in tot.items() if d == 0)
for (d, nm), v in sorted(tot.items(), key=lambda kv: (kv[0][0], -kv[1][1])):
    print(f"  depth {d} {nm:24s} x{v[0]:4d}  {1e3 * v[1]:8.2f} ms")
print(f"  outside the wrapped calls: {1e3 * (t1 - t0 - top):.2f} ms")
print("  timings:", {k: round(v, 2) for k, v in getattr(m, "timings_", {}).items()} if hasattr(m, "timings_") else None)
