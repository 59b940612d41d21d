"""Which ENGINE CALL of a fit is the slow one when a fit of a small shape now and then takes 20 ms instead of 4?
Every HipEngine method the solvers use is wrapped with a timer (no extra synchronisation); the slowest call of each
fit is printed.   python tools/stall_calls.py n p q K method [env=value ...]"""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter("ignore")
for kv in sys.argv[6:]:
    k, v = kv.split("=")
    os.environ[k] = v
import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix
from mbpls_amd.engine import HipEngine

n, p, q, K = (int(v) for v in sys.argv[1:5])
method = sys.argv[5]
log = []


def wrap(name):
    real = getattr(HipEngine, name)

    def timed(self, *a, **k):
        t0 = time.perf_counter()
        try:
            return real(self, *a, **k)
        finally:
            log.append((time.perf_counter() - t0, name))
    setattr(HipEngine, name, timed)


for nm in ("prepare", "prepare_stats", "sweep", "deflate_sweep", "right_mul", "cross", "fetch", "fetch_col", "xt_v", "_xw",
           "slot", "host_array", "gram", "gram_apply", "close", "load_device", "use_lazy_deflation", "put", "sync"):
    wrap(nm)
ctx = mbpls_amd.get_context()
Ts, Y = planted(n, q, 32, seed=0)
X = DeviceMatrix(ctx, n, p, [0, p // 2, p])
dT = ctx.from_host(Ts)
X.fill_latent(dT, 32, 0.5, 1234, 0)
ctx.sync()
for i in range(12):
    log.clear()
    t0 = time.perf_counter()
    m = MBPLS(n_components=K, method=method).fit(X, Y)
    dt = time.perf_counter() - t0
    top = sorted(log, reverse=True)[:3]
    print(f"fit {i}: {1e3 * dt:7.2f} ms; in engine calls {1e3 * sum(t for t, _ in log):7.2f} ms; slowest: "
          + ", ".join(f"{nm} {1e3 * t:.2f}" for t, nm in top)
          + f"; phases {({k: round(1e3 * v, 2) for k, v in m.fit_timings_.items() if v > 2e-4})}", flush=True)
X.close()
