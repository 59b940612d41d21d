"""Host-side timeline of ONE fit (microseconds from the start of fit): which Python / ctypes calls the head, the tail and
any stall of a device-resident fit consist of.
    python tools/host_timeline.py [--n 4096] [--p 125000] [--K 10] [--method NIPALS] [--blocks 1] [--fits 4]
                                  [--show first|last|slowest] [--fresh]
--show first  = the cold fit of the process;  --fresh = a new estimator per fit (as bench.py), kept alive."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--p", type=int, default=125_000)
ap.add_argument("--K", type=int, default=10)
ap.add_argument("--method", default="NIPALS")
ap.add_argument("--blocks", type=int, default=1)
ap.add_argument("--q", type=int, default=1)
ap.add_argument("--fits", type=int, default=4)
ap.add_argument("--show", default="last")
ap.add_argument("--fresh", action="store_true")
ap.add_argument("--tail-profile", action="store_true", help="sys.setprofile from the end of the queue's read-back to the end of fit")
ap.add_argument("--head-profile", action="store_true", help="sys.setprofile from the start of fit to the start of the queue")
ap.add_argument("--rccl", action="store_true", help="the sharded code path with ONE rank (RCCL communicator)")
ap.add_argument("--min-us", type=float, default=0.0)
a = ap.parse_args()

import numpy as np
import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix, _capi, engine, solvers, mbpls as mb

n, p, K = a.n, a.p, a.K
ctx = mbpls_amd.get_context()
if a.rccl:
    mbpls_amd.set_comm(mbpls_amd.RcclComm(ctx, 0, 1, lambda payload: payload))
Ts, Y = planted(n, a.q, 32, 0)
offs = [int(round(b * p / a.blocks)) for b in range(a.blocks + 1)]
X = DeviceMatrix(ctx, n, p, offs)
X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0)
ctx.sync()
log, depth = [], [0]


def wrap(obj, name, label=None):
    f = getattr(obj, name)

    def g(*aa, **k):
        t0 = time.perf_counter()
        depth[0] += 1
        try:
            return f(*aa, **k)
        finally:
            depth[0] -= 1
            log.append((t0, time.perf_counter(), depth[0], label or name))
    setattr(obj, name, g)


for nm in ("load_device", "_set_matrix", "slot", "prepare", "_prepare_implicit", "nipals_resident", "unipals_resident",
           "use_lazy_deflation", "fetch", "fetch_async", "cross", "right_mul", "host_array", "stream_stats", "wait_stats",
           "close", "run_deferred", "_res_sweep", "_res_loadings", "deferred_first_pass", "sync", "check_cluster_passes"):
    wrap(engine.HipEngine, nm)
for nm in ("__init__", "__exit__", "step", "stream_out"):
    wrap(engine._ResidentQueue, nm, "rq." + nm)
for nm in ("sweep", "deflate_sweep", "xt_v", "x_w_slot", "nipals_subspace_loop", "fetch_col", "prepare_stats", "put", "_xw", "x_w_host"):
    if hasattr(engine.HipEngine, nm):
        wrap(engine.HipEngine, nm)
wrap(solvers, "_finalise_device")
wrap(solvers, "_global_sizes")
wrap(mb, "check_array")
wrap(mb.MBPLS, "_store")
wrap(mb.MBPLS, "_engine")
wrap(_capi.DeviceArray, "set", "DeviceArray.set")
wrap(_capi.DeviceArray, "get", "DeviceArray.get")
wrap(_capi.Context, "sync", "ctx.sync")
wrap(_capi.Context, "copier_wait", "copier_wait")
wrap(np.linalg, "pinv")
wrap(np.linalg, "svd")
# device allocations that miss the pool / frees that leave it (the runtime calls behind them may wait for the device)
real_malloc, real_free = ctx.lib.mbpls_hip_malloc, ctx.lib.mbpls_hip_free


def t_malloc(*aa):
    t0 = time.perf_counter()
    try:
        return real_malloc(*aa)
    finally:
        log.append((t0, time.perf_counter(), depth[0], f"hipMalloc {int(aa[1]) / 1e6:.2f} MB (pool miss)"))


def t_free(*aa):
    t0 = time.perf_counter()
    try:
        return real_free(*aa)
    finally:
        log.append((t0, time.perf_counter(), depth[0], "hipFree (pool full)"))


ctx.lib.mbpls_hip_malloc, ctx.lib.mbpls_hip_free = t_malloc, t_free

# the device-to-host copies: of the copier thread (behind events) and the blocking ones of the fitting thread
import threading


def t_copy(name, real, size_arg):
    def f(*aa):
        t0 = time.perf_counter()
        try:
            return real(*aa)
        finally:
            who = "copier" if threading.current_thread().name == "mbpls-copier" else "main"
            log.append((t0, time.perf_counter(), 6 if who == "copier" else depth[0],
                        f"[{who}] {name} {int(aa[size_arg]) / 1e6:.3f} MB"))
    return f


ctx.lib.mbpls_hip_memcpy_d2h = t_copy("d2h", ctx.lib.mbpls_hip_memcpy_d2h, 3)

tail_events = []
if a.head_profile:
    def tracer(frame, event, arg):
        t = time.perf_counter()
        if event in ("call", "return"):
            co = frame.f_code
            tail_events.append((t, event, f"{os.path.basename(co.co_filename)}:{co.co_firstlineno} {co.co_name}"))
        elif event in ("c_call", "c_return"):
            tail_events.append((t, "call" if event == "c_call" else "return", f"<C> {getattr(arg, '__qualname__', getattr(arg, '__name__', str(arg)))}"))

    real_res = engine.HipEngine.nipals_resident

    def res_stop(self, *aa, **kk):
        sys.setprofile(None)
        return real_res(self, *aa, **kk)

    engine.HipEngine.nipals_resident = res_stop
if a.tail_profile:
    real_d2h = ctx.lib.mbpls_hip_memcpy_d2h

    def tracer(frame, event, arg):
        t = time.perf_counter()
        if event in ("call", "return"):
            co = frame.f_code
            tail_events.append((t, event, f"{os.path.basename(co.co_filename)}:{co.co_firstlineno} {co.co_name}"))
        elif event in ("c_call", "c_return"):
            tail_events.append((t, "call" if event == "c_call" else "return", f"<C> {getattr(arg, '__qualname__', getattr(arg, '__name__', str(arg)))}"))

    def d2h_then_trace(*aa):
        r = real_d2h(*aa)
        if int(aa[3]) < 100_000 and sys._getframe(1).f_code.co_name == "__exit__":
            tail_events.clear()
            tail_events.append((time.perf_counter(), "mark", "read-back done"))
            sys.setprofile(tracer)
        return r

    ctx.lib.mbpls_hip_memcpy_d2h = d2h_then_trace
runs, keep = [], []
m = MBPLS(n_components=K, method=a.method)
for i in range(a.fits):
    log.clear()
    if a.fresh:
        m = MBPLS(n_components=K, method=a.method)
        keep.append(m)
    if a.head_profile:
        tail_events.clear()
        tail_events.append((time.perf_counter(), 'mark', 'fit starts'))
        sys.setprofile(tracer)
    t0 = time.perf_counter(); m.fit(X, Y); t1 = time.perf_counter()
    sys.setprofile(None)
    tail = list(tail_events)
    runs.append((t1 - t0, t0, list(log), dict(m.fit_timings_), tail))
print("fits (ms):", [round(1e3 * r[0], 2) for r in runs])
pick = {"first": 0, "last": len(runs) - 1, "slowest": int(np.argmax([r[0] for r in runs]))}[a.show]
dt, t0, lg, phases, tail = runs[pick]
print(f"fit #{pick}: {1e6 * dt:.0f} us   phases (ms): " + str({k: round(1e3 * v, 2) for k, v in phases.items()}))
skip = {"_res_sweep", "_res_loadings", "rq.step", "rq.stream_out"}
for s0, s1, d, nm in sorted(lg):
    if (nm in skip or nm.startswith("[copier]")) and not (s0 - t0 < 700e-6 or s1 - t0 > dt - 1500e-6):
        continue
    if 1e6 * (s1 - s0) < a.min_us:
        continue
    print(f"{1e6 * (s0 - t0):9.0f} {1e6 * (s1 - s0):8.0f} us  {'  ' * d}{nm}")

if (a.tail_profile or a.head_profile) and tail:
    print("---- tail (profiled: every call costs ~1 us extra) ----")
    base = tail[0][0]
    stack = []
    for t, ev, name in tail[1:]:
        if ev == "call":
            stack.append((t, name))
        elif stack:
            t_in, nm = stack.pop()
            if 1e6 * (t - t_in) >= max(a.min_us, 3.0):
                print(f"{1e6 * (t_in - base):8.0f} {1e6 * (t - t_in):7.0f} us  {'  ' * len(stack)}{nm}")
    print(f"{1e6 * (tail[-1][0] - base):8.0f} end of the profiled region")
