"""A tall one-response fit on the TWO-PHASE long-column passes ("cluster_passes" = 0: what a fit falls back to after a
cluster hand-off time-out or a refused cooperative launch) for a rocprofv3 --kernel-trace timeline; prints the per-pass
HIP-event profile and the host phases of the last fit.   python tools/two_phase_one.py n p [fits] [method]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix

n, p = int(sys.argv[1]), int(sys.argv[2])
fits = int(sys.argv[3]) if len(sys.argv) > 3 else 3
method = sys.argv[4] if len(sys.argv) > 4 else "NIPALS"
ctx = mbpls_amd.get_context()
ctx.set_tunable("cluster_passes", 0)
Ts, Y = planted(n, 1, 32, 0)
X = DeviceMatrix(ctx, n, p)
X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0)
ctx.sync()
for i in range(fits):
    ctx.profile(True); ctx.profile_reset()
    t0 = time.perf_counter()
    m = MBPLS(n_components=3, method=method).fit(X, Y)
    ctx.sync()
    dt = time.perf_counter() - t0
    prof = ctx.profile_get(); ctx.profile(False)
    print(f"fit {i}: {1e3 * dt:.2f} ms  " + "  ".join(f"{k}: {v['avg_ms']:.3f} ms x{v['launches']}" for k, v in prof.items()),
          flush=True)
    if dt > 0.010:
        print("   slow:", {k: round(1e3 * v, 2) for k, v in m.fit_timings_.items() if v > 5e-4}, flush=True)
print(m.fit_plan_)
print({k: round(1e3 * v, 2) for k, v in m.fit_timings_.items()})
