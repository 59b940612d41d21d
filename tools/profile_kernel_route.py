"""Where a KERNEL fit of a given shape spends its time on either route: per-kernel HIP-event times and a host profile.
    python tools/profile_kernel_route.py n p q K [blocks]"""
import cProfile
import io
import os
import pstats
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter("ignore")
import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix

n, p, q, K = (int(v) for v in sys.argv[1:5])
blocks = int(sys.argv[5]) if len(sys.argv) > 5 else 2
ctx = mbpls_amd.get_context()
Ts, Y = planted(n, q, 32, seed=0)
offs = [int(round(b * p / blocks)) for b in range(blocks + 1)]
X = DeviceMatrix(ctx, n, p, offs)
dT = ctx.from_host(Ts)
X.fill_latent(dT, 32, 0.5, 1234, 0)
ctx.sync()
for route, env in (("gram", "1"), ("products", "0")):
    os.environ["MBPLS_KERNEL_GRAM"] = env
    for _ in range(2):
        MBPLS(n_components=K, method="KERNEL").fit(X, Y)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); m = MBPLS(n_components=K, method="KERNEL").fit(X, Y); ts.append(time.perf_counter() - t0)
    print(f"== {route}: n={n} p={p} q={q} K={K}: fits {[round(1e3 * t, 2) for t in ts]} ms")
    ctx.profile(True); ctx.profile_reset()
    MBPLS(n_components=K, method="KERNEL").fit(X, Y)
    prof = ctx.profile_get(); ctx.profile(False)
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
        print(f"   {k:28s} x{v['launches']:4d}  total {v['total_ms']:8.3f} ms  avg {v['avg_ms']:7.3f} ms")
    pr = cProfile.Profile(); pr.enable()
    MBPLS(n_components=K, method="KERNEL").fit(X, Y)
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:4500])
X.close()
