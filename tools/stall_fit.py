"""Repeated fits of one shape with pauses between them, for a rocprofv3 --kernel-trace timeline of the SLOWEST one:
    python tools/stall_fit.py n p blocks K method fits"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix

n, p, blocks, K = (int(x) for x in sys.argv[1:5])
method, fits = sys.argv[5], int(sys.argv[6])
ctx = mbpls_amd.get_context()
Ts, Y = planted(n, 1, 32, 0)
offs = [int(round(b * p / blocks)) for b in range(blocks + 1)]
X = DeviceMatrix(ctx, n, p, offs)
X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0)
ctx.sync()
keep = []


def throttled():
    """(periods throttled, microseconds throttled) of this cgroup's CPU quota so far"""
    try:
        d = dict(line.split() for line in open("/sys/fs/cgroup/cpu.stat"))
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0))
    except Exception:
        return (0, 0)


for i in range(fits):
    m = MBPLS(n_components=K, method=method)
    keep.append(m)
    th0 = throttled()
    t0 = time.perf_counter(); m.fit(X, Y); ctx.sync()
    th1 = throttled()
    print(f"fit {i}: {1e3 * (time.perf_counter() - t0):.1f} ms  throttled +{th1[0] - th0[0]} periods +{(th1[1] - th0[1]) / 1e3:.1f} ms  "
          f"{({k: round(1e3 * v, 1) for k, v in m.fit_timings_.items()})}", flush=True)
    time.sleep(0.15)
