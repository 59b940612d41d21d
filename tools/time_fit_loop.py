"""What a caller sees who fits in a loop and lets the previous model go (component scans, hand-written validation loops):
the fit and the release of the model before it, at BASELINE size.   python tools/time_fit_loop.py [p] [K] [fits]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix

n = 4096
p = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
fits = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ctx = mbpls_amd.get_context()
Ts, Y = planted(n, 1, 32, 0)
X = DeviceMatrix(ctx, n, p)
X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0)
ctx.sync()
m = None
t_loop = time.perf_counter()
for i in range(fits):
    t0 = time.perf_counter()
    new = MBPLS(n_components=K).fit(X, Y)
    t1 = time.perf_counter()
    m = None                                    # the previous model goes
    t2 = time.perf_counter()
    m = new
    del new
    print(f"fit {1e3 * (t1 - t0):6.1f} ms   letting the previous model go {1e3 * (t2 - t1):5.1f} ms", flush=True)
print(f"{fits} iterations: {1e3 * (time.perf_counter() - t_loop) / fits:.1f} ms each")
