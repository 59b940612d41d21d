import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mbpls_amd
from mbpls_amd import DeviceMatrix
n, p = 4096, 200_000
ctx = mbpls_amd.get_context()
X = DeviceMatrix(ctx, n, p)
dT = ctx.from_host(np.random.randn(n, 8)); X.fill_latent(dT, 8, 0.5, 1, 0)
v = ctx.from_host(np.random.randn(n)); g = ctx.empty(p); t = ctx.empty(n); s = ctx.empty(1)
def one():
    t0 = time.perf_counter(); X.sweep(v, 1.0, g, t, s); t.get(); return 1e3 * (time.perf_counter() - t0)
for _ in range(5): one()
print("back-to-back:", " ".join(f"{one():6.2f}" for _ in range(8)))
for idle in (0.001, 0.003, 0.01, 0.03, 0.06, 0.1, 0.3, 1.0, 3.0):
    r = []
    for _ in range(6):
        time.sleep(idle); r.append(one())
    print(f"idle {1e3*idle:7.1f} ms (sleep):", " ".join(f"{x:6.2f}" for x in r), flush=True)
for idle in (0.01, 0.06, 0.3):
    r = []
    for _ in range(6):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < idle: pass
        r.append(one())
    print(f"idle {1e3*idle:7.1f} ms (spin): ", " ".join(f"{x:6.2f}" for x in r), flush=True)
