import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mbpls_amd
from mbpls_amd import DeviceMatrix
n, p = 4096, 500_000
ctx = mbpls_amd.get_context()
X = DeviceMatrix(ctx, n, p)
dT = ctx.from_host(np.random.randn(n, 8)); X.fill_latent(dT, 8, 0.5, 1, 0)
v = ctx.from_host(np.random.randn(n)); g = ctx.empty(p); t = ctx.empty(n); s = ctx.empty(1)
big = ctx.empty((1_000_000, 10), zero=True)
def rounds(tag, k=4):
    out = []
    for _ in range(k):
        t0 = time.perf_counter(); X.sweep(v, 1.0, g, t, s); t.get(); out.append(1e3 * (time.perf_counter() - t0))
    print(f"{tag:34s}", " ".join(f"{x:7.2f}" for x in out), flush=True)
rounds("baseline")
t0 = time.perf_counter(); h = big.get(); print(f"big D2H fresh: {1e3*(time.perf_counter()-t0):.2f} ms")
rounds("after big D2H (array alive)")
t0 = time.perf_counter(); del h; print(f"del big: {1e3*(time.perf_counter()-t0):.2f} ms")
rounds("after del big")
hs = [big.get() for _ in range(4)]
rounds("after 4 big D2H")
t0 = time.perf_counter(); hs = None; print(f"del 4 big: {1e3*(time.perf_counter()-t0):.2f} ms")
rounds("after del 4 big")
a = np.empty((1_000_000, 10)); a[:] = 1.0
t0 = time.perf_counter(); b = a / 3.0; c = (a ** 2).sum(axis=0); print(f"numpy ops: {1e3*(time.perf_counter()-t0):.2f} ms")
rounds("after numpy ops")
t0 = time.perf_counter(); big.set(a); ctx.sync(); print(f"big H2D: {1e3*(time.perf_counter()-t0):.2f} ms")
rounds("after big H2D")
W = DeviceMatrix(ctx, n, p); rounds("after DeviceMatrix create")
W.close(); rounds("after DeviceMatrix close (pooled)")
small = [ctx.empty(n) for _ in range(8)]; [x.free() for x in small]; rounds("after small alloc/free (hipFree)")
