import sys, time, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mbpls_amd
from mbpls_amd import DeviceMatrix, _capi
n, p = 4096, 1_000_000
ctx = mbpls_amd.get_context()
X = DeviceMatrix(ctx, n, p); W = DeviceMatrix(ctx, n, p)
dT = ctx.from_host(np.random.randn(n, 8)); X.fill_latent(dT, 8, 0.5, 1, 0); W.fill_latent(dT, 8, 0.5, 1, 0)
v = ctx.from_host(np.random.randn(n)); g = ctx.empty(p); t = ctx.empty(n); s = ctx.empty(1); pv = ctx.empty(p)
mean, scale = ctx.empty(p), ctx.empty(p)
def rounds(tag, k=3):
    out = []
    for _ in range(k):
        t0 = time.perf_counter(); X.standardize_sweep(W, 0, mean, scale, s, v, 1.0, g, t, s); t.get(); out.append(1e3 * (time.perf_counter() - t0))
    print(f"{tag:40s}", " ".join(f"{x:7.2f}" for x in out), flush=True)
rounds("baseline"); rounds("baseline")
h = C.c_void_p()
ctx.lib.mbpls_hip_malloc(ctx.handle, 4096, C.byref(h)); rounds("after tiny hipMalloc")
ctx.lib.mbpls_hip_free(ctx.handle, h); rounds("after tiny hipFree")
hs = []
for i in range(8):
    ctx.lib.mbpls_hip_malloc(ctx.handle, 4096, C.byref(h)); hs.append(h.value)
rounds("after 8 tiny hipMalloc")
for x in hs: ctx.lib.mbpls_hip_free(ctx.handle, x)
rounds("after 8 tiny hipFree")
ctx.lib.mbpls_hip_malloc(ctx.handle, 80_000_000, C.byref(h)); rounds("after 80MB hipMalloc")
ctx.lib.mbpls_hip_free(ctx.handle, h); rounds("after 80MB hipFree")
a = np.empty((1_000_000, 10)); a[:] = 1
b = [a.copy() for _ in range(5)]; rounds("after numpy 400MB alloc")
b = None; rounds("after numpy 400MB free")
big = ctx.empty((1_000_000, 10), zero=True); hb = [big.get() for _ in range(4)]; rounds("after 4x80MB D2H")
hb = None; rounds("after free of D2H targets")
t0 = time.perf_counter(); Z = DeviceMatrix(ctx, n, p); print("fresh 32GB create %.2f ms" % (1e3*(time.perf_counter()-t0)))
t0 = time.perf_counter(); X.standardize_sweep(Z, 0, mean, scale, s, v, 1.0, g, t, s); t.get(); print("first write pass into fresh 32GB: %.2f ms" % (1e3*(time.perf_counter()-t0)))
t0 = time.perf_counter(); X.standardize_sweep(Z, 0, mean, scale, s, v, 1.0, g, t, s); t.get(); print("second: %.2f ms" % (1e3*(time.perf_counter()-t0)))
rounds("W after fresh alloc")
a = np.asfortranarray(np.random.randn(1_000_000, 10)); b = np.asfortranarray(np.random.randn(1_000_000, 10))
t0 = time.perf_counter(); c = a.T.dot(b); print("BLAS K x p . p x K: %.2f ms" % (1e3*(time.perf_counter()-t0)))
rounds("after BLAS cross product")
t0 = time.perf_counter(); r = a.dot(c); print("BLAS p x K . K x K: %.2f ms" % (1e3*(time.perf_counter()-t0)))
rounds("after BLAS right-multiply")
t0 = time.perf_counter(); r = a / np.sqrt((a ** 2).sum(axis=0)); print("numpy colnorm: %.2f ms" % (1e3*(time.perf_counter()-t0)))
rounds("after numpy elementwise")
import threadpoolctl
print(threadpoolctl.threadpool_info())
with threadpoolctl.threadpool_limits(8):
    t0 = time.perf_counter(); c = a.T.dot(b); r = a.dot(c); print("BLAS (8 threads) both: %.2f ms" % (1e3*(time.perf_counter()-t0)))
rounds("after BLAS with 8 threads")
