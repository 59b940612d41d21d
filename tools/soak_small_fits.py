"""Thousands of small fits from host arrays, rotating shapes / methods / responses (upload-matrix cache, bulk copies, pools):
host RSS and free device memory must not drift.   python tools/soak_small_fits.py [fits]"""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter("ignore")
import numpy as np
import psutil

import mbpls_amd
from mbpls_amd import MBPLS

fits = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
ctx = mbpls_amd.get_context()
rng = np.random.RandomState(0)
shapes = [(40, 200, 1), (150, 3000, 1), (60, 500, 3), (33, 1000, 2), (300, 800, 1), (90, 64, 1), (25, 2500, 1)]
data = []
for n, p, q in shapes:
    T = rng.standard_normal((n, 3))
    X = T.dot(rng.standard_normal((3, p))) + 0.3 * rng.standard_normal((n, p))
    Y = T[:, :q] + 0.1 * rng.standard_normal((n, q))
    data.append(([np.ascontiguousarray(X[:, :p // 2]), np.ascontiguousarray(X[:, p // 2:])], Y))
methods = ["NIPALS", "UNIPALS", "KERNEL", "SIMPLS"]
proc = psutil.Process()


def snapshot():
    ctx.sync()
    return proc.memory_info().rss / 1e6, ctx.info()["free_bytes"] / 1e6, ctx._pool_bytes / 1e6


def run(count):
    for i in range(count):
        X, Y = data[i % len(data)]
        m = MBPLS(n_components=2 + i % 3, method=methods[(i // len(data)) % 4]).fit(X, Y)
        if i % 50 == 0:
            m.predict(X)


run(300)
r0 = snapshot()
t0 = time.perf_counter()
run(fits)
dt = time.perf_counter() - t0
r1 = snapshot()
run(fits)
r2 = snapshot()
print(f"{fits} fits in {dt:.1f} s ({1e3 * dt / fits:.2f} ms each)")
for name, a, b, c in zip(("host RSS MB", "device free MB", "pool MB"), r0, r1, r2):
    print(f"{name:15s} after warm-up {a:10.1f}   after {fits} fits {b:10.1f}   after {2 * fits} fits {c:10.1f}")
print("upload matrices kept:", {k: round(v.nbytes / 1e6, 2) for k, v in ctx._mat_cache.items()})
