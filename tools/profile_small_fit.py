"""cProfile of warm small fits from host arrays (two blocks): python tools/profile_small_fit.py [method] [n p K] [fits]"""
import cProfile, os, pstats, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
warnings.simplefilter("ignore")
from mbpls_amd import MBPLS
from oracle import mbpls_oracle as orc
method = sys.argv[1] if len(sys.argv) > 1 else "NIPALS"
n, p, K = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (150, 3000, 5)
fits = int(sys.argv[5]) if len(sys.argv) > 5 else 200
X, Y = orc.make_latent(n, p, 1, r=min(6, p), seed=1)
Xs = [np.ascontiguousarray(X[:, :p // 2]), np.ascontiguousarray(X[:, p // 2:])]
m = MBPLS(n_components=K, method=method)
for _ in range(5):
    m.fit(Xs, Y)
t0 = time.perf_counter()
for _ in range(fits):
    m.fit(Xs, Y)
print(f"n={n} p={p} K={K} {method}: {1e3 * (time.perf_counter() - t0) / fits:.3f} ms per fit (no profiler)")
pr = cProfile.Profile(); pr.enable()
for _ in range(fits):
    m.fit(Xs, Y)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
