"""Where does the REFERENCE algorithm (CPU oracle, explicit deflation) stop meeting ||ts_old - ts|| <= 1e-14 on the
bench's planted data?  n = 4096, one response, K = 38 (run on the build machine: ~100 s at p = 32768).

    python tools/deep_iterations_cpu.py [p] [K] [iteration cap]

Observed (p = 32768, cap 50): iterations [2] * 29 + [6, 50, 50, 4, 50, 50, 8, 50, 50]; residual ||y - Ts Ts^T y|| / ||y|| =
1.6e-2 after 28 components, 5.5e-3 after 30, 1.5e-5 after 38 (explained Y 0.99973 / 0.99997 / 1 - 2e-10): beyond
component 28 the stop rule is out of reach for the reference itself (its loop has no cap, mbpls.py:841).
"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mbpls_oracle as orc

p = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
K = int(sys.argv[2]) if len(sys.argv) > 2 else 38
cap = int(sys.argv[3]) if len(sys.argv) > 3 else 50
X, Y = orc.make_latent(4096, p, 1, seed=0)
t0 = time.time()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m = orc.fit(X, Y, n_components=K, method="NIPALS", faithful=False, max_iter=cap)
print(f"{time.time() - t0:.0f} s; iterations {list(m.nipals_iterations_)}")
Ys = (Y - Y.mean(0)) / Y.std(0)
for k in (10, 20, 25, 28, 30, 34, K):
    r = Ys - m.Ts_[:, :k] @ (m.Ts_[:, :k].T @ Ys)
    print(f"after {k:2d} components: ||y residual|| / ||y|| = {np.linalg.norm(r) / np.linalg.norm(Ys):.2e}, "
          f"explained Y {np.cumsum(m.explained_var_y_)[k - 1]:.10f}")
