"""End-to-end fit() from HOST numpy (row-major) including the upload: the PCIe-inclusive figure."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mbpls_amd import MBPLS
from oracle import mbpls_oracle as orc
n, p = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
X, Y = orc.make_latent(n, p, 1, seed=0)
for order in ("C", "F"):
    Xo = np.asarray(X, order=order)
    m = MBPLS(n_components=10)
    m.fit(Xo, Y)
    t0 = time.perf_counter(); m.fit(Xo, Y); dt = time.perf_counter() - t0
    print(f"host {order}-order n={n} p={p} ({X.nbytes/1e9:.2f} GB): fit {dt:.3f} s  -> {X.nbytes/1e9/dt:.2f} GB/s of input; phases",
          {k: round(v, 3) for k, v in m.fit_timings_.items()})
