#!/bin/bash
python3 - <<'PY'
import time, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from oracle import mbpls_oracle as orc
from mbpls_amd import MBPLS
X, Y = orc.make_latent(150, 3000, 1, r=6, seed=1)
blocks = [np.ascontiguousarray(X[:, :1000]), np.ascontiguousarray(X[:, 1000:])]
for method in ("NIPALS", "UNIPALS", "KERNEL", "SIMPLS"):
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(5): MBPLS(n_components=5, method=method).fit(blocks, Y)
        ts = []
        for _ in range(30):
            t = time.perf_counter(); m = MBPLS(n_components=5, method=method).fit(blocks, Y); ts.append(time.perf_counter() - t)
    print(method, "fit from host arrays: median %.3f ms  min %.3f ms" % (1e3 * np.median(ts), 1e3 * min(ts)), m.fit_plan_.get("solver"))
PY
