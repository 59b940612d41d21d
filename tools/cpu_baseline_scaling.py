"""SURVEY 8(d) 'CPU baseline beside it': the faithful oracle (the reference's numpy op sequence, temporaries
included) timed on the host at several widths p of the C2 shape (n=4096, K=10, q=1), a line fitted in p and
extrapolated to p = 1e6 -- the full size needs ~5x X of RAM (164 GB), which the container does not grant.
Also times the lean oracle (the algorithm the GPU runs, on the CPU).  One JSON line.
    python tools/cpu_baseline_scaling.py [p1 p2 p3 ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mbpls_amd.runtime import cpu_quota
from oracle import mbpls_oracle as orc

n, K = 4096, 10
ps = [int(a) for a in sys.argv[1:]] or [16384, 32768, 65536]
thr = cpu_quota()
import threadpoolctl

X, Y = orc.make_latent(n, max(ps), 1, r=32, seed=0)
rows = []
with threadpoolctl.threadpool_limits(limits=thr):
    for p in ps:
        Xp = np.ascontiguousarray(X[:, :p])
        rec = {"p": p}
        for mode, faithful in (("faithful", True), ("lean", False)):
            t0 = time.perf_counter()
            m = orc.fit(Xp, Y, n_components=K, method="NIPALS", faithful=faithful)
            rec[mode + "_s"] = round(time.perf_counter() - t0, 3)
        rec["iterations"] = m.nipals_iterations_
        rows.append(rec)
pp = np.array([r["p"] for r in rows], dtype=float)
out = {"n": n, "K": K, "blas_threads": thr, "cpus_visible": os.cpu_count(), "numpy": np.__version__, "runs": rows}
for mode in ("faithful", "lean"):
    tt = np.array([r[mode + "_s"] for r in rows])
    a, b = np.polyfit(pp, tt, 1)
    out[mode + "_fit"] = {"s_per_1e6_columns": round(a * 1e6, 2), "intercept_s": round(b, 3),
                          "extrapolated_C2_s": round(a * 1e6 + b, 1)}
print(json.dumps(out))
