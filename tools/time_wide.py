"""Few samples x millions of features: the attributes (p x K arrays) are most of the fit -- passes of 1-2 ms against 240 MB
of columns per component over PCIe.  Fit times and host phases per shape / method; MBPLS_COPIER_TRACE=1 shows the copier's
jobs (DMA / normalisation) beside them.   python tools/time_wide.py"""
import os, sys, time, warnings
warnings.simplefilter("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mbpls_amd
from mbpls_amd import MBPLS, DeviceMatrix
from bench import planted
ctx = mbpls_amd.get_context()
for n, p, K in ((100, 10_000_000, 4), (300, 3_333_333, 4), (1000, 1_000_000, 4), (100, 2_000_000, 10)):
    Ts, Y = planted(n, 1, 32, 0)
    X = DeviceMatrix(ctx, n, p); X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0); ctx.sync()
    for method in ("NIPALS", "UNIPALS", "KERNEL", "SIMPLS"):
        ts = []
        for rep in range(4):
            m = MBPLS(n_components=K, method=method)
            t0 = time.perf_counter(); m.fit(X, Y); ts.append(time.perf_counter() - t0)
        attr_bytes = sum(sum(x.nbytes for x in v) if isinstance(v, list) else v.nbytes for v in (getattr(m, a) for a in ("P_", "W_", "W_concat_", "beta_", "R_") if hasattr(m, a)))
        print(f"n={n} p={p} K={K} {method}: fits {[round(1e3*t,1) for t in ts]} ms; attributes {attr_bytes/1e6:.0f} MB -> {attr_bytes/1e9/min(ts):.1f} GB/s if all of it were copies")
        print("   phases (ms):", {k: round(1e3 * v, 1) for k, v in m.fit_timings_.items()}, flush=True)
    X.close(); del X
