"""x_w (OUT = X W) for c right-hand sides: row-slab kernel (8 per pass) vs matrix-core kernel (64 per pass).
    python tools/time_xw.py [n] [p]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import mbpls_amd
from mbpls_amd import DeviceMatrix

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
p = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
ctx = mbpls_amd.get_context()
X = DeviceMatrix(ctx, n, p)
X.fill_latent(ctx.from_host(np.random.RandomState(0).randn(n, 8)), 8, 0.5, 1, 0)
ctx.sync()
for c in (4, 8, 10, 16, 20, 32, 48, 64):
    W = ctx.from_host(np.random.RandomState(c).randn(p, c))
    out = ctx.empty((n, c))
    line = f"n={n} p={p} c={c}:"
    for thr, name in ((1000, "row-slab"), (0, "mfma")):
        ctx.set_tunable("xw_mfma_from", thr)
        X.x_w(W, c, out); ctx.sync()
        ctx.profile(True); ctx.profile_reset()
        for _ in range(3):
            X.x_w(W, c, out)
        ctx.sync()
        pr = ctx.profile_get()["x_w"]; ctx.profile(False)
        ms = pr["total_ms"] / 3
        line += f"  {name} {ms:.2f} ms ({8.0 * n * p / ms / 1e6:.0f} GB/s per X-read-equivalent, {pr['launches'] // 3} pass)"
    print(line, flush=True)
    W.free(); out.free()
ctx.set_tunable("xw_mfma_from", 8)
