"""Missing-value passes on long columns: the register shapes (R = 32 / 64 rows per thread, spilling) against the streamed
kernel ("nan_long_from"), per-kernel GB/s from the HIP-event profile.
    python tools/time_sparse_long.py [n ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import mbpls_amd
from mbpls_amd import MBPLS, DeviceMatrix

ns = [int(a) for a in sys.argv[1:]] or [8192, 12000, 16384, 32768, 65536, 200_000]
ctx = mbpls_amd.get_context()
for n in ns:
    p = max(64, int(2e9 / (8 * n)))
    rng = np.random.RandomState(0)
    ts = rng.randn(n, 4)
    X = ts.dot(rng.randn(4, p)) + 0.5 * rng.randn(n, p)
    Y = ts[:, :1] + 0.1 * rng.randn(n, 1)
    X[rng.rand(n, p) < 0.01] = np.nan
    D = DeviceMatrix(ctx, n, p)
    D.upload(X, 0)
    for frm, two_phase in ((65536, 1), (0, 0), (0, 1)):
        if n > 65536 and frm:
            continue
        ctx.set_tunable("nan_long_from", frm)
        ctx.set_tunable("nan_long_two_phase", two_phase)
        m = MBPLS(n_components=3, sparse_data=True)
        for rep in range(3):
            ctx.profile(rep == 2)
            ctx.profile_reset()
            t0 = time.perf_counter(); m.fit(D, Y); ctx.sync(); t1 = time.perf_counter()
        prof = ctx.profile_get()
        ctx.profile(False)
        line = "  ".join(f"{k}: {v['avg_ms']:.2f} ms {v['GBps']:.0f} GB/s x{v['launches']}" for k, v in prof.items())
        print(f"n={n} p={p} {('two phases' if two_phase else 'one workgroup per chunk') if frm == 0 else 'registers'}: fit {1e3 * (t1 - t0):.1f} ms  iterations {m.nipals_iterations_}  {line}", flush=True)
    ctx.set_tunable("nan_long_from", 8192)
    ctx.set_tunable("nan_long_two_phase", 1)
    D.close()
