"""Measured device ceilings beside the 8 TB/s HBM3E spec: device-to-device copy of a C2-sized matrix
(read + write) and the column passes' own SCALE_NONE pass (read + write through the pass machinery).
    python tools/copy_ceiling.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import mbpls_amd
from mbpls_amd import DeviceMatrix, _capi

n, p = 4096, 1_000_000
ctx = mbpls_amd.get_context()
A = DeviceMatrix(ctx, n, p)
B = DeviceMatrix(ctx, n, p)
A.fill_latent(ctx.from_host(np.random.RandomState(0).randn(n, 8)), 8, 0.5, 1, 0)
ctx.sync()
nbytes = 8.0 * n * p
for name, fn in (("hipMemcpy D2D", lambda: B.copy_from(A)),
                 ("column pass SCALE_NONE (copy through registers)",
                  lambda: A.standardize_sweep(B, _capi.SCALE_NONE, None, None, None))):
    fn(); ctx.sync()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); fn(); ctx.sync(); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[2]
    print(f"{name}: {1e3 * t:.2f} ms  read+write {2 * nbytes / t / 1e9:.0f} GB/s", flush=True)
