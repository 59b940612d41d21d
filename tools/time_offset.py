"""Never-writes-X fits of offset data (raw spectra: columns many standard deviations from zero) at BASELINE size: the
centring flavour of the passes against the plain one, the first fit of a matrix (stopped after its statistics pass and
queued again) against the following ones (engine.HipEngine._offset_hint).   python tools/time_offset.py [p] [K]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix

n = 4096
p = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = mbpls_amd.get_context()
Ts, Y = planted(n, 1, 32, 0)
for label, c in (("centred data", 0.0), ("offset data (a constant score of 200 among 32 latent ones)", 200.0)):
    T2 = np.hstack((Ts, np.full((n, 1), c / np.sqrt(n))))           # the generator draws a loading per column for it
    X = DeviceMatrix(ctx, n, p)
    X.fill_latent(ctx.from_host(T2), 33, 0.5, 1234, 0)
    ctx.sync()
    times, plans = [], []
    for i in range(6):
        ctx.profile(i == 5); ctx.profile_reset()
        t0 = time.perf_counter()
        m = MBPLS(n_components=K).fit(X, Y)
        times.append(time.perf_counter() - t0)
        plans.append(m.fit_plan_.get("centred_in_registers"))
    prof = ctx.profile_get(); ctx.profile(False)
    print(f"{label}: fits {[round(1e3 * t, 1) for t in times]} ms  centred in registers {plans}  last fit: " +
          "  ".join(f"{k} {v['avg_ms']:.2f} ms x{v['launches']}" for k, v in prof.items()), flush=True)
    X.close()
