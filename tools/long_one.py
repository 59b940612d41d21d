"""One tall fit (NIPALS, K = 3, cluster passes) for rocprofv3 --pmc / --stats runs: python tools/long_one.py n p [fits]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix

n, p = int(sys.argv[1]), int(sys.argv[2])
fits = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ctx = mbpls_amd.get_context()
Ts, Y = planted(n, 1, 32, 0)
X = DeviceMatrix(ctx, n, p)
X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0)
ctx.sync()
for _ in range(fits):
    m = MBPLS(n_components=3).fit(X, Y)
ctx.sync()
print(m.fit_plan_, m.nipals_iterations_, "timeouts", ctx.cluster_status())
