"""One few-samples fit (n <= 1024: csrc/colpass_sub.hip) for rocprofv3 --stats / --pmc runs: python tools/few_one.py n p [fits]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix

n, p = int(sys.argv[1]), int(sys.argv[2])
fits = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ctx = mbpls_amd.get_context()
Ts, Y = planted(n, 1, 16, 0)
X = DeviceMatrix(ctx, n, p, [0, p // 3, p])
X.fill_latent(ctx.from_host(Ts), 16, 0.5, 1234, 0)
ctx.sync()
for _ in range(fits):
    m = MBPLS(n_components=5).fit(X, Y)
ctx.sync()
print(m.fit_plan_)
