"""One missing-value fit on long columns (two streaming phases, csrc/colpass_nan_long.hip) for rocprofv3 --stats / --pmc
runs:   python tools/sparse_long_one.py n p [fits] [two_phase=1|0]"""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

warnings.simplefilter("ignore")
import mbpls_amd
from mbpls_amd import MBPLS, DeviceMatrix

n, p = int(sys.argv[1]), int(sys.argv[2])
fits = int(sys.argv[3]) if len(sys.argv) > 3 else 1
two_phase = int(sys.argv[4]) if len(sys.argv) > 4 else 1
ctx = mbpls_amd.get_context()
ctx.set_tunable("nan_long_two_phase", two_phase)
rng = np.random.RandomState(0)
ts = rng.randn(n, 4)
X = ts.dot(rng.randn(4, p)) + 0.5 * rng.randn(n, p)
Y = ts[:, :1] + 0.1 * rng.randn(n, 1)
X[rng.rand(n, p) < 0.01] = np.nan
D = DeviceMatrix(ctx, n, p)
D.upload(X, 0)
for _ in range(fits):
    m = MBPLS(n_components=3, sparse_data=True).fit(D, Y)
ctx.sync()
print(m.fit_plan_, m.nipals_iterations_)
