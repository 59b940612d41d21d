"""cProfile of one warm fit of the paper's protocol (tools/paper_protocol.py) on a resident matrix: where the HOST spends
a fit whose passes take microseconds.   python tools/profile_paper_host.py a|b method [randint|planted] [top]"""
import cProfile
import os
import pstats
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter("ignore")
import mbpls_amd
from mbpls_amd import MBPLS, DeviceMatrix
from tools.paper_protocol import make

shape, method = sys.argv[1], sys.argv[2]
kind = sys.argv[3] if len(sys.argv) > 3 else "planted"
top = int(sys.argv[4]) if len(sys.argv) > 4 else 30
N, P = (20_000, 1_000) if shape == "a" else (20_000, 20_000)
ctx = mbpls_amd.get_context()
blocks, Y = make(kind, N, P, 10)
dm = DeviceMatrix(ctx, N, P, [0, P // 2, P])
dm.upload(blocks[0], 0); dm.upload(blocks[1], P // 2)
ctx.sync()
for i in range(3):
    t0 = time.perf_counter()
    m = MBPLS(n_components=20, method=method, standardize=True, calc_all=False).fit(dm, Y)
    print(f"fit {i}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
pr = cProfile.Profile()
pr.enable()
m = MBPLS(n_components=20, method=method, standardize=True, calc_all=False).fit(dm, Y)
pr.disable()
print(m.fit_plan_.get("solver"), getattr(m, "nipals_iterations_", None))
print({k: round(1e3 * v, 2) for k, v in m.fit_timings_.items() if v > 2e-4})
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(top)
