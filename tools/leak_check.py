"""Host memory left behind per fit, by allocation size class (an LD_PRELOAD counter of live malloc'd bytes):
    gcc -O2 -shared -fPIC -o /tmp/mshim.so tools/malloc_count/mshim.c -ldl
    LD_PRELOAD=/tmp/mshim.so python3 tools/leak_check.py [n] [general]
Found in round 4: the HIP runtime releases the command objects of a stream only when the STREAM is synchronised -- a copier
thread that waited for its copies through events left 2.2-3 KB behind per copy (csrc/copier.hip)."""
import os, sys, gc, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np
import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix
sh = ctypes.CDLL(os.environ.get("LD_PRELOAD", "/tmp/mshim.so").split(":")[0])
def dump():
    c = (ctypes.c_long * 64)(); b = (ctypes.c_long * 64)(); sh.mshim_dump(c, b); return list(c), list(b)
n, p, K = (int(sys.argv[1]) if len(sys.argv) > 1 else 4096), 125000, 10
if len(sys.argv) > 2 and sys.argv[2] == 'general':
    mbpls_amd.get_context().set_tunable('step_reg', 0)
ctx = mbpls_amd.get_context()
Ts, Y = planted(n, 1, 32, 0)
X = DeviceMatrix(ctx, n, p); X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0); ctx.sync()
for _ in range(30): MBPLS(n_components=K).fit(X, Y)
gc.collect(); c0, b0 = dump()
N = 200
for i in range(N): MBPLS(n_components=K).fit(X, Y)
gc.collect(); c1, b1 = dump()
for i in range(64):
    if c1[i] - c0[i] or b1[i] - b0[i]:
        print(f"size class 2^{i:2d}..: live count {c1[i] - c0[i]:+7d}  live bytes {b1[i] - b0[i]:+12d}  ({(b1[i] - b0[i]) / N / 1e3:+.1f} KB per fit)")
