"""Three C2 fits for a rocprofv3 --kernel-trace timeline (GPU idle gaps between the passes of one fit)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix

n, p, K = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 10
ctx = mbpls_amd.get_context()
if len(sys.argv) > 2 and sys.argv[2] == "rccl":       # the sharded code path with ONE rank: all-reduce behind every pass
    mbpls_amd.set_comm(mbpls_amd.RcclComm(ctx, 0, 1, lambda payload: payload))
Ts, Y = planted(n, 1, 32, 0)
X = DeviceMatrix(ctx, n, p)
X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0)
ctx.sync()
m = MBPLS(n_components=K)
for i in range(3):
    t0 = time.perf_counter(); m.fit(X, Y); ctx.sync()
    print(f"fit {i}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
    time.sleep(0.2)
