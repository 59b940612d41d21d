"""Cluster passes (long columns, n > 8192): rows of a column held by one workgroup of a cluster ("cluster_slab_rows") --
8192 rows are eight waves at 190-256 VGPRs, ONE workgroup per CU; 4096 rows are four waves, two workgroups per CU and
twice as many members per cluster.   python tools/tune_cluster_slab.py [GB] n [n ...]"""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter("ignore")
import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
ns = [int(a) for a in sys.argv[2:]] or [16384, 20000, 65536, 200000]
ctx = mbpls_amd.get_context()
for n in ns:
    p = max(64, int(gb * 1e9 / 8 / n))
    Ts, Y = planted(n, 1, 32, 0)
    X = DeviceMatrix(ctx, n, p)
    X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0)
    ctx.sync()
    for slab in (8192, 6144, 4096, 2048):
        ctx.set_tunable("cluster_slab_rows", slab)
        for method in ("NIPALS", "UNIPALS"):
            m = MBPLS(n_components=4, method=method)
            for rep in range(3):
                ctx.profile(rep == 2); ctx.profile_reset()
                t0 = time.perf_counter(); m.fit(X, Y); ctx.sync(); t1 = time.perf_counter()
            prof = ctx.profile_get(); ctx.profile(False)
            print(f"n={n} p={p} slab {slab} {method}: fit {1e3 * (t1 - t0):.2f} ms  " +
                  "  ".join(f"{k}: {v['GBps']:.0f} GB/s x{v['launches']}" for k, v in prof.items())
                  + f"  [{m.fit_plan_['kernels'][:40]}] timeouts {ctx.cluster_status()}", flush=True)
    ctx.set_tunable("cluster_slab_rows", 8192)
    X.close()
