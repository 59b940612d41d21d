"""Rows per thread for 4096 < n <= 8192 (8 GB matrices): the default choice against R = 10 and R = 16:
python tools/ab_rows.py [n,n,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbpls_amd
from mbpls_amd import MBPLS, DeviceMatrix
from bench import planted
ctx = mbpls_amd.get_context()
sizes = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4500, 5000, 5500, 6000, 7000, 8192]
for n in sizes:
    p = int(8e9 / 8 / n)
    Ts, Y = planted(n, 1, 16, 0)
    X = DeviceMatrix(ctx, n, p, [0, p // 3, p])
    X.fill_latent(ctx.from_host(Ts), 16, 0.5, 1234, 0)
    ctx.sync()
    for rpt in (0, 10, 16):
        if rpt and n > 512 * rpt:
            continue
        ctx.set_tunable("rows_per_thread", rpt)
        try:
            for rep in range(3):
                ctx.profile(rep == 2); ctx.profile_reset()
                t0 = time.perf_counter(); MBPLS(n_components=5).fit(X, Y); ctx.sync(); t1 = time.perf_counter()
            prof = ctx.profile_get(); ctx.profile(False)
            print(f"n={n} rows_per_thread={rpt or 'default'}: fit {1e3 * (t1 - t0):.2f} ms  " +
                  "  ".join(f"{k}: {v['GBps']:.0f}" for k, v in prof.items()), flush=True)
        except Exception as e:
            ctx.profile(False)
            print(n, rpt, "failed", str(e)[:100])
    ctx.set_tunable("rows_per_thread", 0)
    X.close()
