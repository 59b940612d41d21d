import sys, time
sys.path.insert(0, "/root/repo")
import mbpls_amd
from mbpls_amd import MBPLS, DeviceMatrix
from bench import planted
ctx = mbpls_amd.get_context()
for n in (4500, 5000, 6000, 7000, 8192):
    p = int(8e9 / 8 / n)
    Ts, Y = planted(n, 1, 16, 0)
    X = DeviceMatrix(ctx, n, p, [0, p // 3, p])
    X.fill_latent(ctx.from_host(Ts), 16, 0.5, 1234, 0)
    ctx.sync()
    for pf, rpt in ((2, 0), (3, 0), (1, 0), (2, 8)):
        ctx.set_tunable("prefetch", pf)
        ctx.set_tunable("rows_per_thread", rpt)
        m = MBPLS(n_components=5)
        try:
            for rep in range(3):
                ctx.profile(rep == 2); ctx.profile_reset()
                m.fit(X, Y); ctx.sync()
            prof = ctx.profile_get(); ctx.profile(False)
            print(n, "prefetch", pf, "rows_per_thread", rpt, "  ".join(f"{k}: {v['GBps']:.0f}" for k, v in prof.items()), flush=True)
        except Exception as e:
            print(n, pf, rpt, "failed", str(e)[:100])
    ctx.set_tunable("prefetch", 2); ctx.set_tunable("rows_per_thread", 0)
    X.close()
