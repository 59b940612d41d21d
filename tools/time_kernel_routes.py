"""KERNEL p > n on its two routes -- the association matrix X X^T on the f64 matrix cores (MBPLS_KERNEL_GRAM=1) and the
maintained thin products, one read-only pass per component (=0) -- over shapes from the reference's own examples to
BASELINE config[3]; beside each pair the route the cost model (solvers._kernel_on_products) would take.
    python tools/time_kernel_routes.py [small|big|c4]"""
import json
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

warnings.simplefilter("ignore")
import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix, solvers


def one(n, p, q, K, blocks=1, reps=3):
    ctx = mbpls_amd.get_context()
    Ts, Y = planted(n, q, 32, seed=0)
    offs = [int(round(b * p / blocks)) for b in range(blocks + 1)]
    X = DeviceMatrix(ctx, n, p, offs)
    dT = ctx.from_host(Ts)
    X.fill_latent(dT, 32, 0.5, 1234, 0)
    ctx.sync()
    dT.free()
    out = {"n": n, "p": p, "q": q, "K": K, "blocks": blocks}
    models = {}
    for route, env in (("gram", "1"), ("products", "0")):
        os.environ["MBPLS_KERNEL_GRAM"] = env
        ts = []
        for _ in range(reps + 1):
            m = MBPLS(n_components=K, method="KERNEL")
            t0 = time.perf_counter(); m.fit(X, Y); ts.append(time.perf_counter() - t0)
        models[route] = m
        out[route + "_s"] = round(sorted(ts[1:])[len(ts[1:]) // 2], 5)
        out[route + "_phases_ms"] = {k: round(1e3 * v, 2) for k, v in m.fit_timings_.items()}
    os.environ.pop("MBPLS_KERNEL_GRAM")
    a, b = models["gram"], models["products"]
    s = np.sign(np.sum(a.Ts_ * b.Ts_, axis=0))
    out["max_rel_diff"] = float(max(np.max(np.abs(a.Ts_ * s - b.Ts_)) / np.max(np.abs(a.Ts_)),
                                    np.max(np.abs(a.beta_ - b.beta_)) / np.max(np.abs(a.beta_))))

    class _E:           # what the cost model looks at
        sparse, rowmask = False, None
        opts = mbpls_amd.plan.FitOptions()
    out["model_picks"] = "products" if solvers._kernel_on_products(_E, n, K, q, p) else "gram"
    out["faster"] = "products" if out["products_s"] < out["gram_s"] else "gram"
    print(json.dumps(out), flush=True)
    X.close()
    ctx.trim_pool()


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "small"
    if which == "small":
        for n, p, q, K in ((40, 200, 1, 2), (150, 3000, 1, 5), (150, 3000, 3, 5), (300, 20000, 1, 3), (300, 20000, 1, 12),
                           (600, 100000, 2, 4), (1000, 20000, 1, 8), (1000, 200000, 1, 20), (2048, 100000, 1, 5),
                           (2048, 100000, 4, 30)):
            one(n, p, q, K, blocks=2)
    elif which == "big":
        for n, p, q, K in ((4096, 1_000_000, 1, 10), (4096, 1_000_000, 10, 10), (8192, 500_000, 1, 20),
                           (8192, 500_000, 1, 100), (20000, 100_000, 1, 10)):
            one(n, p, q, K, blocks=1, reps=2)
    else:
        one(8192, 2_000_000, 1, 20, reps=2)
