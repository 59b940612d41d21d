"""Run the non-headline BASELINE configs on one GPU and print per-kernel HIP-event profiles.

    python tools/bench_configs.py c3            # UNIPALS, 3 blocks [300k,300k,400k], n=4096, K=10
    python tools/bench_configs.py c4 [p]        # KERNEL, n=8192, p=2,000,000 (or p), K=20
    python tools/bench_configs.py simpls|nipals_q10|predict
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix

F64_PEAK_TF = 78.6


def run(name, n, p, K, q, method, offs, copy, calc_all=True, steps=2):
    ctx = mbpls_amd.get_context()
    Ts, Y = planted(n, q, 32, seed=0)
    X = DeviceMatrix(ctx, n, p, offs)
    dT = ctx.from_host(Ts)
    X.fill_latent(dT, 32, 0.5, 1234, 0)
    ctx.sync()
    times, models = [], []
    for s in range(steps + 1):
        if not copy and s > 0:
            X.fill_latent(dT, 32, 0.5, 1234, 0)
            ctx.sync()
        if s == 1:
            ctx.profile(True); ctx.profile_reset()
        m = MBPLS(n_components=K, method=method, calc_all=calc_all, copy=copy)     # a fresh estimator per fit, as bench.py
        models.append(m)
        t0 = time.perf_counter()
        m.fit(X, Y)
        ctx.sync()
        times.append(time.perf_counter() - t0)
    prof = ctx.profile_get(); ctx.profile(False)
    out = {"config": name, "n": n, "p": p, "K": K, "q": q, "method": method, "blocks": len(offs) - 1,
           "fit_s": [round(t, 4) for t in times], "host_phases_ms": {k: round(1e3 * v, 1) for k, v in m.fit_timings_.items()},
           "kernels": {k: {"launches": v["launches"] // steps, "avg_ms": round(v["avg_ms"], 3), "GBps": round(v["GBps"], 1)}
                       for k, v in prof.items()}}
    if "gram" in prof:
        fl = float(n) * (n + 1) * p                     # all Gram launches of one fit together cover the p columns once
        out["gram_TFLOPs"] = round(fl / (prof["gram"]["total_ms"] / steps * 1e-3) / 1e12, 2)
        out["gram_frac_of_f64_peak"] = round(out["gram_TFLOPs"] / F64_PEAK_TF, 3)
    if hasattr(m, "nipals_iterations_"):
        out["nipals_iterations"] = m.nipals_iterations_
    print(json.dumps(out), flush=True)
    X.close()
    return m


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "c3"
    if which == "c3":
        run("C3 UNIPALS multiblock", 4096, 1_000_000, 10, 1, "UNIPALS", [0, 300_000, 600_000, 1_000_000], True)
    elif which == "c4":
        p = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
        # default copy=True: statistics-only preparation, the products standardise on the fly (no second 131 GB buffer)
        # the default route at this shape (maintained thin products: K + 1 read-only passes), then the 1.34e14-flop Gram
        run("C4 KERNEL (maintained products, default)", 8192, p, 20, 1, "KERNEL", [0, p], True)
        os.environ["MBPLS_KERNEL_GRAM"] = "1"
        run("C4 KERNEL (f64 MFMA Gram, MBPLS_KERNEL_GRAM=1)", 8192, p, 20, 1, "KERNEL", [0, p],
            os.environ.get("MBPLS_IMPLICIT_STD", "1") != "0")
        os.environ.pop("MBPLS_KERNEL_GRAM")
    elif which == "simpls":
        run("SIMPLS", 4096, 1_000_000, 10, 1, "SIMPLS", [0, 1_000_000], True)
    elif which == "nipals_q10":
        run("NIPALS q=10 (paper protocol)", 4096, 1_000_000, 10, 10, "NIPALS", [0, 500_000, 1_000_000], True, steps=1)
    elif which == "unipals_q10":
        run("C3 UNIPALS multiblock q=10", 4096, 1_000_000, 10, 10, "UNIPALS", [0, 300_000, 600_000, 1_000_000], True)
    elif which == "nipals_mb":
        run("NIPALS 4 blocks", 4096, 1_000_000, 10, 1, "NIPALS", [0, 250_000, 500_000, 750_000, 1_000_000], True)
