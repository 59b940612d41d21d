"""Leave-one-out / k-fold cross_val_predict latency at example sizes, folds batched per pass over X (cv_batch: one
response with every method, several-response NIPALS) against the fold-by-fold loop: python tools/time_cv.py"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
warnings.simplefilter("ignore")
from mbpls_amd import MBPLS, cross_val_predict
from oracle import mbpls_oracle as orc

CASES = ((150, 3000, 5, 150, 1, ("NIPALS", "UNIPALS", "KERNEL", "SIMPLS")),
         (150, 3000, 5, 150, 3, ("NIPALS", "UNIPALS", "KERNEL")),
         (150, 3000, 5, 10, 1, ("NIPALS",)),
         (1000, 20000, 8, 10, 1, ("NIPALS", "UNIPALS")),
         (300, 100000, 5, 300, 1, ("NIPALS", "KERNEL")),
         (2048, 400000, 5, 16, 1, ("NIPALS",)))
for n, p, K, cv, q, methods in CASES:
    X, Y = orc.make_latent(n, p, q, seed=1)
    Xs = [np.ascontiguousarray(X[:, :p // 2]), np.ascontiguousarray(X[:, p // 2:])]
    for method in methods:
        res = {}
        for batched in (True, False):
            if not batched and cv >= 300 and method == "NIPALS":
                continue
            m = MBPLS(n_components=K, method=method, full_svd=True)
            cross_val_predict(m, Xs, Y, cv=min(cv, 20), batch_folds=batched)      # warm-up (pools, scratch)
            tt = []
            for _ in range(2):                                                    # the better of two (the first also sizes pools)
                t0 = time.perf_counter(); yhat = cross_val_predict(m, Xs, Y, cv=cv, batch_folds=batched); tt.append(time.perf_counter() - t0)
            t0, t1 = 0.0, min(tt)
            res[batched] = t1 - t0
            print(f"n={n} p={p} K={K} q={q} {method} cv={cv} {'folds batched' if batched else 'fold by fold '}: "
                  f"{1e3 * (t1 - t0):.1f} ms total, {1e3 * (t1 - t0) / cv:.3f} ms per fold, "
                  f"Q2 = {1 - ((Y - yhat) ** 2).sum() / ((Y - Y.mean(axis=0)) ** 2).sum():.4f}", flush=True)
        if len(res) == 2:
            print(f"    batched is {res[False] / res[True]:.1f}x the loop", flush=True)
