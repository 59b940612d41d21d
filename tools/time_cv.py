"""Leave-one-out / k-fold cross_val_predict latency at example sizes, folds batched per pass over X (one-response
NIPALS, cv_batch) against the fold-by-fold loop: python tools/time_cv.py"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
warnings.simplefilter("ignore")
from mbpls_amd import MBPLS, cross_val_predict
from oracle import mbpls_oracle as orc

for n, p, K, cv in ((150, 3000, 5, 150), (150, 3000, 5, 10), (1000, 20000, 8, 10), (300, 100000, 5, 300), (2048, 400000, 5, 16)):
    X, Y = orc.make_latent(n, p, 1, seed=1)
    Xs = [np.ascontiguousarray(X[:, :p // 2]), np.ascontiguousarray(X[:, p // 2:])]
    for method, batched in (("NIPALS", True), ("NIPALS", False), ("KERNEL", False)):
        if (not batched and method == "NIPALS" and cv >= 300) or (method == "KERNEL" and n >= 2048):
            continue
        m = MBPLS(n_components=K, method=method)
        cross_val_predict(m, Xs, Y, cv=min(cv, 20), batch_folds=batched)      # warm-up (pools, scratch)
        t0 = time.perf_counter(); yhat = cross_val_predict(m, Xs, Y, cv=cv, batch_folds=batched); t1 = time.perf_counter()
        print(f"n={n} p={p} K={K} {method} cv={cv} {'folds batched' if batched else 'fold by fold '}: "
              f"{1e3 * (t1 - t0):.1f} ms total, {1e3 * (t1 - t0) / cv:.3f} ms per fold, "
              f"Q2 = {1 - ((Y - yhat) ** 2).sum() / ((Y - Y.mean()) ** 2).sum():.4f}", flush=True)
