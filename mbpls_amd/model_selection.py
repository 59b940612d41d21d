"""Cross-validation on a matrix that stays resident in HBM.

The reference's users call ``sklearn.model_selection.cross_val_predict(MBPLS(k), X, y, cv=len(X))``
(examples/real_world_applications/*.ipynb): leave-one-out = n complete refits, each on a fresh row
slice of X.  Here X is uploaded once and only ever read:

* batched folds (``cv_batch``): up to 256 folds are fitted per pass over X -- every read of the raw matrix serves all
  of them through many-right-hand-side products, the folds' column statistics and deflations are applied
  algebraically.  Covers one response with every method (they fit the same model then) and several responses with
  NIPALS and with the p > n branches of UNIPALS / KERNEL -- also under column shards (every rank holds its columns of
  every block; the n x F products and the small sums are all-reduced, every rank returns all predictions);
* fold by fold (everything else: several-response SIMPLS, the n >= p branches, missing values): a fold is a
  *row mask* handed to the standardise pass, which writes zero rows for the held-out samples into the working copy and
  takes the column statistics over the training samples only; predictions for the held-out rows come from one ``x_w``
  pass over the same resident matrix.  No re-upload, no host slicing.
"""
from __future__ import annotations

import numpy as np

from . import _capi, runtime
from .mbpls import MBPLS, _is_block_list


def _resident(X):
    """Upload host blocks once (feature-major, side by side); pass a DeviceMatrix through.
    Returns (matrix, owned, host blocks or None)."""
    if isinstance(X, _capi.DeviceMatrix):
        return X, False, None
    blocks = X if _is_block_list(X) else [X]
    blocks = [np.asarray(b, dtype=np.float64) for b in blocks]
    n = blocks[0].shape[0]
    offs = np.concatenate(([0], np.cumsum([b.shape[1] for b in blocks]))).astype(np.int64)
    ctx = runtime.get_context()
    dm = _capi.DeviceMatrix(ctx, n, int(offs[-1]), offs)
    for b, o in zip(blocks, offs[:-1]):
        dm.upload(b, int(o))
    return dm, True, blocks


def cross_val_predict(estimator, X, y, cv=5, groups=None, batch_folds=True):
    """Out-of-fold predictions of ``estimator`` (an ``MBPLS``), like sklearn's ``cross_val_predict``."""
    with runtime.get_context().lock:          # (one caller at a time per context, as for fit / predict / transform)
        return _cross_val_predict(estimator, X, y, cv, groups, batch_folds)


def _cross_val_predict(estimator, X, y, cv, groups, batch_folds):
    """Out-of-fold predictions of ``estimator`` (an ``MBPLS``), like sklearn's ``cross_val_predict``.

    cv: int (KFold without shuffling, as sklearn does for regressors), a CV splitter, or an iterable of
    (train, test) index pairs.  Returns an (n, q) array (``predict``'s shape).

    One-response models of every method and several-response NIPALS / UNIPALS / KERNEL (p > n) models are fitted up to
    256 folds at a time (as many as device memory holds), every pass over X serving all of them (``cv_batch``;
    ``batch_folds=False`` forces the fold-by-fold loop, which everything else takes)."""
    if not isinstance(estimator, MBPLS):
        raise TypeError("cross_val_predict of mbpls_amd works on mbpls_amd.MBPLS estimators")
    y = np.asarray(y, dtype=np.float64)
    y2 = y.reshape(-1, 1) if y.ndim == 1 else y
    n = y2.shape[0]
    if hasattr(cv, "split") or isinstance(cv, (int, np.integer)) or cv is None:
        from sklearn.model_selection import check_cv
        splits = list(check_cv(cv, y2, classifier=False).split(np.zeros((n, 1)), y2, groups))
    else:
        splits = [(np.asarray(tr), np.asarray(te)) for tr, te in cv]
    # sklearn's cross_val_predict: every sample is predicted exactly once ("only works for partitions")
    tested = np.concatenate([np.asarray(te).ravel() for _, te in splits]) if splits else np.empty(0, dtype=int)
    if tested.size != n or not np.array_equal(np.sort(tested), np.arange(n)):
        raise ValueError("cross_val_predict only works for partitions")
    Xdev, owned, host_blocks = _resident(X)
    try:
        if Xdev.n != n:
            raise ValueError(f"Found input variables with inconsistent numbers of samples: [{Xdev.n}, {n}]")
        from . import cv_batch
        if batch_folds and cv_batch.eligible(estimator, y2.shape[1], n, Xdev.p, len(splits), runtime.get_context(),
                                             runtime.get_comm()):
            try:
                return cv_batch.predict_folds(estimator, Xdev, runtime.get_context(), runtime.get_comm(), y2, splits)
            except cv_batch.DegenerateFold:
                pass                     # a fold broke down: the loop below treats every fold like a single fit
        out = np.full((n, y2.shape[1]), np.nan)
        pending = _HeldOutPredictions(Xdev, out, sparse=estimator.sparse_data is True)
        for train, test in splits:
            est = MBPLS(**{**estimator.get_params(), "copy": True})
            mask = np.zeros(n, dtype=bool)
            mask[train] = True
            est._fit(Xdev, y2, mask)
            pending.add(est, test)
        pending.flush()
        return out
    finally:
        if owned:
            Xdev.close()


class _HeldOutPredictions:
    """Predictions of the fold-by-fold loop without one full pass over X PER FOLD (``est.predict(Xdev)[test]`` computes
    all n rows to keep a few).  predict is (X_std beta) sigma_y + mu_y (mbpls.py:1386) with the fold's own scalers:
        X_std,f beta_f = X (beta_f / sigma_f) - mu_f^T (beta_f / sigma_f)
    -- a product of the RAW resident matrix with p x q columns and a constant.  The folds' columns are collected and
    multiplied together: one pass over X per up to 64 fold-responses instead of one per fold; every fold keeps its
    held-out rows.  Missing values (``sparse_data=True``: an absent entry contributes nothing, :1376-1384): the constant
    becomes a per-row sum over the PRESENT columns -- a second, indicator product with the columns mu_f o (beta_f /
    sigma_f).  Under column shards the products and the constants are summed over the ranks."""

    MAX_COLUMNS = 64

    def __init__(self, Xdev, out, sparse):
        self.X, self.out, self.sparse = Xdev, out, bool(sparse)
        self.q = out.shape[1]
        ctx = runtime.get_context()
        budget = (ctx.info()["free_bytes"] + ctx._pool_bytes) // 8
        self.cols = int(max(self.q, min(self.MAX_COLUMNS, budget // (8 * max(1, Xdev.p) * (2 if sparse else 1)))))
        self.items = []
        self.used = 0

    def add(self, est, test):
        if self.used + self.q > self.cols:
            self.flush()
        self.items.append((est, np.asarray(test)))
        self.used += self.q

    def flush(self):
        if not self.items:
            return
        from .engine import HipEngine
        X, q, p = self.X, self.q, self.X.p
        eng = HipEngine(runtime.get_context(), runtime.get_comm())
        ctx = eng.ctx
        try:
            nc = q * len(self.items)
            W = np.empty((p, nc), order="F")
            M = np.empty((p, nc), order="F") if self.sparse else None
            const = np.zeros((1, nc))
            for f, (est, _) in enumerate(self.items):
                cols = slice(f * q, (f + 1) * q)
                if est.standardize:
                    mean = np.concatenate([sc.mean_ for sc in est.x_scalers_])
                    scale = np.concatenate([sc.scale_ for sc in est.x_scalers_])
                    W[:, cols] = est.beta_ / scale[:, None]
                    if self.sparse:
                        M[:, cols] = W[:, cols] * mean[:, None]
                    else:
                        const[0, cols] = mean.dot(W[:, cols])
                else:
                    W[:, cols] = est.beta_
                    if self.sparse:
                        M[:, cols] = 0.0
            dW = ctx.from_host(W)
            Yh = eng._xw(dW, nc, False, None, None, mat=X, nan="skip" if self.sparse else None)
            dW.free()
            if self.sparse and any(est.standardize for est, _ in self.items):
                dM = ctx.from_host(M)
                Yh = Yh - eng._xw(dM, nc, False, None, None, mat=X, nan="indicator")
                dM.free()
            elif not self.sparse:
                Yh = Yh - eng.comm.allreduce_host(const)
            for f, (est, test) in enumerate(self.items):
                y = Yh[test, f * q:(f + 1) * q]
                if not np.all(np.isfinite(y)):
                    raise ValueError("Input X contains NaN, infinity or a value too large for dtype('float64').")
                self.out[test] = est.y_scaler_.inverse_transform(y) if est.standardize else y
        finally:
            eng.close()
            self.items, self.used = [], 0
