"""Cross-validation folds batched per read of X (the reference's notebooks call
``cross_val_predict(MBPLS(k), X, y, cv=len(X))``: leave-one-out = n complete refits).

One-response NIPALS for F folds at once.  Fold f is a row mask m_f with its own column statistics mu_f, sigma_f; its
standardised, deflated matrix is never formed:

    Z_f = diag(m_f) (X - 1 mu_f^T) diag(1/sigma_f),      X_k,f = (I - Ts_f Ts_f^T) Z_f

(the rank-one deflations X <- X - ts p^T with p = X^T ts, mbpls.py:928/:969, are projections: the scores are
orthonormal), so the two X-sized products of a NIPALS iteration (:856, :875) for ALL folds are two many-right-hand-side
products with the one RAW matrix resident in HBM,

    X_k,f^T u_f   = (X^T u'_f - mu_f (1^T u'_f)) / sigma_f,    u'_f = u_f - Ts_f (Ts_f^T u_f)            [xt_v, F columns]
    X_k,f,b w_fb  = (I - Ts_f Ts_f^T) [ m_f o (X_b (w_fb / sigma_fb) - mu_fb^T (w_fb / sigma_fb)) ]       [x_w, F columns]

(u_f, ts_fl vanish on the held-out rows).  Between the two products, the element-wise p x F algebra and its block
sums run on the device (csrc/cv_batch.hip, mbpls_hip_cv_weights); the n x F x K algebra -- the deflation is far cheaper
on the n side than through the loadings -- stays on the host like the n x B algebra of a single fit.  Only n x F arrays
and a few scalars per fold cross PCIe per sweep; the finished weight / loading columns are fetched once per batch.  Per-fold statistics come from the full-data centred sums minus the held-out rows' share (corrected
two-pass form, sklearn's constant-column rule).  A batch of up to 256 folds costs 5 K product launches instead of that many
complete fits of ~110 launches each -- what leave-one-out on a small data set (n = 150, p = 3000, K = 5: the
reference's notebooks) spends its time on.

Used by ``model_selection.cross_val_predict`` for method='NIPALS', one response, the 2-norm stop rule, complete data, a
host copy of X (per-fold statistics, held-out rows), p <= 16384 and p x K small enough for at least 16 folds per batch
(32 MB of fetched columns) -- the regime of the reference's notebooks, where a fit is all launches.  Wider matrices take
the fold-by-fold loop: their passes hide the launches, and the per-fold statistics / end-of-fit algebra of a batch are
still host numpy here (on the device they would make this the faster path at any width: next step).  Same numbers as a loop of
fits (tests/test_gpu_parity.py) to ~1e-12.
"""
from __future__ import annotations

import warnings

import numpy as np

from . import solvers
from .engine import HipEngine

_EPS = np.finfo(np.float64).eps
MAX_FOLDS = 256                   # folds per batch (the matrix-core products read X once per 64 right-hand sides)
MAX_STACK_BYTES = 32 << 20        # the two p x F x K stacks of finished columns are fetched once per batch
MAX_P = 16384                     # (per-fold statistics and the end-of-fit algebra of a batch are host numpy on p x F arrays)
MAX_HOST_BYTES = 1 << 30          # host copy of X (statistics of the folds, held-out rows)


def eligible(estimator, q, n, p, n_folds):
    K = int(estimator.n_components)
    return (estimator.method == "NIPALS" and estimator.sparse_data is not True and q == 1 and n_folds >= 2
            and estimator.nipals_convergence_norm in (2, -2, "fro", "nuc")
            and p <= MAX_P and 8 * p * K * min(n_folds, 16) <= MAX_STACK_BYTES)


def predict_folds(estimator, X_host, Xdev, ctx, comm, y, splits, batch=None):
    """out-of-fold predictions (n x 1) of a one-response NIPALS model for every (train, test) pair of ``splits``;
    X_host: the n x p host matrix (blocks side by side), Xdev: the same, raw, resident in HBM"""
    n, p = X_host.shape
    K = int(estimator.n_components)
    if batch is None:                                # as many folds per read of X as the fetched stacks allow
        batch = int(max(1, min(MAX_FOLDS, MAX_STACK_BYTES // (8 * p * K))))
    std = bool(estimator.standardize)
    tol = float(estimator.max_tol)
    offs = list(Xdev.block_offsets)
    y = np.asarray(y, dtype=np.float64).reshape(n)
    out = np.full((n, 1), np.nan)
    mu0 = X_host.mean(axis=0) if std else np.zeros(p)
    Xc = X_host - mu0 if std else X_host
    Xc2 = Xc * Xc if std else None
    eng = HipEngine(ctx=ctx, comm=comm)
    eng.load_device(Xdev, copy=False)                # raw matrix, only ever read
    from . import runtime
    try:
        with runtime.host_threads():                 # BLAS pools sized to the CPU quota, not to the visible CPUs
            for s0 in range(0, len(splits), batch):
                _one_batch(eng, X_host, Xc, Xc2, mu0, y, splits[s0:s0 + batch], out, K, std, tol, offs)
    finally:
        eng.X = None                                 # the caller owns the matrix
        eng.close()
    return out


def _fold_stats(X_host, Xc, Xc2, mu0, M, std):
    """mean / scale (p x F) of the training rows of every fold: full-data centred sums minus the held-out rows' share
    (corrected two-pass form), sklearn's constant-column rule; columns whose downdated variance is lost to cancellation
    (a column that is constant on one fold's training rows) are taken again from the rows themselves"""
    n, p = X_host.shape
    F = M.shape[1]
    if not std:
        return np.zeros((p, F)), np.ones((p, F))
    n_f = M.sum(axis=0)
    H = 1.0 - M
    hsum = Xc.T.dot(H)                               # sums of the held-out rows, centred at the full-data mean
    dmean = -hsum / n_f                              # mean_f - mu0
    ss0 = Xc2.sum(axis=0)
    ssq = ss0[:, None] - Xc2.T.dot(H) - n_f * dmean * dmean
    mean = mu0[:, None] + dmean
    risky = ssq <= 1e-9 * ss0[:, None]               # cancellation: what is left is rounding of the full-data sums
    for j, f in zip(*np.nonzero(risky)):
        col = X_host[M[:, f] != 0.0, j]
        mean[j, f] = col.mean()
        d = col - mean[j, f]
        ssq[j, f] = np.sum(d * d) - np.sum(d) ** 2 / n_f[f]
    var = np.maximum(ssq, 0.0) / n_f
    bound = n_f * _EPS * var + (n_f * mean * _EPS) ** 2
    scale = np.sqrt(var)
    scale[var <= bound] = 1.0                        # sklearn: a constant column keeps scale 1
    return mean, scale


def _one_batch(eng, X_host, Xc, Xc2, mu0, y, part, out, K, std, tol, offs):
    from . import _capi
    n, p = X_host.shape
    F, B = len(part), len(offs) - 1
    ctx, lib = eng.ctx, eng.ctx.lib
    M = np.zeros((n, F))
    for f, (tr, te) in enumerate(part):
        M[np.asarray(tr), f] = 1.0
    n_f = M.sum(axis=0)
    mu, sig = _fold_stats(X_host, Xc, Xc2, mu0, M, std)
    rs = 1.0 / sig
    if std:                                          # ColumnScaler.fit on the training responses of every fold
        ymu = (y[:, None] * M).sum(axis=0) / n_f
        d = (y[:, None] - ymu) * M
        var = ((d * d).sum(axis=0) - d.sum(axis=0) ** 2 / n_f) / n_f
        ysig = np.where(var <= n_f * _EPS * var + (n_f * ymu * _EPS) ** 2, 1.0, np.sqrt(var))
    else:
        ymu, ysig = np.zeros(F), np.ones(F)
    Ys = (y[:, None] - ymu) / ysig * M               # n x F, zero on the held-out rows
    # everything p-sized stays in HBM: the folds' statistics, the product X^T U, weights, what X is multiplied with,
    # and the K finished weight / loading columns of every fold (component-major: block k holds the p x F columns)
    dmu, drs = ctx.from_host(mu), ctx.from_host(rs)
    dsu, dal, dsums = ctx.empty(F), ctx.empty(F), ctx.empty(B * F * 2)
    G, Cs, dg = eng.slot("cvG", F, zero=False), eng.slot("cvC", F, zero=False), ctx.empty((p, F))
    Wst, Pst = eng.slot("cvW", F * K, zero=False), eng.slot("cvP", F * K, zero=False)
    pf8 = 8 * p * F

    def weights(U, alpha):
        """g = alpha (X^T U - mu su) rs, C = g rs on the device for every fold; returns the block sums (B x F x 2)"""
        eng.xt_v(U, 1.0, "cvG", 0)                                           # X^T U (p x F), raw X
        dsu.set(U.sum(axis=0)); dal.set(alpha)
        _capi.check(lib.mbpls_hip_cv_weights(eng.X.handle, F, G.ptr, dmu.ptr, drs.ptr, dsu.ptr, dal.ptr, dg.ptr, Cs.ptr,
                                             dsums.ptr))

    def keep(stack, k, cols):
        """finished columns of component k (all folds, or the folds in ``cols``) from dg into a stack"""
        if cols is None:
            _capi.check(lib.mbpls_hip_memcpy_d2d(ctx.handle, stack.ptr + k * pf8, dg.ptr, pf8))
        else:
            for f in cols:
                _capi.check(lib.mbpls_hip_memcpy_d2d(ctx.handle, stack.ptr + k * pf8 + 8 * p * int(f), dg.ptr + 8 * p * int(f),
                                                     8 * p))

    Ts = np.zeros((F, n, K))                                                 # fold-major: the deflation is n x K algebra
    nu_all = np.ones((B, F, K)); V = np.zeros((F, K))

    def deflate(A, k):
        """(I - Ts_f Ts_f^T) a_f for every fold (columns of the n x F array A): n x k algebra per fold"""
        if not k:
            return A
        T = Ts[:, :, :k]                                                     # F x n x k
        c = np.matmul(T.transpose(0, 2, 1), A.T[:, :, None])                 # F x k x 1
        return A - np.matmul(T, c)[:, :, 0].T

    def sweep(U, alpha, k):
        """block sums of squares of g = alpha X_k^T u (left on the device) and t~_b = X_k,b g_b -- for every fold"""
        weights(deflate(U, k), alpha)
        XC = eng.x_w_slot("cvC", 0, F, per_block=True)                       # per block: X_b C_b (n x F)
        sums = dsums.get().reshape(B, F, 2)
        t = np.empty((B, n, F))
        for b in range(B):
            t[b] = deflate((XC[b] - sums[b, :, 1]) * M, k)
        return sums[:, :, 0], t

    try:
        for k in range(K):
            u = Ys.copy()
            active = np.ones(F, dtype=bool)
            ts_old = np.zeros((n, F))
            nu_fin = np.ones((B, F)); ts_fin = np.zeros((n, F)); v_fin = np.zeros(F)
            run = 1
            while active.any():
                uu = (u * u).sum(axis=0)
                ssq, t = sweep(u, 1.0 / uu, k)                               # (:856, :875)
                nu = np.sqrt(ssq)                                            # ||w~_b|| (:860)
                Tm = t / nu[:, None, :]                                      # t_b = X_b w_b (:875-877)
                sw = (Tm * u).sum(axis=1) / uu                               # B x F (:879)
                sw /= np.sqrt((sw * sw).sum(axis=0))
                ts = (Tm * sw[:, None, :]).sum(axis=0)                       # (:882)
                ts /= np.sqrt((ts * ts).sum(axis=0))
                diff = np.sqrt(((ts_old - ts) ** 2).sum(axis=0)) if run > 1 else np.full(F, np.inf)   # (:887)
                v = (Ys * ts).sum(axis=0) / (ts * ts).sum(axis=0)            # (:899)
                un = Ys * (v / (v * v))                                      # (:911)
                un /= np.sqrt((un * un).sum(axis=0))
                floor_hit = run >= solvers.NIPALS_Q1_NOISE_FLOOR_ITER
                done = active & ((diff <= tol) | floor_hit)
                if floor_hit and (active & (diff > tol)).any():
                    warnings.warn(f"NIPALS component {k}: ||ts_old - ts|| stays above max_tol={tol} in "
                                  f"{int((active & (diff > tol)).sum())} fold(s) (rounding noise of an exhausted "
                                  f"response); stopped after {run} iterations.")
                if done.any():                                               # this iteration's g is their W~[:, k]
                    keep(Wst, k, None if done.all() else np.nonzero(done)[0])
                    nu_fin[:, done] = nu[:, done]; ts_fin[:, done] = ts[:, done]; v_fin[done] = v[done]
                ts_old[:, active] = ts[:, active]
                u[:, active] = un[:, active]
                active = active & ~done
                run += 1
            # loadings p = X_k^T ts (:928; ts is unit length and orthogonal to the earlier scores up to rounding)
            weights(deflate(ts_fin, k), np.ones(F))
            keep(Pst, k, None)
            Ts[:, :, k] = ts_fin.T; nu_all[:, :, k] = nu_fin; V[:, k] = v_fin
        # every fold: R = W (P^T W)^+, beta = R V^T (:986-989) -- stacked K x K algebra on the host -- and the
        # predictions of its held-out rows (:1369-1408)
        Wh, Ph = Wst.get(), Pst.get()                                        # p x (K F): column k F + f
        D = 1.0 / np.sqrt((nu_all ** 2).sum(axis=0))                         # F x K: 1 / ||W~[:, k]||
        Cm = np.empty((F, K, K))
        for f in range(F):                           # P_f^T W_f on column-strided views (leading dimension p F: no copies)
            Cm[f] = Ph[:, f::F].T.dot(Wh[:, f::F])
        Mv = np.matmul(D[:, :, None] * np.linalg.pinv(Cm * D[:, None, :]), V[:, :, None])[:, :, 0]   # F x K
        for f, (tr, te) in enumerate(part):
            te = np.asarray(te)
            beta = Wh[:, f::F].dot(Mv[f])                                    # R_f V_f^T (p)
            Zt = (X_host[te] - mu[:, f]) * rs[:, f]
            out[te, 0] = Zt.dot(beta) * ysig[f] + ymu[f]
    finally:
        for a_ in (dmu, drs, dsu, dal, dsums, dg):
            a_.free()
