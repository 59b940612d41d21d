"""Cross-validation folds batched per read of X (the reference's notebooks call
``cross_val_predict(MBPLS(k), X, y, cv=len(X))``: leave-one-out = n complete refits).

One-response NIPALS for F folds at once.  Fold f is a row mask m_f with its own column statistics mu_f, sigma_f; its
standardised, deflated matrix is never formed:

    Z_f = diag(m_f) (X - 1 mu_f^T) diag(1/sigma_f),      X_k,f = (I - Ts_f Ts_f^T) Z_f

(the rank-one deflations X <- X - ts p^T with p = X^T ts, mbpls.py:928/:969, are projections: the scores are
orthonormal), so the two X-sized products of a NIPALS iteration (:856, :875) for ALL folds are two many-right-hand-side
products with the one RAW matrix resident in HBM (matrix cores, X read once per 64 folds),

    X_k,f^T u_f   = (X^T u'_f - mu_f (1^T u'_f)) / sigma_f,    u'_f = u_f - Ts_f (Ts_f^T u_f)            [xt_v, F columns]
    X_k,f,b w_fb  = (I - Ts_f Ts_f^T) [ m_f o (X_b (w_fb / sigma_fb) - mu_fb^T (w_fb / sigma_fb)) ]       [x_w, F columns]

(u_f, ts_fl vanish on the held-out rows).  Everything p-sized stays on the device (csrc/cv_batch.hip):
* the folds' statistics: full-data mean / variance from one statistics pass, the held-out rows' centred sums and sums of
  squares from two products with the held-out indicators (``mbpls_hip_xt_v_centred``), ``mbpls_hip_cv_fold_stats``
  (corrected two-pass form around the full-data mean, sklearn's constant-column rule; the few entries that lose their
  variance to cancellation are taken again from the rows themselves);
* the p x F algebra between the two products and its block sums (``mbpls_hip_cv_weights``);
* the finished weight / loading columns of every fold and the end-of-fit algebra: P_f^T W_f per fold
  (``mbpls_hip_cross_product_dev`` on column-strided views), the K x K pseudo-inverses on the host (one stacked call),
  beta_f = W_f M_f V_f^T (``mbpls_hip_right_multiply_dev``), and the predictions of ALL rows for all folds as one more
  product, of which every fold keeps its held-out rows.
Only n x F arrays and K x K matrices cross PCIe; the n x F x K algebra -- with the deflation on the n side it is far
cheaper than through the loadings -- stays on the host like the n x B algebra of a single fit.

A batch of F folds costs 5 K + 3 products instead of F complete fits: leave-one-out on a small data set (n = 150,
p = 3000, K = 5: all launches) and k-fold on a matrix that fills the GPU (every read of X serves all folds) alike.

Used by ``model_selection.cross_val_predict`` for complete data, K <= 32, at least 8 folds' worth of device memory and
(round 4, ``eligible``) one response with EVERY method -- they fit the same model then --, several-response NIPALS
(2-norm stop rule; the folds' inner loops in q-space, ``_qspace_loops_device``) and the p > n branches of several-response
UNIPALS / KERNEL (stacked thin SVDs); anything else takes the fold-by-fold loop.  Same numbers as that loop and as a
loop of fits of the CPU oracle (tests/test_gpu_parity.py) to ~1e-12.
"""
from __future__ import annotations

import warnings

import numpy as np

from . import _capi, solvers
from .engine import HipEngine

class DegenerateFold(Exception):
    """a fold's iteration broke down (zero covariance): model_selection falls back to the fold-by-fold loop"""


_EPS = np.finfo(np.float64).eps
MAX_FOLDS = 256                   # folds per batch (the matrix-core products read X once per 64 right-hand sides)
MIN_FOLDS = 8                     # fewer folds per read of X than this: fold-by-fold fits


def _folds_per_batch(ctx, p, K, n_folds):
    """as many folds per batch as a quarter of the free device memory holds: (2 K + 6) p x F arrays"""
    free = ctx.info()["free_bytes"] + ctx._pool_bytes
    return int(max(0, min(MAX_FOLDS, n_folds, (free // 4) // (8 * p * (2 * K + 6)))))


MAX_RESPONSES = 16


def _agree_min(comm, value):
    """the smallest ``value`` over the ranks of a column-sharded run (every rank must take the same route and the same
    batch size, or their collectives no longer pair up); the value itself on one GPU"""
    if not getattr(comm, "is_distributed", False):
        return value
    world, rank = int(comm.world_size), int(comm.rank)
    mine = np.zeros(world)
    mine[rank] = float(value)
    return type(value)(comm.allreduce_host(mine).min())


def eligible(estimator, q, n, p, n_folds, ctx=None, comm=None):
    """One response: EVERY method -- with a single response NIPALS, UNIPALS, KERNEL and SIMPLS fit the same model
    (ordinary PLS1 on the concatenated blocks: identical ``beta_`` and predictions to 1e-14), so
    their folds take the batched NIPALS passes, without the confirming second iteration.  Several responses: NIPALS
    (the inner loop of every fold advances together, a fold leaves it when its own stop rule is met).
    ``p``: this rank's columns; under column shards (``comm``) the branch conditions use the total width and the
    memory test is agreed between the ranks."""
    p_total = p if not getattr(comm, "is_distributed", False) else int(comm.allreduce_host(np.array([float(p)]))[0])
    K = int(estimator.n_components)
    if estimator.sparse_data is True or n_folds < 2 or K > 32:
        return False
    if q == 1:
        ok = estimator.method in ("NIPALS", "UNIPALS", "KERNEL", "SIMPLS")
        if estimator.method == "NIPALS":
            ok = estimator.nipals_convergence_norm in (2, -2, "fro", "nuc")
        if estimator.method == "SIMPLS" and not estimator.standardize:
            # SIMPLS centres its scores (:1007-1009) whatever the data; on data that is NOT centred it is therefore a
            # different model from the other three (found by the randomised sweep of round 5): fold by fold
            ok = False
    elif estimator.method == "NIPALS":
        ok = q <= MAX_RESPONSES and estimator.nipals_convergence_norm in (2, -2, "fro", "nuc")
    else:
        # UNIPALS / KERNEL with several responses: their p > n branches (mbpls.py:481-572, :694-805) take the top left
        # singular vector of the same n x q matrix G the NIPALS loop iterates on (times a q x q factor of Y^T Y), so a
        # component is the same q sweeps per batch plus F thin SVDs; the n >= p branches are a different model (an
        # eigenvector of a non-symmetric matrix is not its singular vector) and stay with the fold-by-fold loop
        ok = estimator.method in ("UNIPALS", "KERNEL") and q <= MAX_RESPONSES and p_total > n
    if ok and ctx is not None:
        ok = _agree_min(comm, _folds_per_batch(ctx, p, K + (q if q > 1 else 0), n_folds)) >= min(MIN_FOLDS, n_folds)
    return ok


def predict_folds(estimator, Xdev, ctx, comm, y, splits, batch=None):
    """out-of-fold predictions (n x q) for every (train, test) pair of ``splits`` (see ``eligible`` for the models
    covered); Xdev: the raw matrix resident in HBM (blocks side by side); it is only ever read"""
    n, p = Xdev.n, Xdev.p
    K = int(estimator.n_components)
    y = np.asarray(y, dtype=np.float64).reshape(n, -1)
    q = y.shape[1]
    if batch is None:
        batch = max(1, _agree_min(comm, _folds_per_batch(ctx, p, K + (q if q > 1 else 0), len(splits))))
    std = bool(estimator.standardize)
    out = np.full((n, q), np.nan)
    # one response: ONE sweep per component -- a method without iteration, or NIPALS, whose second iteration only confirms
    # the first (the default route of a single fit as well; MBPLS_NIPALS_Q1_FIXED_POINT=0 makes the folds iterate like
    # the reference)
    from .plan import FitOptions
    one_pass = q == 1 and (estimator.method != "NIPALS" or FitOptions.from_env().q1_fixed_point)
    method = estimator.method if q > 1 else "NIPALS"
    eng = HipEngine(ctx=ctx, comm=comm)
    eng.load_device(Xdev, copy=False)
    from . import runtime
    try:
        with runtime.host_threads():                 # BLAS pools sized to the CPU quota, not to the visible CPUs
            # statistics pass: full-data mean / variance in one read, nothing written -- and the same
            # validation as the fold-by-fold loop: a NaN / inf anywhere poisons a column's sum of squares -> ValueError
            eng.prepare_stats(std)
            eng._kscale = False                      # (the products below take the RAW matrix)
            for s0 in range(0, len(splits), batch):
                _one_batch(eng, Xdev, y, splits[s0:s0 + batch], out, K, std, float(estimator.max_tol), one_pass, method)
    finally:
        eng.X = None                                 # the caller owns the matrix
        eng.close()
    return out


def _fix_risky(Xdev, M, n_f, dmu, drs, p, F):
    """the (column, fold) entries whose downdated variance was lost to cancellation: statistics from the rows"""
    rs = drs.get()
    lib, ctx = Xdev.ctx.lib, Xdev.ctx
    for j, f in zip(*np.nonzero(rs < 0.0)):
        col = Xdev.download(int(j), 1)[M[:, f] != 0.0, 0]
        mean = col.mean()
        d = col - mean
        var = (np.sum(d * d) - np.sum(d) ** 2 / n_f[f]) / n_f[f]
        scale = 1.0 if var <= n_f[f] * _EPS * var + (n_f[f] * mean * _EPS) ** 2 else np.sqrt(var)
        fix = np.array([mean, 1.0 / scale])
        off = 8 * (int(f) * p + int(j))
        _capi.check(lib.mbpls_hip_memcpy_h2d(ctx.handle, dmu.ptr + off, fix[0:1].ctypes.data, 8))
        _capi.check(lib.mbpls_hip_memcpy_h2d(ctx.handle, drs.ptr + off, fix[1:2].ctypes.data, 8))
        ctx.sync()


def _qspace_loops_device(ctx, G, Ys, tol, max_iter, k):
    """_qspace_loops with the iteration itself on the device: one wave per fold (csrc/cv_batch.hip k_cv_qloops); the
    host only forms the folds' q x q moments and, afterwards, ts and v (a hundred iterations of a dozen numpy calls
    per component were half of a batch's time)."""
    F, q = G.shape[1], G.shape[2]
    M = np.empty((F, 3, q, q))
    M[:, 0] = np.einsum("nfa,nfb->fab", G, G)
    M[:, 1] = np.einsum("nfa,nfb->fab", Ys, G)
    M[:, 2] = np.einsum("nfa,nfb->fab", Ys, Ys)
    dM = ctx.from_host(M.ravel())
    dout = ctx.empty(F * (2 * q + 3))
    try:
        _capi.check(ctx.lib.mbpls_hip_cv_qloops(ctx.handle, F, q, dM.ptr, float(tol), int(max_iter), dout.ptr))
        o = dout.get().reshape(F, 2 * q + 3)
    finally:
        dM.free(); dout.free()
    ct, c_used, flag, n_it = o[:, :q], o[:, q:2 * q], o[:, 2 * q], o[:, 2 * q + 1].astype(np.int64)
    if (flag == 2.0).any():
        if not np.isfinite(o).all():
            raise DegenerateFold(f"component {k}")
        warnings.warn(f"NIPALS component {k}: no convergence to max_tol={tol} after {max_iter} iterations in "
                      f"{int((flag == 2.0).sum())} fold(s); the reference would loop forever here.")
    ts = np.einsum("nfq,fq->nf", G, ct)
    ts /= np.sqrt((ts * ts).sum(axis=0))
    v = np.einsum("nfq,nf->fq", Ys, ts) / (ts * ts).sum(axis=0)[:, None]
    if not (np.isfinite(ts).all() and np.isfinite(v).all() and np.isfinite(c_used).all()):
        raise DegenerateFold(f"component {k}")
    return n_it, c_used, ts, v


def _qspace_loops(G, Ys, tol, max_iter, k):
    """The NIPALS inner loop (mbpls.py:841-913) of EVERY fold in the q-dimensional space of its iterates -- the batched
    form of solvers._subspace_loop_qspace: with M1 = G^T G, M2 = Y^T G, M3 = Y^T Y per fold, an iteration is O(q^2) per
    fold and all folds advance together (a fold leaves when ||ts_old - ts|| <= tol from the second iteration on).
    G, Ys: n x F x q.  Returns (iterations F, c of the last sweep vector F x q, ts n x F, v F x q)."""
    F, q = G.shape[1], G.shape[2]
    M1 = np.einsum("nfa,nfb->fab", G, G)
    M2 = np.einsum("nfa,nfb->fab", Ys, G)
    M3 = np.einsum("nfa,nfb->fab", Ys, Ys)
    c = np.zeros((F, q)); c[:, 0] = 1.0                                      # the start vector Y[:, 0] (:838)
    active = np.ones(F, dtype=bool)
    ct_old = None
    c_fin, ct_fin, n_it = np.zeros((F, q)), np.zeros((F, q)), np.zeros(F, dtype=np.int64)
    run = 0
    def quad(M, a):                                                          # a_f^T M_f a_f (matmul: the fast path)
        return (a * np.matmul(M, a[:, :, None])[:, :, 0]).sum(axis=1)

    while active.any():
        run += 1
        ct = c / np.sqrt(quad(M1, c))[:, None]                               # ts = G c~, unit length (:882-883)
        if run > 1:
            diff = np.sqrt(np.maximum(quad(M1, ct_old - ct), 0.0))           # (:887)
        else:
            diff = np.full(F, np.inf)
        v = np.matmul(M2, ct[:, :, None])[:, :, 0]                           # (:899)
        vv = (v * v).sum(axis=1)
        un = np.sqrt(quad(M3, v)) / vv                                       # ||Y v / v^T v|| (:911-913)
        done = active & (((run >= 2) & (diff <= tol)) | (run >= max_iter))
        if run >= max_iter and (active & (diff > tol)).any():
            warnings.warn(f"NIPALS component {k}: no convergence to max_tol={tol} after {max_iter} iterations in "
                          f"{int((active & (diff > tol)).sum())} fold(s); the reference would loop forever here.")
        c_fin[done] = c[done]; ct_fin[done] = ct[done]; n_it[done] = run
        ct_old = ct
        nxt = active & ~done
        c[nxt] = (v / vv[:, None] / un[:, None])[nxt]
        active = nxt
        if not np.isfinite(c[active]).all():
            raise DegenerateFold(f"component {k}")
    ts = np.einsum("nfq,fq->nf", G, ct_fin)
    ts /= np.sqrt((ts * ts).sum(axis=0))
    v = np.einsum("nfq,nf->fq", Ys, ts) / (ts * ts).sum(axis=0)[:, None]
    if not (np.isfinite(ts).all() and np.isfinite(v).all()):
        raise DegenerateFold(f"component {k}")
    return n_it, c_fin, ts, v


def _one_batch(eng, Xdev, y, part, out, K, std, tol, one_pass=False, method="NIPALS"):
    n, p = Xdev.n, Xdev.p
    q = y.shape[1]
    offs = list(Xdev.block_offsets)
    F, B = len(part), len(offs) - 1
    ctx, lib = eng.ctx, eng.ctx.lib
    M = np.zeros((n, F))
    for f, (tr, te) in enumerate(part):
        M[np.asarray(tr), f] = 1.0
    n_f = M.sum(axis=0)
    nf1 = n_f[:, None]
    if std:                                          # ColumnScaler.fit on the training responses of every fold
        ymu = np.einsum("nq,nf->fq", y, M) / nf1                             # F x q
        d = (y[:, None, :] - ymu[None]) * M[:, :, None]                      # n x F x q
        var = ((d * d).sum(axis=0) - d.sum(axis=0) ** 2 / nf1) / nf1
        ysig = np.where(var <= nf1 * _EPS * var + (nf1 * ymu * _EPS) ** 2, 1.0, np.sqrt(var))
    else:
        ymu, ysig = np.zeros((F, q)), np.ones((F, q))
    Ys = (y[:, None, :] - ymu[None]) / ysig[None] * M[:, :, None]            # n x F x q, zero on the held-out rows
    pf8 = 8 * p * F
    G, Cs = eng.slot("cvG", F, zero=False), eng.slot("cvC", F, zero=False)
    Wst, Pst = eng.slot("cvW", F * K, zero=False), eng.slot("cvP", F * K, zero=False)
    dg, dmu, drs = ctx.empty((p, F)), ctx.empty((p, F)), ctx.empty((p, F))
    dsu, dal, dsums = ctx.empty(F), ctx.empty(F), ctx.empty(B * F * 2)
    tmp = []
    try:
        # ---- column statistics of every fold's training rows, on the device
        if std:
            dH, dnf, drisky = ctx.from_host(1.0 - M), ctx.from_host(n_f), ctx.empty(1)
            tmp += [dH, dnf, drisky]
            _capi.check(lib.mbpls_hip_xt_v_centred(Xdev.handle, eng.mean.ptr, 0, dH.ptr, n, F, G.ptr, p))    # h1 -> cvG
            _capi.check(lib.mbpls_hip_xt_v_centred(Xdev.handle, eng.mean.ptr, 1, dH.ptr, n, F, dg.ptr, p))   # h2 -> dg
            _capi.check(lib.mbpls_hip_cv_fold_stats(Xdev.handle, F, eng.mean.ptr, eng.var.ptr, G.ptr, dg.ptr, dnf.ptr,
                                                    dmu.ptr, drs.ptr, drisky.ptr))
            if drisky.get().view(np.uint64)[0]:
                _fix_risky(Xdev, M, n_f, dmu, drs, p, F)
        else:
            _capi.check(lib.mbpls_hip_memset(ctx.handle, dmu.ptr, 0, pf8))
            drs.set(np.ones((p, F)))

        def weights(U, alpha):
            """sweep: g = alpha (X^T U - mu su) rs, C = g rs for every fold, block sums in dsums (all on the device)"""
            eng.xt_v(U, 1.0, "cvG", 0)                                       # X^T U (p x F), raw X
            dsu.set(U.sum(axis=0)); dal.set(alpha)
            _capi.check(lib.mbpls_hip_cv_weights(Xdev.handle, F, 0, G.ptr, dmu.ptr, drs.ptr, dsu.ptr, dal.ptr, dg.ptr,
                                                 Cs.ptr, dsums.ptr))

        def keep(stack, k, cols):
            """finished columns of component k (all folds, or the folds in ``cols``) from dg into a stack"""
            if cols is None:
                _capi.check(lib.mbpls_hip_memcpy_d2d(ctx.handle, stack.ptr + k * pf8, dg.ptr, pf8))
            else:
                for f in cols:
                    _capi.check(lib.mbpls_hip_memcpy_d2d(ctx.handle, stack.ptr + k * pf8 + 8 * p * int(f),
                                                         dg.ptr + 8 * p * int(f), 8 * p))

        Ts = np.zeros((F, n, K))                                             # fold-major: the deflation is n x K algebra
        nu_all = np.ones((B, F, K)); V = np.zeros((F, q, K))

        def deflate(A, k):
            """(I - Ts_f Ts_f^T) a_f for every fold (columns of the n x F array A): n x k algebra per fold"""
            if not k:
                return A
            T = Ts[:, :, :k]                                                 # F x n x k
            c = np.matmul(T.transpose(0, 2, 1), A.T[:, :, None])             # F x k x 1
            return A - np.matmul(T, c)[:, :, 0].T

        def sweep(U, alpha, k):
            """block sums of squares of g = alpha X_k^T u (left on the device) and t~_b = X_k,b g_b -- for every fold"""
            weights(deflate(U, k), alpha)
            XC = eng.x_w_slot("cvC", 0, F, per_block=True)                   # per block: X_b C_b (n x F)
            sums = eng.comm.allreduce_host(dsums.get()).reshape(B, F, 2)   # (block sums: over the column shards too)
            t = np.empty((B, n, F))
            for b in range(B):
                t[b] = deflate((XC[b] - sums[b, :, 1]) * M, k)
            return sums[:, :, 0], t

        Sst = eng.slot("cvS", F * q, zero=False) if q > 1 else None          # X_k^T (P y_j) of fold f: column j F + f
        for k in range(K):
            if q > 1:
                # Several responses: every sweep vector of the inner loop is u = Y c and the super score it yields is
                # ts ~ G c with G = P X (X^T (P Y)) (solvers._fit_nipals_subspace), so q sweeps per COMPONENT -- all folds
                # at once -- give every fold its n x q matrix G_f and the whole loop is q x q algebra, advanced for all
                # folds together on the host; weights and block scores of the last sweep vector are combinations of
                # the same products.
                Bk = np.empty((B, n, F, q))
                for j in range(q):
                    _, tb = sweep(Ys[:, :, j], np.ones(F), k)                # g_j = X_k^T (P y_j) in dg, t~_b = P X_b g_j,b
                    _capi.check(lib.mbpls_hip_memcpy_d2d(ctx.handle, Sst.ptr + j * pf8, dg.ptr, pf8))
                    Bk[:, :, :, j] = tb
                Gk = Bk.sum(axis=0)                                          # n x F x q: P X X^T P Y of every fold
                if method == "NIPALS":
                    n_it, c_used, ts_fin, v_fin = _qspace_loops_device(ctx, Gk, Ys, tol, solvers.NIPALS_MAX_ITER, k)
                    u_used = np.einsum("nfq,fq->nf", Ys, c_used)             # sweep vector of the last iteration
                    uu = (u_used * u_used).sum(axis=0)
                    coef = c_used / uu[:, None]                              # w~ = S c / u^T u (:856)
                else:
                    # UNIPALS (:489-504): ts = tlsv(G F(Y^T Y)), w = X^T (Y v);  KERNEL (:712-731): the same with the
                    # deflated responses Y_d = P Y in the factor and w = X^T (Y_d (Y_d^T ts)) -- all folds in stacked
                    # LAPACK calls (F thin SVDs of n x q, F symmetric q x q eigen-decompositions)
                    Yf = Ys if method == "UNIPALS" else np.stack([deflate(Ys[:, :, j], k) for j in range(q)], axis=2)
                    lam, Q = np.linalg.eigh(np.einsum("nfa,nfb->fab", Yf, Yf))
                    Fm = Q * np.sqrt(np.clip(lam, 0.0, None))[:, None, :]    # F(Y^T Y) per fold (solvers._psd_factor)
                    thin = np.matmul(Gk.transpose(1, 0, 2), Fm)              # F x n x q
                    ts_fin = np.linalg.svd(thin, full_matrices=False)[0][:, :, 0].T.copy()   # n x F, unit length
                    v_fin = np.einsum("nfq,nf->fq", Ys, ts_fin) / (ts_fin * ts_fin).sum(axis=0)[:, None]
                    c_used = v_fin if method == "UNIPALS" else np.einsum("nfq,nf->fq", Yf, ts_fin)
                    u_used = np.einsum("nfq,fq->nf", Yf, c_used)
                    uu = np.ones(F)
                    coef = c_used
                    if not (np.isfinite(ts_fin).all() and np.isfinite(coef).all()):
                        raise DegenerateFold(f"component {k}")
                dco = ctx.from_host(np.ascontiguousarray(coef).ravel())
                _capi.check(lib.mbpls_hip_right_multiply_folds(ctx.handle, p, F, Sst.ptr, p * F, p, q, dco.ptr, 1,
                                                               Wst.ptr + k * pf8, p, p))
                dco.free()
                Bc = np.einsum("bnfq,fq->bnf", Bk, c_used)                   # P X_b X_b^T P u
                nu_fin = np.sqrt(np.einsum("nf,bnf->bf", u_used, Bc)) / uu   # ||w~_b|| (:860)
                weights(deflate(ts_fin, k), np.ones(F))                      # loadings p = X_k^T ts (:928)
                keep(Pst, k, None)
                Ts[:, :, k] = ts_fin.T; nu_all[:, :, k] = nu_fin; V[:, :, k] = v_fin
                continue
            u = Ys[:, :, 0].copy()                                           # the first response starts every component (:838)
            active = np.ones(F, dtype=bool)
            ts_old = np.zeros((n, F))
            nu_fin = np.ones((B, F)); ts_fin = np.zeros((n, F)); v_fin = np.zeros((F, q))
            run = 1
            prev_diff, stalls = None, np.zeros(F, dtype=np.int64)
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
                v = np.einsum("nfq,nf->fq", Ys, ts) / (ts * ts).sum(axis=0)[:, None]     # F x q (:899)
                un = np.einsum("nfq,fq->nf", Ys, v) / (v * v).sum(axis=1)    # (:911)
                un /= np.sqrt((un * un).sum(axis=0))
                if not (np.isfinite(un[:, active]).all() and np.isfinite(ts[:, active]).all()):
                    # a fold without any covariance left (v = Y^T ts = 0) or with a degenerate score: the fold-by-fold
                    # loop reports that fold the way a single fit does
                    raise DegenerateFold(f"component {k}")
                if one_pass:
                    diff = np.zeros(F)                                       # no iteration: the first sweep is the model
                # stagnation rule of the fold-by-fold loop (solvers.q1_noise_floor), per fold (one response only)
                floor_hit = np.zeros(F, dtype=bool)
                if run >= 2 and q == 1:
                    floor_hit, stalls = solvers.q1_noise_floor(run, diff, prev_diff, stalls)
                    prev_diff = diff
                floor_hit = floor_hit & active & (diff > tol)
                capped = run >= solvers.NIPALS_MAX_ITER
                if capped and (active & (diff > tol)).any():
                    warnings.warn(f"NIPALS component {k}: no convergence to max_tol={tol} after {run} iterations in "
                                  f"{int((active & (diff > tol)).sum())} fold(s); the reference would loop forever here.")
                done = active & ((diff <= tol) | floor_hit | capped)
                if floor_hit.any():
                    warnings.warn(f"NIPALS component {k}: ||ts_old - ts|| stays above max_tol={tol} in "
                                  f"{int(floor_hit.sum())} fold(s) (rounding noise of an exhausted "
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
            Ts[:, :, k] = ts_fin.T; nu_all[:, :, k] = nu_fin; V[:, :, k] = v_fin

        # ---- every fold: C_f = P_f^T W_f on the device (column k of fold f is column k F + f of the stacks: a p x K
        # matrix with leading dimension p F), the K x K pseudo-inverses stacked on the host, beta_f = W_f M_f V_f^T on the
        # device (:986-989)
        dC = ctx.empty(F * K * K)
        tmp.append(dC)
        Bq = G if q == 1 else eng.slot("cvB", F * q, zero=False)             # beta of response j, fold f: column j F + f
        for f in range(F):
            _capi.check(lib.mbpls_hip_cross_product_dev(ctx.handle, p, Pst.ptr + 8 * p * f, p * F, K, Wst.ptr + 8 * p * f,
                                                        p * F, K, dC.ptr + 8 * K * K * f))
        Cm = eng.comm.allreduce_host(dC.get()).reshape(F, K, K)           # P_f^T W_f: summed over the column shards
        D = 1.0 / np.sqrt((nu_all ** 2).sum(axis=0))                         # F x K: 1 / ||W~[:, k]||
        Mv = np.matmul(D[:, :, None] * np.linalg.pinv(Cm * D[:, None, :]), V.transpose(0, 2, 1))     # F x K x q
        dMv = ctx.from_host(np.ascontiguousarray(Mv).ravel())                # per fold a K x q row-major matrix
        tmp.append(dMv)
        # beta_f = W_f M_f V_f^T for every fold in one launch: beta_f[:, j] -> column j F + f
        _capi.check(lib.mbpls_hip_right_multiply_folds(ctx.handle, p, F, Wst.ptr, p * F, p, K, dMv.ptr, q, Bq.ptr, p * F, p))
        # ---- predictions of all rows for all folds: one more product per response (:1369-1408); a fold keeps its
        # held-out rows
        for j in range(q):
            _capi.check(lib.mbpls_hip_cv_weights(Xdev.handle, F, 1, Bq.ptr + 8 * p * F * j, dmu.ptr, drs.ptr, None, None,
                                                 dg.ptr, Cs.ptr, dsums.ptr))
            XC = eng.x_w_slot("cvC", 0, F, per_block=True)
            sums = eng.comm.allreduce_host(dsums.get()).reshape(B, F, 2)   # (block sums: over the column shards too)
            yhat = sum(XC[b] - sums[b, :, 1] for b in range(B)) * ysig[:, j] + ymu[:, j]          # n x F
            for f, (tr, te) in enumerate(part):
                te = np.asarray(te)
                out[te, j] = yhat[te, f]
    finally:
        for a_ in [dg, dmu, drs, dsu, dal, dsums] + tmp:
            a_.free()
