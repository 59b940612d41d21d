"""Host drivers of the four solver branches of ``MBPLS.fit`` (reference mbpls.py:384-1048).

Control flow and the tiny n x B / n x q / K x K algebra stay on the host in numpy, exactly
where the reference has them; every X-sized operation is a call into the engine (HIP
kernels).  The formulations are the lean restatements of SURVEY.md Appendix A, each verified
against the imported reference to <= 5e-15:

* postponed normalisation  t_b = (X_b w~_b) / ||w~_b||      -> weights and scores in ONE read
* loadings + deflation (+ next component's first sweep)      -> ONE read + ONE write
* analytic explained variance (ts^T ts)(p^T p)               -> no n x p temporaries
* rank-q thin-factor singular vectors                        -> no n x n SVD / ARPACK
"""
from __future__ import annotations

import warnings

import numpy as np

# Guard against the reference's unbounded ``while diff_t > max_tol`` (mbpls.py:841): on data
# without latent structure it never terminates.  Far above any converging case (q=10: ~100).
NIPALS_MAX_ITER = 5000
# One complete response: u = Y v / ||Y v|| = +-y/||y|| whatever ts was (v is 1 x 1), so the NIPALS iteration sits at its
# fixed point after the FIRST pass and every later pass recomputes the same ts; successive ts then differ by rounding
# only.  Where y is (nearly) exhausted -- beyond ~29 components at n = 4096 on the bench's planted data: explained Y
# > 0.9999 -- that rounding is amplified by ||y|| / ||y residual|| and ||ts_old - ts|| <= 1e-14 (mbpls.py:841) is out of
# reach for the reference too: its loop does not end (the CPU oracle, explicit deflation, shows 2 iterations for
# components 0..28 and then 6, cap, cap, 4, cap, ... on the same data -- tools/deep_iterations_cpu.py).  More passes
# only re-roll the rounding, so the one-response loop gives up once the difference has STAGNATED -- not before this many
# iterations, after two consecutive iterations without at least halving it, and in any case after
# NIPALS_Q1_NOISE_FLOOR_MAX -- ALWAYS with a warning, and the component is flagged in ``nipals_converged_``.  (A fixed
# cut-off at four iterations also stopped components the reference finishes in five: a difference that still shrinks is
# convergence, not noise.)  Listed among the deliberate deviations.
NIPALS_Q1_NOISE_FLOOR_ITER = 4
NIPALS_Q1_NOISE_FLOOR_MAX = 12


def q1_noise_floor(run, diff, prev_diff, stalls):
    """One-response NIPALS, after iteration ``run`` (>= 2) with difference ``diff`` (``prev_diff``: the one before, or
    None): returns (give_up, stalls).  Works element-wise on arrays (one entry per cross-validation fold)."""
    if prev_diff is None:
        return (np.zeros(np.shape(diff), dtype=bool) if np.ndim(diff) else False), stalls
    stalled = np.asarray(diff) >= 0.5 * np.asarray(prev_diff)
    stalls = np.where(stalled, np.asarray(stalls) + 1, 0)
    give_up = ((run >= NIPALS_Q1_NOISE_FLOOR_ITER) & (stalls >= 2)) | (run >= NIPALS_Q1_NOISE_FLOOR_MAX)
    if not np.ndim(diff):
        return bool(give_up), int(stalls)
    return give_up, stalls


def _norm(x, order):
    """np.linalg.norm of the n x 1 *matrix* ts_old - ts (mbpls.py:887): matrix-norm rules."""
    return np.linalg.norm(np.asarray(x).reshape(-1, 1), ord=order)


_NORM_ORDERS_OK = set()


def _check_norm_order(order):
    """raises for an invalid ``nipals_convergence_norm`` exactly as the reference does at :887 (numpy's own errors);
    an order that passed once is not tried again (for order 2 the trial is an SVD: 35 us at the head of every fit)"""
    key = (type(order).__name__, order if isinstance(order, (int, float, str)) else repr(order))
    if key not in _NORM_ORDERS_OK:
        _norm(np.zeros((2, 1)), order)
        _NORM_ORDERS_OK.add(key)


def _psd_factor(G):
    lam, Q = np.linalg.eigh((G + G.T) * 0.5)
    return Q * np.sqrt(np.clip(lam, 0.0, None))


def _tlsv(thin, reduce=None):
    """first left singular vector of a thin matrix (host LAPACK), n x 1.  ``reduce`` (column shards: the ROWS of a p x q
    operand are split over the ranks): the sum over ranks of a small host array -- the Gram route below is then the
    only one (its q x q matrix and the norm of the vector are sums over the row shards).
    Tall and thin (m >= 8 q): from the leading
    eigenvector v1 of the q x q Gram matrix, u1 = A v1 / ||A v1|| -- for the LEADING pair the squared spectrum costs at
    most a factor (s1 + s2)/s1 in the perturbation bound, and one q x q eigenproblem + two products replace LAPACK's
    QR-then-SVD of all q vectors (20 000 x 10: 1.3 -> 0.1 ms per component on the GPU box's host, 39 -> 1.8 ms here);
    a nearly double leading singular value (where the vector is ill-determined either way) takes the SVD."""
    m, q = thin.shape
    if reduce is not None:
        if q == 1:
            return thin / np.sqrt(reduce(thin.T.dot(thin)).item())
        lam, vec = np.linalg.eigh(reduce(thin.T.dot(thin)))
        u = thin.dot(vec[:, -1:])
        return u / np.sqrt(reduce(u.T.dot(u)).item())
    if q == 1:
        return thin / np.linalg.norm(thin)
    if m >= 8 * q:
        lam, vec = np.linalg.eigh(thin.T.dot(thin))
        if lam[-1] > 0.0 and lam[-1] - lam[-2] > 1e-4 * lam[-1]:
            u = thin.dot(vec[:, -1:])
            return u / np.linalg.norm(u)
    return np.linalg.svd(thin, full_matrices=False)[0][:, 0:1]


def _bip_corrected(a, sizes):
    """mbpls.py:950-964: a_b (1 - p_b/p) renormalised; [1] for a single block."""
    a = np.asarray(a, dtype=np.float64).ravel()
    if a.size == 1:
        return np.array([1.0])
    sizes = np.asarray(sizes, dtype=np.float64)
    f = a * (1.0 - sizes / sizes.sum())
    return f / f.sum()


def _bip_corrected_all(A, sizes):
    """_bip_corrected for every column of A (B x K) at once"""
    if A.shape[0] == 1:
        return np.ones_like(A)
    sizes = np.asarray(sizes, dtype=np.float64)
    f = A * (1.0 - sizes / sizes.sum())[:, None]
    return f / f.sum(axis=0)


def _split(arr, offsets):
    return [arr[offsets[b]:offsets[b + 1]] for b in range(len(offsets) - 1)]


class _Responses:
    """Y side of the NIPALS iteration (mbpls.py:890-913), also for responses with missing values
    (``sparse_data=True``, :832-836 and :891-906): the Y loadings of a column are regressed over its present rows,
    the Y scores of a row over its present responses.  The reference patches the rows it finds in
    ``sparse_X_info_[block][0]`` -- the sparse rows of the LAST X block, a loop variable left over from the X side
    (:903) -- and fails (LinAlgError) on a missing response anywhere else; here every row with a missing response is
    regressed over what is present, which is the same numbers wherever the reference produces any."""

    def __init__(self, Y):
        self.miss = np.isnan(Y)
        self.sparse = bool(self.miss.any())
        self.Y = np.where(self.miss, 0.0, Y) if self.sparse else Y
        if self.sparse:
            self.present = (~self.miss).astype(np.float64)
            self.col_sparse = self.miss.any(axis=0)
            self.row_sparse = self.miss.any(axis=1)
        self.ssq = float((self.Y ** 2).sum())                    # nansum(Y**2) (:827)

    def start(self):
        """u of a new component: Y[:, 0] (:838); with missing responses the first complete column (:836), a uniform
        random vector when there is none (:833-834)"""
        if not self.sparse:
            return self.Y[:, 0:1]
        dense = np.where(~self.col_sparse)[0]
        if dense.size == 0:
            return np.random.rand(self.Y.shape[0], 1)
        return self.Y[:, dense[0]:dense[0] + 1]

    def loadings(self, ts):
        """v = Y^T ts / ts^T ts (:899 / :891-897)"""
        num = self.Y.T.dot(ts)
        v = num / ts.T.dot(ts)
        if self.sparse:
            den = self.present.T.dot(ts ** 2)
            v = np.where(self.col_sparse[:, None], num / np.where(den == 0.0, 1.0, den), v)
        return v

    def scores(self, v):
        """u = Y v / v^T v, unit length (:911-913 / :900-907)"""
        num = self.Y.dot(v)
        u = num / v.T.dot(v)
        if self.sparse:
            den = self.present.dot(v ** 2)
            u = np.where(self.row_sparse[:, None], num / np.where(den == 0.0, 1.0, den), u)
        return u / np.linalg.norm(u)


def _plan(eng, **kw):
    """record the route the fit takes (MBPLS.fit_plan_)"""
    eng.plan.update(kw)


class FitResult(dict):
    __getattr__ = dict.__getitem__

    def __setattr__(self, k, v):
        self[k] = v


def _host_array(eng, shape):
    """F-ordered host array for a p x K attribute that is filled column by column DURING the fit (the copies' own page
    faults hide behind the passes: no helper-thread mapping for these -- except for the LAST column, which arrives behind
    the last pass)"""
    return eng.host_array(shape, prefault=False, prefault_tail_only=True)


class _ColumnsOut:
    """The columns of a finished SIMPLS component (w, r, p: :1005, :1010, :1011) and beta_ = R V^T (:1044) leave with the
    context's copier thread, behind what is queued on the compute stream at that moment and beside everything that
    follows -- instead of a synchronous copy in the fitting thread next to one pass (with few samples a pass is shorter
    than the copy of its three columns: n = 100, p = 10^7, K = 4 spent 24 of 34 ms there), and instead of a p x K product
    on the host for beta_ (7 ms of those 34)."""

    def __init__(self, eng, K, q):
        self.eng, self.ticket = eng, 0
        self.W_h, self.R_h, self.P_h = (_host_array(eng, (eng.p, K)) for _ in range(3))
        self.beta_h = eng.host_array((eng.p, q))

    def send(self, c):
        self.ticket = self.eng.send_cols([("W", c, self.W_h), ("R", c, self.R_h), ("P", c, self.P_h)])

    def send_beta(self, V):
        """V: q x K (the reference's V_); beta_ is formed in a slot from the R slot"""
        eng = self.eng
        eng.slot("beta", V.shape[0])
        eng.right_mul(("R", 0, V.shape[1]), V.T, ("beta", 0))
        self.ticket = eng.send_slot("beta", self.beta_h)

    def wait(self):
        ticket, self.ticket = self.ticket, 0
        self.eng.wait_sent(ticket)


class _ResponseProducts:
    """S = X^T (P Y) (p x q: columns 0..q-1 of a device slot, column q receives the loadings of the component at hand) and
    the per-block products X_b S_b (n x q each, host), kept up to date WITHOUT passes over X.  P = I - Ts Ts^T carries
    the deflation (:969, :535): P_{k+1} = P_k - ts ts^T / ts^T ts gives
        X^T P_{k+1} Y = S - p (ts^T Y),      X_b S_b <- X_b S_b - (X_b p_b)(ts^T Y),      p = X^T ts / ts^T ts,
    and the ONE pass a component needs is the fused sweep with v = ts: it returns the loadings p AND X_b p_b.  Every
    other X-sized quantity of a component is a combination of the columns of S -- the weights X^T (P Y c) = S c, their
    block products X_b (S c)_b = (X_b S_b) c, the matrix X X^T P Y the scores come from = sum_b X_b S_b.
    Set-up: column 0 together with the statistics pass (fused sweep, v = Y[:, 0]), the other columns by one X^T Y and one
    per-block X W product (or one more sweep when q = 2): 3 + K passes per fit."""

    def __init__(self, eng, Y, standardize, sslot="S"):
        n, B, q = eng.n, eng.nblocks, Y.shape[1]
        self.eng, self.q, self.sslot, self.Y = eng, q, sslot, Y
        self.varxb, tb, _ = eng.prepare(standardize, v=Y[:, 0:1], alpha=1.0, g_slot=(sslot, 0))
        self.XS = [np.empty((n, q)) for _ in range(B)]
        for b in range(B):
            self.XS[b][:, 0] = tb[:, b]
        if q == 2:
            tb, _ = eng.sweep(Y[:, 1:2], 1.0, (sslot, 1))
            for b in range(B):
                self.XS[b][:, 1] = tb[:, b]
        elif q > 2:
            eng.scaled_products()                           # (raw matrix: the products standardise in registers too)
            eng.xt_v(Y[:, 1:], 1.0, sslot, 1)
            for b, blk in enumerate(eng.x_w_slot(sslot, 1, q - 1, per_block=True)):
                self.XS[b][:, 1:] = blk

    def total(self):
        """X X^T P Y (before the projection of its rows)"""
        out = self.XS[0].copy()
        for blk in self.XS[1:]:
            out += blk
        return out

    def combination(self, coef, dst):
        """column ``dst`` = S coef (device);  returns X_b (S coef)_b (n x B) and ||(S coef)_b||^2 (B)"""
        eng, offs = self.eng, self.eng.block_offsets
        coef = np.asarray(coef, dtype=np.float64).reshape(self.q, 1)
        eng.right_mul((self.sslot, 0, self.q), coef, dst)
        ssq = np.array([eng.cross((dst[0], dst[1], 1), (dst[0], dst[1], 1), rows=(offs[b], offs[b + 1]))[0, 0]
                        for b in range(eng.nblocks)])
        return np.column_stack([blk.dot(coef) for blk in self.XS]), ssq

    def component(self, ts, inv_tt, pcol):
        """the pass of a component: p = X^T ts / ts^T ts (stored as column ``pcol``), then S and X_b S_b without ts;
        returns ||p_b||^2 (B)"""
        eng, q = self.eng, self.q
        tb, ssq_p = eng.sweep(ts, inv_tt, (self.sslot, q))
        eng.right_mul((self.sslot, q, 1), np.array([[1.0]]), pcol)
        r = np.asarray(ts, dtype=np.float64).reshape(1, -1).dot(self.Y)          # ts^T Y (1 x q)
        M = np.vstack([np.eye(q), -r])
        eng.right_mul((self.sslot, 0, q + 1), M, (self.sslot, 0))               # (in place: a row is read whole first)
        for b, blk in enumerate(self.XS):
            blk -= tb[:, b:b + 1].dot(r)
        return ssq_p


class _StreamedColumns:
    """The p x K attributes of a host-driven fit leaving the device one column at a time: the columns of component k
    are final once the small kernels behind its pass are queued, and leave with the engine's copier (engine.send_cols)
    beside everything that follows -- the end of the fit then fetches R_ and beta_ only.  W_b[:, k] = w_b / ||w_b||
    (:860, :510-512) is formed by the copier from the raw column as it lands."""

    def __init__(self, eng, K, wslot, pslot="P", keep_raw=False, q=None, concat=False):
        self.eng, self.wslot, self.pslot = eng, wslot, pslot
        if q is not None:       # R_ / beta_ only exist at the very end: their pages are mapped while the GPU works
            eng.host_array((eng.p, K), reserve="R")
            eng.host_array((eng.p, q), reserve="beta")
        self.W = _host_array(eng, (eng.p, K))
        self.P = _host_array(eng, (eng.p, K))
        self.W_raw = _host_array(eng, (eng.p, K)) if keep_raw else None
        # KERNEL: W_concat_ = raw column / its norm over ALL blocks (mbpls.py:733) beside the block weights
        self.Wc = _host_array(eng, (eng.p, K)) if concat else None
        if concat:
            eng.slot("Wcd", K)

    def component_done(self, k, blk_norm):
        """the columns of component k are final and queued: they leave with the engine's copier behind what is queued
        now, beside everything that follows, and the block weights are formed from the raw column as it lands.  (Until
        round 6 they were copied synchronously beside ONE pass; with few samples a pass is shorter than these copies and
        the fitting thread waited: n = 100, p = 10^7, three responses 56 -> 33 ms, profiles/r06_wide_few_samples.txt)"""
        eng = self.eng
        nu = np.array(blk_norm, dtype=np.float64).ravel()
        raw = self.W_raw if self.W_raw is not None else self.W
        cols = [(self.wslot, k, raw), (self.pslot, k, self.P)]
        if self.Wc is not None:
            # KERNEL: the same raw column over its norm across ALL blocks (:733) -- a second column on the device (a
            # copier job carries one normalisation): 8p more bytes over PCIe for a p-length division on the host
            eng.right_mul((self.wslot, k, 1), np.array([[1.0 / np.sqrt(float(np.sum(nu * nu)))]]), ("Wcd", k))
            cols.insert(1, ("Wcd", k, self.Wc))
        self.ticket = eng.send_cols(cols, unit=(raw, self.W, k, nu))

    def finish(self):
        ticket, self.ticket = getattr(self, "ticket", 0), 0
        if ticket:
            self.eng.wait_sent(ticket)


def _global_sizes(eng):
    """Block sizes of the WHOLE problem (summed over shards) for the BIP correction."""
    # (once per fit: under column shards this is a collective -- every rank asks the same number of times, in the same
    # places, so remembering the answer on the engine of THIS fit keeps the ranks in step)
    cached = getattr(eng, "_global_sizes_of_fit", None)
    if cached is not None and cached[0] == tuple(eng.block_sizes):
        return cached[1].copy()
    s = eng.comm.allreduce_host(np.asarray(eng.block_sizes, dtype=np.float64))
    eng._global_sizes_of_fit = (tuple(eng.block_sizes), np.array(s, dtype=np.float64))
    return s


def _finalise_device(eng, res, wslot, K, col_norm, blk_norm, V, pslot="P", W_host=None, P_host=None, fetch=True, C=None):
    """End-of-fit algebra on the device (mbpls.py:986-991, :569-572, :733-738).

    wslot holds raw weight columns; the unit-length weights are W = wslot diag(1/col_norm) and the
    block weights W_b = wslot_b diag(1/blk_norm[b]).  With C = P^T wslot (K x K, device reduction,
    all-reduced over shards) and D = diag(1/col_norm):
        R_ = W (P^T W)^+ = wslot [D (C D)^+],   beta_ = R_ V^T = wslot [D (C D)^+ V^T]
    so only K x K matrices touch the host; p x K arrays are fetched once each, as attributes."""
    D = 1.0 / np.asarray(col_norm, dtype=np.float64)
    if C is None:                       # (a device-resident fit brings it along with its other results)
        C = eng.cross((pslot, 0, K), (wslot, 0, K))
    M = D[:, None] * np.linalg.pinv(C * D[None, :])
    q = V.shape[0]
    eng.slot("R", K)
    eng.right_mul((wslot, 0, K), M, ("R", 0))
    eng.slot("beta", q)
    eng.right_mul((wslot, 0, K), M.dot(V.T), ("beta", 0))
    offs = eng.block_offsets
    if W_host is None:      # block weights not streamed out during the fit: scale on the device
        eng.slot("Wblk", K)
        for b in range(eng.nblocks):
            eng.right_mul((wslot, 0, K), np.diag(1.0 / blk_norm[b]), ("Wblk", 0), rows=(offs[b], offs[b + 1]))
    def fetch_all(**kw):
        """D2H of the p x K attributes; kw = dict(overlapped=True) copies on the copy stream (the slots are
        final) so it can run beside further passes over X."""
        res.R_ = eng.fetch("R", **kw)
        res.beta_ = eng.fetch("beta", **kw)
        res.W_ = _split(W_host if W_host is not None else eng.fetch("Wblk", **kw), offs)
        res.P_ = _split(P_host if P_host is not None else eng.fetch(pslot, **kw), offs)

    if fetch == "async" and W_host is not None and P_host is not None:
        # R_ and beta_ leave through the helper thread while the caller does its host-side bookkeeping
        (_, wait_r), (_, wait_b) = eng.fetch_async("R"), eng.fetch_async("beta")

        def finish():
            res.R_, res.beta_ = wait_r(), wait_b()
            res.W_, res.P_ = _split(W_host, offs), _split(P_host, offs)

        return finish
    if not fetch:
        return fetch_all
    fetch_all()
    if fetch == "async":
        return lambda: None


def _finish_resident(eng, res, out, W_host, P_host):
    """R_ / beta_ of a device-resident fit were computed and sent on their way by the queue itself (engine
    _ResidentQueue.__exit__); returns the callable that waits for them and fills in the p-length attributes"""
    wait_r, wait_b = out["finalised"]
    offs = eng.block_offsets

    def finish():
        res.R_, res.beta_ = wait_r(), wait_b()
        res.W_, res.P_ = _split(W_host, offs), _split(P_host, offs)

    return finish


# ----------------------------------------------------------------------------------- NIPALS
def _prepare_products(eng, standardize):
    """a1 for the solvers whose X-sized work is products on a matrix that is never deflated (several-response NIPALS /
    UNIPALS, SIMPLS): a standardised working copy when it fits -- else (or when forced, MBPLS_KERNEL_IMPLICIT)
    statistics only: the products standardise their operand in registers and the caller's matrix is never written.
    Returns the per-block sums of squares of the standardised matrix."""
    if eng.stats_only_ok():
        return eng.prepare_stats(standardize)
    return eng.prepare(standardize)[0]


def _nipals_on_subspace(eng, q):
    """Several responses make NIPALS iterate 50-500 times per component.  Every sweep vector of the loop is u = Y c and
    the super score it yields is ts ~ P (sum_b X_b X_b^T) P u (P = I - Ts Ts^T carries the deflation), so with
    G = P X (X^T (P Y)) -- two many-right-hand-side products over X per COMPONENT -- the whole loop is n x q algebra:
    ts = G c / ||G c||.  X is standardised once and never deflated; a component costs three passes over X whatever its
    iteration count."""
    return q > 1 and q <= 16 and not eng.sparse and eng.opts.nipals_subspace


def _subspace_loop_host(G, Y, c0, norm, tol, max_iter):
    """the loop of csrc/nipals_step.hip k_nipals_subspace_loop in numpy (engines without the kernel, n > 8192)"""
    c = np.asarray(c0, dtype=np.float64).reshape(-1, 1)
    ts_old, run, diff = None, 0, 1.0
    while True:
        run += 1
        c_used = c
        a = G.dot(c)
        ts = a / np.linalg.norm(a)                          # (:882-883)
        if run > 1:
            diff = _norm(ts_old - ts, norm)                 # (:887)
        ts_old = ts
        v = Y.T.dot(ts) / ts.T.dot(ts)                      # (:899)
        u = Y.dot(v) / v.T.dot(v)                           # (:911)
        un = np.linalg.norm(u)
        u = u / un
        c = v / v.T.dot(v).item() / un
        if run >= 2 and diff <= tol:
            return run, False, ts, u, v, c_used
        if run >= max_iter:
            return run, True, ts, u, v, c_used


def _subspace_loop_qspace(G, Y, c0, tol, max_iter):
    """The same loop for LONG score vectors (n > 8192, where the kernel's LDS-resident form does not apply) and the
    2-norm stop rule, entirely in q x q algebra: with M1 = G^T G, M2 = Y^T G, M3 = Y^T Y every quantity of an iteration
    is a function of c -- ||G c||^2 = c^T M1 c, v = M2 c~ (c~ = c/||G c||, ts^T ts = 1), ||Y v||^2 = v^T M3 v -- and
    the stop rule ||ts_old - ts|| = ||G (c~_old - c~)|| = sqrt(d^T M1 d) is taken on the DIFFERENCE of the
    coefficient vectors (formed in q dimensions, so it resolves 1e-14 where 2 - 2 cos would stop at 1e-8).
    An iteration costs O(q^2) instead of five passes over n x q arrays (n = 10^6: 10 ms each, hundreds per component)."""
    M1, M2, M3 = G.T.dot(G), Y.T.dot(G), Y.T.dot(Y)
    c = np.asarray(c0, dtype=np.float64).reshape(-1, 1)
    ct_old, run, diff = None, 0, 1.0
    while True:
        run += 1
        c_used = c
        ct = c / np.sqrt(c.T.dot(M1).dot(c).item())         # ts = G c~, unit length (:882-883)
        if run > 1:
            d = ct_old - ct
            diff = np.sqrt(max(d.T.dot(M1).dot(d).item(), 0.0))                 # (:887)
        ct_old = ct
        v = M2.dot(ct)                                      # (:899)
        vv = v.T.dot(v).item()
        un = np.sqrt(v.T.dot(M3).dot(v).item()) / vv        # ||Y v / v^T v|| (:911-913)
        c = v / vv / un
        if (run >= 2 and diff <= tol) or run >= max_iter:
            ts = G.dot(ct)
            ts = ts / np.linalg.norm(ts)
            u = Y.dot(v) / vv
            u = u / np.linalg.norm(u)
            return run, not (run >= 2 and diff <= tol), ts, u, Y.T.dot(ts) / ts.T.dot(ts), c_used


def _fit_nipals_subspace(eng, Y, K, standardize, max_tol, norm, calc_all, max_iter):
    """mbpls.py:809-993 with the inner loop (:841-913) in the q-dimensional space of its iterates.  Per component:
    G = P X (X^T (P Y)), the loop (one kernel launch), the weights of the last sweep vector w~ = X^T (P u)/(u^T u) (:856)
    with u = Y c, their block products X_b w~_b -- t_b = P X_b w~_b / ||w~_b|| (:875-877), sw_b ~ ||w~_b|| (:879) -- all
    from the maintained products of _ResponseProducts (no pass), then ONE pass: the loadings p = X^T ts (:928).
    X is never written (implicit standardisation, nothing deflated).  3 + K passes per fit."""
    n, B = eng.n, eng.nblocks
    q = Y.shape[1]
    sizes = _global_sizes(eng)
    eng.slot("Wn", K); eng.slot("P", K); eng.slot("S", q + 1)      # S = [X^T (P Y) | the loadings of the component]
    eng.use_lazy_deflation(K, "P", prefer="lazy")           # (never deflates: the never-writes-X passes when they apply)
    cols = _StreamedColumns(eng, K, "Wn", keep_raw=True, q=q)
    prod = _ResponseProducts(eng, Y, standardize)
    varxb = prod.varxb
    varx = varxb.sum()
    vary = (Y ** 2).sum()
    Ts = np.empty((n, K), order="F")        # (its leading columns are contiguous operands of the projections)
    U = np.empty((n, K)); V = np.empty((q, K)); A = np.empty((B, K))
    T = [np.empty((n, K)) for _ in range(B)]
    nu_all = np.empty((B, K))
    evx, evy, evxb, Acorr, iters = [], [], np.empty((B, K)), np.empty((B, K)), []
    converged = []
    _check_norm_order(norm)                           # raises for invalid orders exactly as the reference (:887)
    kind = {2: 0, -2: 0, "fro": 0, "nuc": 0, 1: 1, -1: 1, np.inf: 2, -np.inf: 3}.get(norm)
    c0 = np.zeros((q, 1)); c0[0, 0] = 1.0                   # the start vector Y[:, 0] (:838)
    for k in range(K):
        Tp = Ts[:, :k]
        G0 = prod.total()                                   # X X^T P Y
        G = G0 - Tp.dot(Tp.T.dot(G0))
        out = eng.nipals_subspace_loop(G, Y, c0, kind, max_tol, max_iter)
        if out is None and kind == 0 and n > 8192:
            out = _subspace_loop_qspace(G, Y, c0, max_tol, max_iter)
        if out is None:
            out = _subspace_loop_host(G, Y, c0, norm, max_tol, max_iter)
        n_it, capped, ts, u, v, c_used = out
        if capped:
            warnings.warn(f"NIPALS component {k}: no convergence to max_tol={max_tol} after {max_iter} "
                          f"iterations; the reference would loop forever here.")
        iters.append(n_it)
        converged.append(not capped)
        u_used = Y.dot(c_used)                              # sweep vector of the last iteration
        uu = u_used.T.dot(u_used).item()
        # w~ = X^T (P u) / uu = S c / uu, X_b w~_b, ||w~_b||^2: combinations of maintained products
        t_raw, ssq = prod.combination(c_used / uu, ("Wn", k))
        # the pass of the component: p = X^T ts (ts is orthogonal to the earlier scores), ||p_b||^2; the previous
        # component's attribute columns leave meanwhile
        ssq_p = prod.component(ts, 1.0, ("P", k))
        nu = np.sqrt(ssq)
        Tm = (t_raw - Tp.dot(Tp.T.dot(t_raw))) / nu         # t_b = P X_b w_b (:875-877)
        sw = nu.reshape(-1, 1) / np.linalg.norm(nu)         # sw_b ~ t_b^T u = ||w~_b|| uu (:879-880)
        Ts[:, k:k + 1] = ts; U[:, k:k + 1] = u; V[:, k:k + 1] = v; A[:, k:k + 1] = sw ** 2
        nu_all[:, k] = nu
        cols.component_done(k, nu)
        for b in range(B):
            T[b][:, k] = Tm[:, b]
        if calc_all:                                        # (:932-964)
            tt = ts.T.dot(ts).item()
            evx.append(tt * ssq_p.sum() / varx)
            evy.append(tt * v.T.dot(v).item() / vary)
            evxb[:, k] = tt * ssq_p / varxb
            Acorr[:, k] = _bip_corrected(sw ** 2, sizes)
    res = FitResult(Ts_=Ts, T_=T, U_=U, V_=V, A_=A, nipals_iterations_=iters, nipals_converged_=converged)
    if calc_all:
        res.update(explained_var_x_=evx, explained_var_y_=evy, explained_var_xblocks_=evxb, A_corrected_=Acorr)
    else:
        res.update(explained_var_x_=[], explained_var_y_=[], explained_var_xblocks_=np.empty((B, 0)),
                   A_corrected_=np.empty((B, 0)))
    cols.finish()
    _finalise_device(eng, res, "Wn", K, np.sqrt((nu_all ** 2).sum(axis=0)), nu_all, V, W_host=cols.W, P_host=cols.P)
    res.W_non_normal_ = _split(cols.W_raw, eng.block_offsets)
    return res


def _nipals_on_grams(eng, q):
    """Several responses make NIPALS iterate 50-500 times per component, and every iteration is a pass over X.  With
    p > n the same iterates can be had from the per-block association matrices A_b = X_b X_b^T (n x n each):
    t_b = X_b w_b = A_b u / sqrt(u^T A_b u), so an iteration touches B n^2 doubles instead of n p.  Worth it once the
    fit needs more than about n/75 passes (the Gram costs n(n+1)p flop) -- always the case for q > 1 in practice."""
    if q == 1 or not eng.has_association_matrices or eng.sparse or not eng.opts.nipals_gram:
        return False
    p_total = int(eng.comm.allreduce_host(np.array([float(eng.p)]))[0])
    return p_total > eng.n and eng.n <= 16384 and 8.0 * eng.nblocks * eng.n ** 2 <= 16 * 2 ** 30


def _fit_nipals_grams(eng, Y, K, standardize, max_tol, norm, calc_all, max_iter):
    """mbpls.py:809-993 with the inner loop (:841-913) on the association matrices.  X is standardised once and never
    deflated: the deflation X_b <- (I - ts ts^T) X_b (:969) becomes A_b <- P A_b P, P = I - Ts Ts^T, applied to the
    vectors around each product.  Per component the loop costs B n^2 doubles per iteration; the weights and loadings
    of the converged component are one pass over X:  w~ = X^T (P u)/(u^T u) (:856),  p = X^T ts (:928)."""
    n, B = eng.n, eng.nblocks
    q = Y.shape[1]
    sizes = _global_sizes(eng)
    varxb, _, _ = eng.prepare(standardize)
    eng.gram_blocks()
    varx = varxb.sum()
    vary = (Y ** 2).sum()
    eng.slot("Wn", K); eng.slot("P", K); eng.slot("WP", 2)
    Ts = np.empty((n, K), order="F")        # (its leading columns are contiguous operands of the projections)
    U = np.empty((n, K)); V = np.empty((q, K)); A = np.empty((B, K))
    T = [np.empty((n, K)) for _ in range(B)]
    nu_all = np.empty((B, K))
    evx, evy, evxb, Acorr, iters = [], [], np.empty((B, K)), np.empty((B, K)), []
    converged = []
    # norm of the n x 1 difference as numpy's MATRIX norm (:887): 2, 'fro', 'nuc', -2 -> ||d||_2; 1, -1 -> sum |d|;
    # inf -> max |d|; -inf -> min |d|; anything else raises exactly as in the reference
    _check_norm_order(norm)
    kind = {2: 0, -2: 0, "fro": 0, "nuc": 0, 1: 1, -1: 1, np.inf: 2, -np.inf: 3}.get(norm)
    on_device = kind is not None and K <= 16 and q <= 16
    Yc = np.asfortranarray(Y)
    for k in range(K):
        Tp = Ts[:, :k]
        u = Y[:, 0:1]
        diff, run, ts_old = 1.0, 1, None
        if on_device:
            # ts ~ T sw = P (sum_b A_b) P u: the loop needs one product with the summed Grams per iteration and runs
            # on the device (nipals_step.hip); the per-block quantities of the LAST iteration are taken below
            n_it, capped, ts, u, v, z = eng.gram_power_loop(Yc, Tp, u, kind, max_tol, max_iter)
            if capped:
                warnings.warn(f"NIPALS component {k}: no convergence to max_tol={max_tol} after {max_iter} "
                              f"iterations; the reference would loop forever here.")
            uu = Y[:, 0:1].T.dot(Y[:, 0:1]).item() if n_it == 1 else 1.0     # u^T u of the sweep vector used last
            Az0 = eng.gram_apply_blocks(z)
            quad = z.T.dot(Az0).ravel()
            Tm = (Az0 - Tp.dot(Tp.T.dot(Az0))) / np.sqrt(quad)               # t_b (:875-877)
            sw = np.sqrt(quad).reshape(-1, 1)                                # sw_b ~ t_b^T u = sqrt(u^T A_b u) (:879)
            sw = sw / np.linalg.norm(sw)
            z_used, uu_used = z, uu
            run = n_it + 1
            diff = 0.0
        while diff > max_tol:
            uu = u.T.dot(u).item()
            z = u - Tp.dot(Tp.T.dot(u))                    # P u
            Az0 = eng.gram_apply_blocks(z)                  # A_b,0 (P u)
            quad = z.T.dot(Az0).ravel()                     # u^T A_b,k u = ||X_b^T u||^2          (:860)
            Az = Az0 - Tp.dot(Tp.T.dot(Az0))                # P A_b,0 P u = A_b,k u
            Tm = Az / np.sqrt(quad)                         # t_b = X_b w_b                          (:875-877)
            sw = Tm.T.dot(u) / uu                           # (:879)
            sw = sw / np.linalg.norm(sw)
            ts = Tm.dot(sw)                                 # (:882)
            ts = ts / np.linalg.norm(ts)
            if run > 1:
                diff = _norm(ts_old - ts, norm)             # (:887)
            ts_old = ts.copy()
            z_used, uu_used = z, uu
            v = Y.T.dot(ts) / ts.T.dot(ts)                  # (:899)
            u = Y.dot(v) / v.T.dot(v)                       # (:911)
            u = u / np.linalg.norm(u)
            run += 1
            if run - 1 >= max_iter and diff > max_tol:
                warnings.warn(f"NIPALS component {k}: no convergence to max_tol={max_tol} after {max_iter} "
                              f"iterations (last diff {diff:.3e}); the reference would loop forever here.")
                break
        iters.append(run - 1)
        converged.append(bool(diff <= max_tol))
        # the converged component's p-length vectors: one read of X for both
        ssq = eng.xt_v(np.hstack((z_used / uu_used, ts - Tp.dot(Tp.T.dot(ts)))), 1.0, "WP", 0, want_ssq=True)
        eng.right_mul(("WP", 0, 1), np.array([[1.0]]), ("Wn", k))
        eng.right_mul(("WP", 1, 1), np.array([[1.0]]), ("P", k))
        nu = np.sqrt(ssq[:, 0])                             # ||w~_b||
        ssq_p = ssq[:, 1]
        Ts[:, k:k + 1] = ts; U[:, k:k + 1] = u; V[:, k:k + 1] = v; A[:, k:k + 1] = sw ** 2
        nu_all[:, k] = nu
        for b in range(B):
            T[b][:, k] = Tm[:, b]
        if calc_all:                                        # (:932-964)
            tt = ts.T.dot(ts).item()
            evx.append(tt * ssq_p.sum() / varx)
            evy.append(tt * v.T.dot(v).item() / vary)
            evxb[:, k] = tt * ssq_p / varxb
            Acorr[:, k] = _bip_corrected(sw ** 2, sizes)
    res = FitResult(Ts_=Ts, T_=T, U_=U, V_=V, A_=A, nipals_iterations_=iters, nipals_converged_=converged)
    if calc_all:
        res.update(explained_var_x_=evx, explained_var_y_=evy, explained_var_xblocks_=evxb, A_corrected_=Acorr)
    else:
        res.update(explained_var_x_=[], explained_var_y_=[], explained_var_xblocks_=np.empty((B, 0)),
                   A_corrected_=np.empty((B, 0)))
    _finalise_device(eng, res, "Wn", K, np.sqrt((nu_all ** 2).sum(axis=0)), nu_all, V)
    res.W_non_normal_ = _split(eng.fetch("Wn"), eng.block_offsets)
    return res


def _fit_nipals_q1_fixed_point(eng, Y, K, standardize, calc_all):
    """The default route of one-response NIPALS (MBPLS_NIPALS_Q1_FIXED_POINT=0 selects the reference's two passes per
    component instead: same attributes, 2K + 1 reads of X instead of K + 1).  One response: u after the first pass of a component is
    sign(v) y/||y|| whatever ts was (v is 1 x 1, and v = y^T ts > 0: a sum of squares), so the second pass of the
    reference's loop (:841) computes w~_2 = ||y|| w~_1 and reproduces ts exactly -- the iteration is at its fixed point
    after ONE pass and the second only re-derives it.  Without it a component is: weights X_k^T u with u = y/||y||, block
    scores, ts, loadings -- which is UNIPALS with one response (:481-572: sw_b = t_b^T u / u^T u is proportional to
    ||w~_b||, so ts = T sw is X w / ||X w||): the same model, fitted by the one-pass-per-component queue of fit_unipals.
    NIPALS' own attributes on top: W_non_normal_ (the raw weights of the last sweep vector, :975) and the iteration count,
    reported as 1 per component."""
    res = fit_unipals(eng, Y, K, standardize, keep_raw_weights=True)
    _plan(eng, solver="nipals, one response, fixed point after one pass: " + str(eng.plan.get("solver", "")).replace("unipals: ", ""))
    raw = res.pop("_W_raw_host", None)                     # (FitResult is a dict)
    # the raw weight columns came with the streamed copies of the queue; a late fetch of the 8 p K-byte slot into
    # unmapped pages cost 15-20 ms at p = 10^6
    res.W_non_normal_ = _split(raw if raw is not None else eng.fetch("Wc"), eng.block_offsets)
    res.nipals_iterations_ = [1] * K
    res.nipals_converged_ = [True] * K
    if not calc_all:                                        # (:932-964 are skipped by the reference then)
        B = eng.nblocks
        res.update(explained_var_x_=[], explained_var_y_=[], explained_var_xblocks_=np.empty((B, 0)),
                   A_corrected_=np.empty((B, 0)))
    return res


def fit_nipals(eng, Y, K, standardize, max_tol=1e-14, norm=2, calc_all=True, max_iter=None):
    """mbpls.py:809-993.  Per component: iterate {sweep(u) -> T -> sw, ts -> v -> u} until
    ||ts_old - ts|| <= max_tol (never on the first pass, :884), then p = X^T ts and
    X <- X - ts p^T fused with the next component's first sweep (u = Y[:, 0], :838)."""
    n, B = eng.n, eng.nblocks
    q = Y.shape[1]
    max_iter = NIPALS_MAX_ITER if max_iter is None else max_iter
    if _nipals_on_subspace(eng, q):
        _plan(eng, solver="nipals: inner loop in the q-space of its iterates, one pass per component (maintained products)")
        return _fit_nipals_subspace(eng, Y, K, standardize, max_tol, norm, calc_all, max_iter)
    if _nipals_on_grams(eng, q):
        _plan(eng, solver="nipals: inner loop on per-block association matrices")
        return _fit_nipals_grams(eng, Y, K, standardize, max_tol, norm, calc_all, max_iter)
    resp = _Responses(Y)
    Y = resp.Y                                              # missing responses (sparse_data=True) as zeros + masks
    if q == 1 and eng.opts.q1_fixed_point and not eng.sparse and not resp.sparse and max_iter >= 1:
        _check_norm_order(norm)                       # raises for invalid orders exactly as the reference (:887)
        return _fit_nipals_q1_fixed_point(eng, Y, K, standardize, calc_all)
    sizes = _global_sizes(eng)
    u0 = resp.start()
    alpha0 = 1.0 / u0.T.dot(u0).item()
    eng.slot("Wn", K)
    # One response converges in two passes per component: lazy deflation (no rewrite of X) wins.  Several
    # responses iterate 50-500 times per component (paper protocol, q = 10): every sweep then pays the lazy
    # correction (~6 %), far more than one rewrite of X per component (measured 7.05 s -> 6.72 s).
    eng.use_lazy_deflation(K, "P", prefer="lazy" if q == 1 else "explicit")
    # norm of the n x 1 difference as numpy's MATRIX norm (:887), see _fit_nipals_grams
    _check_norm_order(norm)
    kind = {2: 0, -2: 0, "fro": 0, "nuc": 0, 1: 1, -1: 1, np.inf: 2, -np.inf: 3}.get(norm)
    # One response: the whole fit is queued on the device.  Its first pass (statistics + first sweep) needs nothing but
    # the matrix, the start vector and the W~ slot: launched NOW, so the GPU is at work while the host allocates the
    # other slots and the host copies of the attributes (0.2-0.3 ms of every fit; ~10 ms of allocations of the first)
    go_resident = not resp.sparse and max_iter >= 2 and eng.resident_supported(q, K, kind)
    first = None
    if go_resident:
        first = eng.deferred_first_pass(standardize, u0, alpha0, ("Wn", 0))
        first()
    eng.slot("P", K)
    Ts = np.empty((n, K)); U = np.empty((n, K)); V = np.empty((q, K)); A = np.empty((B, K))
    T = [np.empty((n, K)) for _ in range(B)]
    nu_all = np.empty((B, K))
    evx, evy, evxb, Acorr, iters = [], [], np.empty((B, K)), np.empty((B, K)), []
    converged = []          # per component: the stop rule of mbpls.py:841 was met (not a cap / noise-floor stop)
    # host buffers of the attributes that only exist at the very end of the fit (nothing hides THEIR page faults):
    # allocated now, their pages mapped by the helper thread while the GPU has 2K + 1 passes to do
    eng.host_array((eng.p, K), reserve="R")
    eng.host_array((eng.p, q), reserve="beta")
    # host copies of the p x K attributes, filled column by column while later passes run on the GPU
    Wn_h, W_h, P_h = (_host_array(eng, (eng.p, K)) for _ in range(3))
    vary = resp.ssq
    k_start = 0
    if go_resident:
        # (two iterations per component -- what one response takes; the host reads the results once.  Components the
        # speculation did not hold for are resumed below.)
        out = eng.nipals_resident(Y, K, kind, max_tol, "Wn", Wn_h, W_h, P_h, None)      # first pass already queued
        if out.get("reason") == "offset":
            # badly offset columns: the same queue again with the passes that centre in registers (the first pass is
            # repeated because a sharded run has already summed its exchange buffer once more)
            out = eng.nipals_resident(Y, K, kind, max_tol, "Wn", Wn_h, W_h, P_h, first)
        resident = out.get("reason") != "barrier"
    else:
        resident = False
    _plan(eng, solver="nipals: device-resident queue" if resident else "nipals: host-driven loop")
    if resident:
        _plan(eng, resumed_on_host_at=None if out["k_done"] == K else int(out["k_done"]))
        _plan(eng, r_and_beta=out.get("r_route") if out["k_done"] == K else "one product behind the last pass")
        varxb = out["ssq_x"]                                # summed over the column shards on the device already
        if not np.all(np.isfinite(varxb)):
            raise ValueError("Input X contains NaN, infinity or a value too large for dtype('float64').")
        varx = varxb.sum()
        k_start = out["k_done"]
        whole = k_start == K
        finish = None
        res = FitResult()
        if whole:
            # every component came from the device: the downloaded arrays ARE the attributes (no n x K host copies --
            # at n = 10^6 those cost more than a pass over X)
            Ts, U = out["Ts"], out["U"]
            T = [out["Tb"][:, b, :].T for b in range(B)]
            # end-of-fit algebra first (K x K pinv here, R_ / beta_ on the device, their copies in the helper thread);
            # the per-component bookkeeping below runs while they travel
            nu_all[:] = np.sqrt(out["scal"][:, 0, :].T)
            V[:] = out["V"]
            if out.get("finalised"):
                finish = _finish_resident(eng, res, out, W_h, P_h)      # (already launched by the queue)
            else:
                finish = _finalise_device(eng, res, "Wn", K, np.sqrt((nu_all ** 2).sum(axis=0)), nu_all, V, W_host=W_h,
                                          P_host=P_h, fetch="async", C=out.get("PtW"))
        if not whole:
            out["wait_copies"]()                            # the host loop below writes the same host columns
        else:
            # every component came from the device: the recorded scalars are the attributes, for all components at once
            A[:] = out["scal"][:, 1, :].T
            iters = [int(i) for i in out["iters"]]
            converged = [True] * K                          # (the device step records a component only when it converged)
            if calc_all:                                    # (:932-964), analytic (ts^T ts)(p^T p)
                out["wait_scores"]()                        # (the n-vectors arrive through the copier thread)
                ssq_p = out["scal"][:, 2, :]                # K x B
                tt = np.einsum("ik,ik->k", Ts, Ts)
                evx = list(tt * ssq_p.sum(axis=1) / varx)
                evy = list(tt * np.einsum("jk,jk->k", V, V) / vary)
                evxb[:] = (tt[:, None] * ssq_p / varxb).T
                Acorr[:] = _bip_corrected_all(A, sizes)
        for k in range(k_start if not whole else 0):
            ts, v = out["Ts"][:, k:k + 1], out["V"][:, k:k + 1]
            V[:, k:k + 1] = v
            if not whole:
                Ts[:, k:k + 1] = ts; U[:, k] = out["U"][:, k]
                for b in range(B):
                    T[b][:, k] = out["Tb"][k, b]
            A[:, k] = out["scal"][k, 1]
            nu_all[:, k] = np.sqrt(out["scal"][k, 0])
            iters.append(int(out["iters"][k]))
            converged.append(True)                          # (the device step records a component only when it converged)
            if calc_all:                                    # (:932-964), analytic (ts^T ts)(p^T p)
                ssq_p = out["scal"][k, 2]
                tt = ts.T.dot(ts).item()
                evx.append(tt * ssq_p.sum() / varx)
                evy.append(tt * v.T.dot(v).item() / vary)
                evxb[:, k] = tt * ssq_p / varxb
                Acorr[:, k] = _bip_corrected(A[:, k], sizes)
        if k_start < K:
            t_raw, ssq = eng.sweep(u0, alpha0, ("Wn", k_start))     # first sweep of the component to resume with
    else:
        varxb, t_raw, ssq = eng.prepare(standardize, v=u0, alpha=alpha0, g_slot=("Wn", 0))
        varx = varxb.sum()
    # Missing values in X, one complete response: the second pass of a component re-derives the first here as well -- the
    # masked weights w~_j = sum_present x_ij u_i / sum_present u_i^2 (:848-852) are homogeneous of degree -1 in u, and u
    # after the first pass is y / ||y|| whatever ts was (:900-907) -- so ONE pass per component (a NaN sweep less: 5 of
    # 17 ms per component at p = 10^6); W_non_normal_ takes the factor ||y|| the second pass would have applied.
    one_pass = (q == 1 and eng.opts.q1_fixed_point and eng.sparse and not resp.sparse and max_iter >= 1
                and not resident and k_start == 0)
    if one_pass:
        _plan(eng, solver="nipals, one response, fixed point after one pass: host-driven loop (missing values)")
    w_scale = float(np.linalg.norm(u0)) if one_pass else 1.0
    for k in range(k_start, K):
        u = u0
        diff, run, ts_old = 1.0, 1, None
        prev_diff, stalls = None, 0
        while diff > max_tol:
            uu = u.T.dot(u).item()
            if run > 1:
                t_raw, ssq = eng.sweep(u, 1.0 / uu, ("Wn", k))
            nu = np.sqrt(ssq)                               # ||w~_b||           (:860)
            Tm = t_raw / nu                                 # t_b = X_b w_b      (:875-877)
            sw = Tm.T.dot(u) / uu                           # (:879)
            sw = sw / np.linalg.norm(sw)
            ts = Tm.dot(sw)                                 # (:882)
            ts = ts / np.linalg.norm(ts)
            if run > 1:
                prev_diff = diff if run > 2 else None
                diff = _norm(ts_old - ts, norm)             # (:887)
            ts_old = ts.copy()
            v = resp.loadings(ts)                           # (:899)
            u = resp.scores(v)                              # (:911-913)
            run += 1
            if one_pass:
                diff = 0.0                                  # (at its fixed point: the second pass would reproduce ts)
                break
            if run - 1 >= max_iter and diff > max_tol:
                warnings.warn(f"NIPALS component {k}: no convergence to max_tol={max_tol} after {max_iter} "
                              f"iterations (last diff {diff:.3e}); the reference would loop forever here.")
                break
            floor_hit = False
            if q == 1 and not resp.sparse and run - 1 >= 2 and diff > max_tol:
                floor_hit, stalls = q1_noise_floor(run - 1, diff, prev_diff, stalls)
            if floor_hit:
                warnings.warn(f"NIPALS component {k}: ||ts_old - ts|| = {diff:.3e} stays above max_tol={max_tol} after "
                              f"{run - 1} iterations -- with one response the iteration has been at its fixed point "
                              f"since the first pass, the difference is rounding noise (y is nearly exhausted); "
                              f"stopped here, the reference would keep iterating (nipals_converged_[{k}] = False).")
                break
        iters.append(run - 1)
        converged.append(bool(diff <= max_tol))
        last = (k == K - 1)

        def stream_out(k=k, nu=nu):
            # runs on the host WHILE the deflation pass of component k executes: W~[:, k] is final (its last
            # sweep has completed) and so is P[:, k-1]; W_b[:, k] = w~_b / ||w~_b|| (:860) is formed here too
            eng.fetch_col("Wn", k, Wn_h)
            for b in range(B):
                r0, r1 = eng.block_offsets[b], eng.block_offsets[b + 1]
                np.divide(Wn_h[r0:r1, k], nu[b], out=W_h[r0:r1, k])
            if w_scale != 1.0:
                Wn_h[:, k] *= w_scale                       # w~(y / ||y||) = ||y|| w~(y): the raw weights of the pass left out
            if k > 0:
                eng.fetch_col("P", k - 1, P_h)

        # loadings (:928, no division: ts is unit length) + deflation (:969) + next first sweep
        ssq_p, t_next, ssq_next = eng.deflate_sweep(ts, 1.0, not last, ("P", k),
                                                    None if last else u0, alpha0,
                                                    None if last else ("Wn", k + 1), overlap=stream_out)
        Ts[:, k:k + 1] = ts; U[:, k:k + 1] = u; V[:, k:k + 1] = v; A[:, k:k + 1] = sw ** 2
        nu_all[:, k] = nu
        for b in range(B):
            T[b][:, k] = Tm[:, b]
        if calc_all:                                        # (:932-964), analytic (ts^T ts)(p^T p)
            tt = ts.T.dot(ts).item()
            evx.append(tt * ssq_p.sum() / varx)
            evy.append(tt * v.T.dot(v).item() / vary)
            evxb[:, k] = tt * ssq_p / varxb
            Acorr[:, k] = _bip_corrected(sw ** 2, sizes)
        t_raw, ssq = t_next, ssq_next
    if not resident:
        res, finish = FitResult(), None
    res.update(Ts_=Ts, T_=T, U_=U, V_=V, A_=A, nipals_iterations_=iters, nipals_converged_=converged)
    if calc_all:
        res.update(explained_var_x_=evx, explained_var_y_=evy, explained_var_xblocks_=evxb, A_corrected_=Acorr)
    else:
        res.update(explained_var_x_=[], explained_var_y_=[], explained_var_xblocks_=np.empty((B, 0)),
                   A_corrected_=np.empty((B, 0)))
    if callable(finish):
        with eng._timed("wait_R_beta"):
            finish()
        with eng._timed("wait_streamed_columns"):
            out["wait_copies"]()                            # streamed columns of W~, W, P complete
    else:
        # ||W~[:, k]||^2 = sum_b ||w~_b||^2 of the last iteration (already summed over shards)  (:986-987)
        eng.fetch_col("P", K - 1, P_h)
        _finalise_device(eng, res, "Wn", K, np.sqrt((nu_all ** 2).sum(axis=0)), nu_all, V, W_host=W_h, P_host=P_h)
    res.W_non_normal_ = _split(Wn_h, eng.block_offsets)
    return res


# ----------------------------------------------------------------------------------- UNIPALS
def _fit_unipals_wide_multi(eng, Y, K, standardize):
    """UNIPALS, p > n, several responses (mbpls.py:481-572) without ever deflating X.  The deflation :535 is
    X_k = (I - ts ts^T) X_{k-1} = P X_0 with P = I - Ts Ts^T (successive ts are orthonormal), so
      S-factor :489   X_k X_k^T Y = P X_0 (X_0^T (P Y))         -- maintained (_ResponseProducts): no pass
      weights  :503   X_k^T u = X_0^T (P u),  block scores :515-517  X_k,b w_b = P X_0,b w_b,
      loadings :520   X_k^T ts / ts^T ts = X_0^T ts / ts^T ts     (ts is orthogonal to the earlier scores)
    and only the loadings need a pass -- the fused sweep with v = ts, which also returns X_0,b p_b for the maintained
    products: ONE pass per component instead of four plus a rewrite."""
    n, B = eng.n, eng.nblocks
    q = Y.shape[1]
    sizes = _global_sizes(eng)
    eng.slot("Wc", K); eng.slot("P", K); eng.slot("S", q + 1)      # S = [X^T (P Y) | the loadings of the component]
    eng.use_lazy_deflation(K, "P", prefer="lazy")           # (never deflates: the never-writes-X passes when they apply)
    cols = _StreamedColumns(eng, K, "Wc", q=q)
    prod = _ResponseProducts(eng, Y, standardize)
    varxb = prod.varxb
    varx = varxb.sum()
    vary = (Y ** 2).sum()
    F = _psd_factor(Y.T.dot(Y))
    Ts = np.empty((n, K), order="F")        # (its leading columns are contiguous operands of the projections)
    U = np.empty((n, K)); V = np.empty((q, K)); A = np.empty((B, K))
    T = [np.empty((n, K)) for _ in range(B)]
    blk_nrm = np.empty((B, K)); tot_nrm = np.empty(K)
    evx, evy, evxb, Acorr = [], [], np.empty((B, K)), np.empty((B, K))
    for k in range(K):
        Tp = Ts[:, :k]
        M = prod.total()
        M = M - Tp.dot(Tp.T.dot(M))                         # X_k X_k^T Y (:489)
        ts = _tlsv(M.dot(F))                                # (:489-493)
        tt = ts.T.dot(ts).item()
        v = Y.T.dot(ts) / tt                                # (:496)
        u = Y.dot(v); un = np.linalg.norm(u); u = u / un    # (:499-500)
        # w_raw = X_k^T u = X_0^T P Y v / ||Y v|| = S v / ||Y v|| (:503), X_0,b w_raw,b, ||w_raw,b||^2: maintained products
        t_raw, ssq = prod.combination(v / un, ("Wc", k))
        # the pass of the component: p = X_0^T ts / ts^T ts (:520); the previous component's columns leave meanwhile
        ssq_p = prod.component(ts, 1.0 / tt, ("P", k))
        t_raw = t_raw - Tp.dot(Tp.T.dot(t_raw))             # X_k,b w_raw,b = P X_0,b w_raw,b (:515-517)
        tot = ssq.sum()
        blk_nrm[:, k] = np.sqrt(ssq); tot_nrm[k] = np.sqrt(tot)
        cols.component_done(k, blk_nrm[:, k])
        A[:, k] = ssq / tot                                 # ||w_b||^2 of the unit-length w (:512)
        Tm = t_raw / np.sqrt(ssq)
        Ts[:, k:k + 1] = ts; U[:, k:k + 1] = u; V[:, k:k + 1] = v
        for b in range(B):
            T[b][:, k] = Tm[:, b]
        evx.append(tt * ssq_p.sum() / varx)                 # (:521-533)
        evxb[:, k] = tt * ssq_p / varxb
        evy.append(tt * v.T.dot(v).item() / vary)           # (:538-540)
        Acorr[:, k] = _bip_corrected(A[:, k], sizes)        # (:542-553)
    res = FitResult(Ts_=Ts, T_=T, U_=U, V_=V, A_=A, explained_var_x_=evx, explained_var_y_=evy,
                    explained_var_xblocks_=evxb, A_corrected_=Acorr)
    cols.finish()
    _finalise_device(eng, res, "Wc", K, tot_nrm, blk_nrm, V, W_host=cols.W, P_host=cols.P)    # (:569-572)
    return res


def fit_unipals(eng, Y, K, standardize, keep_raw_weights=False):
    """mbpls.py:384-574 (calc_all is ignored by the reference here).
    p>n (:481-572): ts = tlsv((X (X^T Y)) F(Y^T Y));  n>=p (:388-479): w = tlsv(X^T Y).
    keep_raw_weights (device-resident one-response queue only): the un-normalised weight columns are streamed to the
    host as well and returned as ``_W_raw_host`` (the NIPALS fixed-point route hands them out as W_non_normal_)."""
    n, B = eng.n, eng.nblocks
    q = Y.shape[1]
    # One response: w = tlsv(X^T y) and ts = X w / ||X w|| of the n >= p branch (:396-417) are the X^T y / ||.|| and
    # X (X^T y) / ||.|| of the p > n branch (:489-503) -- the same component up to the sign convention of the SVD --
    # so tall data takes the one-pass-per-component path as well (device-resident queue, X never written).
    # (the total width is a collective under column shards: only asked where the branch depends on it)
    p_total = eng.p if q == 1 else int(eng.comm.allreduce_host(np.array([float(eng.p)]))[0])
    wide = p_total > n or q == 1
    if wide and q > 1 and not eng.sparse and eng.opts.unipals_nodeflate:
        _plan(eng, solver="unipals: several responses, one pass per component (maintained products)")
        return _fit_unipals_wide_multi(eng, Y, K, standardize)
    y0 = Y[:, 0:1]
    eng.slot("S", max(q, 1))
    if wide:
        # one response: every pass is a sweep / deflate_sweep, lazy deflation applies.  Several responses: X (X^T Y)
        # goes through the many-right-hand-side products, which need the deflated matrix itself -> explicit.
        # (the n >= p branch mixes in xt_v / x_w on the deflated X as well: explicit)
        eng.use_lazy_deflation(K, "P", prefer="lazy" if q == 1 else "explicit")
    fuse = wide and q == 1
    # One response: a component is ONE pass and everything between two passes is n x B algebra -- the whole fit is queued
    # on the device (engine.unipals_resident), the host reads the results once.  Its first pass (statistics + first
    # sweep) needs nothing but the matrix, the start vector and the S slot: launched NOW, so the GPU is at work while
    # the host allocates the other slots and the host copies of the attributes (0.1 ms of every fit)
    go_resident = fuse and eng.resident_supported(q, K, 0)
    first = None
    if go_resident:
        first = eng.deferred_first_pass(standardize, y0, 1.0, ("S", 0))
        first()
    else:
        sizes = _global_sizes(eng)
    eng.slot("Wc", K)
    eng.slot("P", K)
    # n >= p: the stored matrix is only read (deflation applied to the small operands, below) once a rewrite of it costs
    # more than the n x q projections on the host: from about half a gigabyte (a 160 MB matrix -- the paper's shape (a)
    # -- is deflated in 50 us, the projections cost 0.2 ms)
    nodeflate_tall = (not wide) and eng.opts.unipals_nodeflate and 8.0 * n * p_total >= 5e8
    vary = (Y ** 2).sum()
    Ts = np.empty((n, K), order="F"); U = np.empty((n, K)); V = np.empty((q, K)); A = np.empty((B, K))
    T = [np.empty((n, K)) for _ in range(B)]
    blk_nrm = np.empty((B, K)); tot_nrm = np.empty(K)
    evx, evy, evxb, Acorr = [], [], np.empty((B, K)), np.empty((B, K))
    # p > n: W_ and P_ are streamed to the host column by column while the deflation passes run (as in NIPALS)
    W_h = _host_array(eng, (eng.p, K)) if wide else None
    P_h = _host_array(eng, (eng.p, K)) if wide else None
    if go_resident:
        eng.host_array((eng.p, K), reserve="R")
        eng.host_array((eng.p, q), reserve="beta")
        Wraw_h = _host_array(eng, (eng.p, K)) if keep_raw_weights else None
        for attempt in range(2):
            # (first pass already queued; offset data: queued again, first pass included, with centring passes)
            out = eng.unipals_resident(Y, K, "Wc", "S", W_h, P_h, first if attempt else None, Wraw_h)
            if out.get("reason") != "offset":
                break
        resident = out.get("reason") != "barrier"
        # (the block sizes of the whole problem are a collective under column shards, and its read-back waits for the
        # stream: asked only now, with the queue drained -- behind the first pass it kept the host from building the
        # queue for the length of that pass)
        sizes = _global_sizes(eng)
    else:
        resident = False
    _plan(eng, solver=("unipals: device-resident queue, one pass per component" if resident else
                       "unipals: host-driven loop" + (", one pass per component" if fuse else "")))
    if resident:
        _plan(eng, r_and_beta=out.get("r_route"))
        varxb = out["ssq_x"]
        if not np.all(np.isfinite(varxb)):
            raise ValueError("Input X contains NaN, infinity or a value too large for dtype('float64').")
        varx = varxb.sum()
        Ts, U = out["Ts"], out["U"]                         # the downloaded arrays are the attributes
        T = [out["Tb"][:, b, :].T for b in range(B)]
        out["wait_scores"]()                                # (they arrive through the copier thread, before the last pass)
        # the recorded scalars are the attributes, for all components at once
        V[:] = out["V"]
        ssq, ssq_p = out["scal"][:, 0, :], out["scal"][:, 2, :]      # K x B each
        A[:] = out["scal"][:, 1, :].T
        blk_nrm[:] = np.sqrt(ssq).T
        tot_nrm[:] = np.sqrt(ssq.sum(axis=1))
        tt = np.einsum("ik,ik->k", Ts, Ts)
        evx = list(tt * ssq_p.sum(axis=1) / varx)           # (:521-533)
        evxb[:] = (tt[:, None] * ssq_p / varxb).T
        evy = list(tt * np.einsum("jk,jk->k", V, V) / vary)  # (:538-540)
        Acorr[:] = _bip_corrected_all(A, sizes)             # (:542-553)
        res = FitResult(Ts_=Ts, T_=T, U_=U, V_=V, A_=A, explained_var_x_=evx, explained_var_y_=evy,
                        explained_var_xblocks_=evxb, A_corrected_=Acorr)
        if out.get("finalised"):
            _finish_resident(eng, res, out, W_h, P_h)()     # (:569-572, launched by the queue)
        else:
            _finalise_device(eng, res, "Wc", K, tot_nrm, blk_nrm, V, W_host=W_h, P_host=P_h, C=out.get("PtW"))
        out["wait_copies"]()                                # streamed columns of W, P complete
        if keep_raw_weights:
            res._W_raw_host = Wraw_h
        return res
    varxb, m_raw, g_ssq = eng.prepare(standardize, v=y0 if fuse else None, alpha=1.0, g_slot=("S", 0) if fuse else None)
    yy = y0.T.dot(y0).item()
    varx = varxb.sum()
    F = _psd_factor(Y.T.dot(Y))
    for k in range(K):
        if wide:
            if fuse:
                M = m_raw.sum(axis=1, keepdims=True)        # X (X^T y) = sum_b X_b g_b
            elif getattr(eng, "_lazy_pslot", None) is None:
                # X (X^T Y) for all q responses in two reads of the (explicitly deflated) X: matrix-core xt_v, x_w
                eng.xt_v(Y, 1.0, "S", 0)
                M = eng.x_w_slot("S", 0, q)
            else:
                M = np.empty((n, q))
                for c in range(q):
                    tr, _ = eng.sweep(Y[:, c:c + 1], 1.0, None)
                    M[:, c] = tr.sum(axis=1)
            ts = _tlsv(M.dot(F))                            # (:489-493)
            v = Y.T.dot(ts) / ts.T.dot(ts)                  # (:496)
            u = Y.dot(v); u = u / np.linalg.norm(u)         # (:499-500)
            if fuse:
                # one response: u = Y v / ||Y v|| is c y with a scalar c, so the weights w_raw = X^T u (:503) and the
                # block scores X_b w_raw_b (:515-517) are c times the X^T y and X_b (X^T y)_b that the fused pass just
                # produced for M = X (X^T y) -- the same products, not taken twice
                c = u.T.dot(y0).item() / yy
                eng.right_mul(("S", 0, 1), np.array([[c]]), ("Wc", k))
                # (stream_out below copies this column on the COPY stream while the deflation pass runs: ordered behind
                # this small kernel by the fence deflate_sweep records -- without one the copy could overtake it, seen
                # with two ranks sharing one GPU in the randomised sharded sweep of round 5)
                t_raw, ssq = c * m_raw, c * c * g_ssq
            else:
                t_raw, ssq = eng.sweep(u, 1.0, ("Wc", k))   # w_raw = X^T u (:503), t~_b = X_b w_raw_b
        else:
            # n >= p (:388-479).  The deflation :470 is X_k = P X_0 with P = I - Ts Ts^T (unit, mutually orthogonal
            # scores), so X_k^T Y = X_0^T (P Y), X_k,b w_b = P X_0,b w_b and p = X_k^T ts = X_0^T ts: three READS of the
            # stored matrix per component and no rewrite (MBPLS_UNIPALS_NODEFLATE=0: deflate in place as the reference)
            Tp = Ts[:, :k]
            lazy = nodeflate_tall
            eng.xt_v(Y - Tp.dot(Tp.T.dot(Y)) if (lazy and k) else Y, 1.0, "S", 0)
            C = eng.fetch("S")[:, :q]
            w = _tlsv(C, eng.comm.allreduce_host if eng.comm.is_distributed else None)   # (:396-400): S = C C^T is symmetric
            eng.put_col("Wc", k, w)
            tb = eng.x_w_slot("Wc", k, 1, per_block=True)
            t_raw = np.hstack(tb)
            if lazy and k:
                t_raw = t_raw - Tp.dot(Tp.T.dot(t_raw))
            ssq = eng.comm.allreduce_host(np.array([float((w[eng.block_offsets[b]:eng.block_offsets[b + 1]] ** 2).sum())
                                                    for b in range(B)]))
            ts = t_raw.sum(axis=1, keepdims=True)           # X w (:416)
            ts = ts / np.linalg.norm(ts)
            v = Y.T.dot(ts) / ts.T.dot(ts)
            u = Y.dot(v); u = u / np.linalg.norm(u)
        tot = ssq.sum()
        blk_nrm[:, k] = np.sqrt(ssq); tot_nrm[k] = np.sqrt(tot)
        a = ssq / tot                                       # ||w_b||^2 of the unit-length w (:512)
        Tm = t_raw / np.sqrt(ssq)                           # t_b = X_b w_b/||w_b||  (:517)
        tt = ts.T.dot(ts).item()
        last = (k == K - 1)
        nxt = fuse and not last

        def stream_out(k=k, nrm=np.sqrt(ssq)):
            # host work WHILE the deflation pass runs: the weight column k is final, so is P[:, k-1]
            eng.fetch_col("Wc", k, W_h)
            for b in range(B):
                r0, r1 = eng.block_offsets[b], eng.block_offsets[b + 1]
                W_h[r0:r1, k] /= nrm[b]                     # w_b / ||w_b|| (:510-512)
            if k > 0:
                eng.fetch_col("P", k - 1, P_h)

        ssq_p, m_raw, g_ssq = eng.deflate_sweep(ts, 1.0 / tt, (not last) and (wide or not nodeflate_tall),
                                            ("P", k), y0 if nxt else None, 1.0,
                                            ("S", 0) if nxt else None,   # p (:520), X - ts p^T (:535)
                                            overlap=stream_out if wide else None)
        Ts[:, k:k + 1] = ts; U[:, k:k + 1] = u; V[:, k:k + 1] = v; A[:, k] = a
        for b in range(B):
            T[b][:, k] = Tm[:, b]
        evx.append(tt * ssq_p.sum() / varx)                 # (:521-533)
        evxb[:, k] = tt * ssq_p / varxb
        evy.append(tt * v.T.dot(v).item() / vary)           # (:538-540)
        Acorr[:, k] = _bip_corrected(a, sizes)              # (:542-553)
    res = FitResult(Ts_=Ts, T_=T, U_=U, V_=V, A_=A, explained_var_x_=evx, explained_var_y_=evy,
                    explained_var_xblocks_=evxb, A_corrected_=Acorr)
    if wide:
        eng.fetch_col("P", K - 1, P_h)
    _finalise_device(eng, res, "Wc", K, tot_nrm if wide else np.ones(K), blk_nrm, V,   # (:569-572 / :476-479)
                     W_host=W_h, P_host=P_h)
    return res


# ----------------------------------------------------------------------------------- KERNEL
def _kernel_extras(eng, res, Y, P, V, Ts, varxb, wslot, wnorm2):
    """calc_all outputs of the KERNEL method (mbpls.py:653-689, :740-802) without the second,
    X-deflating sweep: deflation-free block scores T_b = X_b W_b - Ts triu(P_b^T W_b, 1).
    ``wslot`` holds columns proportional to W_concat_ on the device, ``wnorm2[b, k]`` their squared
    per-block norms (already summed over shards)."""
    B = eng.nblocks
    K = Ts.shape[1]
    sizes = _global_sizes(eng)
    offs = eng.block_offsets
    Wraw = eng.fetch(wslot)[:, :K]
    W = [w / np.sqrt(wnorm2[b]) for b, w in enumerate(_split(Wraw, offs))]   # (:748 / :607)
    A = wnorm2 / wnorm2.sum(axis=0)                         # ||W_concat_[J_b, k]||^2 (:750 / :609)
    XW = eng.x_w_slot(wslot, 0, K, per_block=True)
    T = []
    for b in range(B):
        PtW = eng.comm.allreduce_host(P[offs[b]:offs[b + 1]].T.dot(W[b]))
        T.append(XW[b] / np.sqrt(wnorm2[b]) - Ts.dot(np.triu(PtW, 1)))
    varx = varxb.sum()
    vary = (Y ** 2).sum()
    tt = (Ts ** 2).sum(axis=0)
    pp_b = eng.comm.allreduce_host(np.array([(P[offs[b]:offs[b + 1]] ** 2).sum(axis=0) for b in range(B)]))
    res.update(T_=T, W_=W, A_=A,
               A_corrected_=np.column_stack([_bip_corrected(A[:, k], sizes) for k in range(K)]),
               explained_var_x_=list(tt * pp_b.sum(axis=0) / varx),
               explained_var_xblocks_=tt * pp_b / varxb[:, None],
               explained_var_y_=list(tt * (V ** 2).sum(axis=0) / vary))


def _kernel_extras_device(eng, res, Y, V, Ts, varxb, wslot, wnorm2, XW=None):
    """_kernel_extras with the p x K operands left on the device: slot "P" holds the loadings, slot "Wblk"
    the per-block normalised weights (both produced by _finalise_device); K x K cross products per block.
    ``XW``: the per-block products X_b wslot_b (n x K each) when the caller has them already."""
    B = eng.nblocks
    K = Ts.shape[1]
    sizes = _global_sizes(eng)
    offs = eng.block_offsets
    A = wnorm2 / wnorm2.sum(axis=0)                         # ||W_concat_[J_b, k]||^2 (:750)
    if XW is None:
        XW = eng.x_w_slot(wslot, 0, K, per_block=True)
    T, pp_b = [], np.empty((B, K))
    for b in range(B):
        rows = (offs[b], offs[b + 1])
        PtW = eng.cross(("P", 0, K), ("Wblk", 0, K), rows=rows)
        T.append(XW[b] / np.sqrt(wnorm2[b]) - Ts.dot(np.triu(PtW, 1)))
        pp_b[b] = np.diag(eng.cross(("P", 0, K), ("P", 0, K), rows=rows))
    varx = varxb.sum()
    vary = (Y ** 2).sum()
    tt = (Ts ** 2).sum(axis=0)
    res.update(T_=T, A_=A,
               A_corrected_=np.column_stack([_bip_corrected(A[:, k], sizes) for k in range(K)]),
               explained_var_x_=list(tt * pp_b.sum(axis=0) / varx),
               explained_var_xblocks_=tt * pp_b / varxb[:, None],
               explained_var_y_=list(tt * (V ** 2).sum(axis=0) / vary))


# Sustained rates the route of KERNEL p > n is chosen with (one MI355X): the f64 matrix-core Gram
# and a read-only column pass.  Only their ratio matters.
_GRAM_FLOPS = 71.0e12
_PASS_BYTES = 6.5e12


_KERNEL_STREAMED_COLUMNS = 1 << 18


def _kernel_on_products(eng, n, K, q, p_total):
    """KERNEL p > n: the association matrix X X^T (n (n + 1) p flop on the f64 matrix cores, then O(n^2) per
    component) or the maintained thin products of _kernel_scores_on_products (K + 1 passes over X, + 1 or 2 to set up
    several responses)?  The Gram only wins while n(n+1)/71e12 < passes * 8n/6.5e12, i.e. n below ~ 87 (K + 3); at
    BASELINE config[3] (n = 8192, K = 20) the passes are 4 x faster.  And it only wins on narrow matrices: with few
    samples the fit is the PCIe time of its p x K attributes, which the products route sends out column by column
    beside its passes while the Gram route has them all at the very end (n = 100, p = 10^7, K = 4: 56 against 33 ms;
    n = 60, p = 10^6: 5.5 / 3.1; n = 200, p = 2 10^5: 2.8 / 2.6; n = 250, p = 10^5, K = 10: 2.9 / 3.5 --
    profiles/r06_wide_few_samples.txt), so from 2^18 columns on the products are taken whatever n.
    MBPLS_KERNEL_GRAM = 1 | 0 forces either; the choice is the same on every rank of a column-sharded fit (n, K, q and the
    total width are)."""
    if eng.sparse or getattr(eng, "rowmask", None) is not None:
        return False
    forced = eng.opts.kernel_gram
    if forced is not None:
        return forced == "0"
    if p_total >= _KERNEL_STREAMED_COLUMNS:
        return True
    passes = K + 1 + (0 if q == 1 else (1 if q == 2 else 2))
    return n * (n + 1.0) / _GRAM_FLOPS + 3 * 8.0 * n / _PASS_BYTES > passes * 8.0 * n / _PASS_BYTES


def _kernel_scores_on_products(eng, Y, K, standardize):
    """The component loop of KERNEL p > n (mbpls.py:704-728) WITHOUT the association matrix.  With orthonormal scores
    the two-sided deflations :723-724 are AS_X,k = P X X^T P and AS_Y,k = P Y Y^T P, P = I - Ts Ts^T, so
        S_k = AS_X,k AS_Y,k = [P X (X^T (P Y))] (P Y)^T                                   (:725)
    has rank <= q and its first left singular vector (:712-715) is that of the n x q matrix M L with
    M = P X (X^T (P Y)) and L L^T = (P Y)^T (P Y).  X^T (P Y) (p x q) and the per-block X_b (X^T (P Y))_b (n x q) are the
    maintained products of _ResponseProducts -- the one pass a component costs is the fused sweep with v = ts, which
    returns X^T ts (a column of :734 already) and X_b (X^T ts)_b.  The weights need no pass either:
        u = AS_Y,k ts / ||.|| = (P Y) c / ||.||,  c = (P Y)^T ts  (:718-719)   =>   X^T u = [X^T (P Y)] c / ||.||   (:731)
    and X_b (X^T u)_b = [X_b (X^T (P Y))_b] c / ||.|| are the block products of calc_all (:752).  X is never written.
    Leaves the raw weights X^T U_ in slot "UT" and the loadings X^T Ts (Ts^T Ts)^+ (:734) in slot "P"; returns
    (Ts, U, ||X_b||^2, per-block ||(X^T U_)_b||^2 (B x K), [X_b (X^T U_)_b] (n x K per block), the columns streamed to
    the host during the fit (block weights, W_concat_, loadings), the host loadings or None)."""
    n, B = eng.n, eng.nblocks
    q = Y.shape[1]
    eng.slot("UT", K); eng.slot("Pl", K); eng.slot("S", q + 1)     # S = [X^T (P Y) | X^T ts of the component at hand]
    eng.use_lazy_deflation(K, "Pl", prefer="lazy")        # (never deflates: the never-writes-X passes when they apply)
    _plan(eng, solver="kernel p > n on maintained thin products X^T (P Y), X_b (X^T (P Y))_b: one pass over X per "
                      "component, no association matrix")
    # finished columns leave beside the passes (copy stream; block / column normalisation on the idle host): X^T u and
    # X^T ts / ts^T ts of component k while the pass of component k + 1 runs -- the end of the fit fetches R_, beta_ only
    cols = _StreamedColumns(eng, K, "UT", pslot="Pl", q=q, concat=True)
    prod = _ResponseProducts(eng, Y, standardize)
    varxb = prod.varxb
    Yd = Y.copy()
    Ts = np.empty((n, K), order="F"); U = np.empty((n, K))
    wnorm2 = np.empty((B, K)); tts = np.empty(K)
    XW = [np.empty((n, K)) for _ in range(B)]
    for k in range(K):
        Tp = Ts[:, :k]
        M = prod.total()
        M = M - Tp.dot(Tp.T.dot(M))                         # P X X^T P Y (AS_X,k Y_d; :723)
        ts = _tlsv(M.dot(_psd_factor(Yd.T.dot(Yd))))       # (:712-715): S = AS_X,k Y_d Y_d^T
        if k:       # (an exhausted response leaves M at rounding level: keep the scores orthonormal all the same)
            ts = ts - Tp.dot(Tp.T.dot(ts))
            ts = ts / np.linalg.norm(ts)
        c = Yd.T.dot(ts)
        u = Yd.dot(c); un = np.linalg.norm(u); u = u / un   # (:718-719)
        t_raw, wnorm2[:, k] = prod.combination(c / un, ("UT", k))    # X^T u (:731), X_b (X^T u)_b, ||(X^T u)_b||^2
        tts[k] = ts.T.dot(ts).item()
        # the pass: X^T ts / ts^T ts; X^T (P Y), X_b (..)_b without ts (the previous component's columns leave meanwhile)
        prod.component(ts, 1.0 / tts[k], ("Pl", k))
        cols.component_done(k, np.sqrt(wnorm2[:, k]))
        Yd = Yd - ts.dot(ts.T.dot(Yd))                      # AS_Y = D AS_Y D (:724)
        Ts[:, k:k + 1] = ts; U[:, k:k + 1] = u
        for b in range(B):
            XW[b][:, k] = t_raw[:, b]
    G = np.linalg.pinv(Ts.T.dot(Ts))
    eng.slot("P", K)
    C = tts[:, None] * G
    eng.right_mul(("Pl", 0, K), C, ("P", 0))                # P = X^T Ts (Ts^T Ts)^+ (:734), on the device for P^T W below
    cols.finish()
    # the streamed columns ARE the loadings when the scores are orthonormal (C = I to rounding: always, unless a
    # response was exhausted): no second copy of a p x K array then
    P_host = cols.P if float(np.max(np.abs(C - np.eye(K)))) <= 1e-13 else None
    return Ts, U, varxb, wnorm2, XW, cols, P_host


def fit_kernel(eng, Y, K, standardize, calc_all=True):
    """mbpls.py:576-807.  p>n (:694-805): association matrix A = X X^T on the matrix cores, then the
    component loop runs on n x q thin factors: the reference's D A D (D = I - ts ts^T, :722-725) is
    applied implicitly, A_k Y_d = D_k..D_1 A_0 (D_1..D_k Y_d), one A_0 @ (n x q) product per component."""
    n, B = eng.n, eng.nblocks
    q = Y.shape[1]
    p_total = int(eng.comm.allreduce_host(np.array([float(eng.p)]))[0])
    products = p_total > n and _kernel_on_products(eng, n, K, q, p_total)
    if products:
        pass                                                # (prepared by the route itself: statistics + first sweep)
    elif p_total > n and eng.stats_only_ok():
        # p > n: every X-sized operation below is a matrix product (Gram, X^T [U Ts], X_b W_b) that can standardise
        # its operand in registers -- statistics only, X is neither copied nor written
        varxb = eng.prepare_stats(standardize)
    else:
        varxb, _, _ = eng.prepare(standardize)
    res = FitResult()
    empty_blocks = dict(W_=[np.empty((sz, 0)) for sz in eng.block_sizes], T_=[np.empty((n, 0)) for _ in range(B)],
                        A_=np.empty((B, 0)), A_corrected_=np.empty((B, 0)), explained_var_x_=[],
                        explained_var_y_=[], explained_var_xblocks_=np.empty((B, 0)))
    if p_total > n and products:
        Ts, U, varxb, wnorm2, XWraw, cols, P_host = _kernel_scores_on_products(eng, Y, K, standardize)
        G = np.linalg.pinv(Ts.T.dot(Ts))
        V = Y.T.dot(Ts).dot(G)                          # (:735)
        cn = np.sqrt(wnorm2.sum(axis=0))
        res.update(Ts_=Ts, U_=U, V_=V)
        # R_, beta_ (:737-738) and the block weights on the device; of the p x K arrays only R_ and beta_ are fetched here
        _finalise_device(eng, res, "UT", K, cn, np.sqrt(wnorm2), V, fetch=False)
        wait_r, wait_b = eng.fetch_async("R")[1], eng.fetch_async("beta")[1]
        if calc_all:        # X_b W_b came with the maintained products: no further pass over X
            _kernel_extras_device(eng, res, Y, V, Ts, varxb, "UT", wnorm2, XW=XWraw)
        offs = eng.block_offsets
        res.R_, res.beta_ = wait_r(), wait_b()
        res.W_ = _split(cols.W, offs)
        res.P_ = _split(P_host if P_host is not None else eng.fetch("P"), offs)
        res.W_concat_ = cols.Wc
        if not calc_all:
            res.update(empty_blocks)
        return res
    if p_total > n:
        _plan(eng, solver="kernel p > n: association matrix X X^T on the f64 matrix cores, O(n^2) rank updates")
        eng.gram()                                          # AS_X (:704)
        # the host is idle while the matrix cores build the association matrix: the p x K attributes' host arrays are
        # allocated now (their pages mapped by the helper thread: the copies at the end then run at PCIe speed instead of
        # page-fault speed, 4 x 320 MB at BASELINE config[3]: 80 -> 45 ms) and what a refit still owes the previous
        # fit's attributes is released now (53 ms of page-table teardown at that size)
        for name, cols in (("R", K), ("beta", q), ("Wblk", K), ("P", K), ("Wc", K)):
            eng.host_array((eng.p, cols), reserve=name)
        eng.run_deferred(all_of_it=True)
        Yd = Y.copy()
        Ts = np.empty((n, 0)); U = np.empty((n, 0))
        for k in range(K):
            Z = Yd
            for i in range(k):                              # D_1..D_k Y_d (no-op up to round-off)
                ti = Ts[:, i:i + 1]
                Z = Z - ti.dot(ti.T.dot(Z))
            AZ = eng.gram_apply(Z)
            for i in range(k):
                ti = Ts[:, i:i + 1]
                AZ = AZ - ti.dot(ti.T.dot(AZ))
            ts = _tlsv(AZ.dot(_psd_factor(Yd.T.dot(Yd))))  # (:712-715): S = A_k Y_d Y_d^T
            u = Yd.dot(Yd.T.dot(ts)); u = u / np.linalg.norm(u)   # (:718-719)
            Yd = Yd - ts.dot(ts.T.dot(Yd))                  # AS_Y = D AS_Y D (:724)
            Ts = np.hstack((Ts, ts)); U = np.hstack((U, u))
        eng.slot("UT", 2 * K)
        ssq = eng.xt_v(np.hstack((U, Ts)), 1.0, "UT", 0, want_ssq=True)   # X^T [U Ts] in one call (:731, :734)
        # post-processing on the device: only K x K matrices visit the host, every p x K attribute is
        # fetched once
        wnorm2 = ssq[:, :K]                             # per-block ||(X^T U)_b||^2, all shards
        cn = np.sqrt(wnorm2.sum(axis=0))
        G = np.linalg.pinv(Ts.T.dot(Ts))
        V = Y.T.dot(Ts).dot(G)                          # (:735)
        eng.slot("P", K)
        eng.right_mul(("UT", K, K), G, ("P", 0))        # P = X^T Ts (Ts^T Ts)^+ (:734)
        eng.slot("Wc", K)
        eng.right_mul(("UT", 0, K), np.diag(1.0 / cn), ("Wc", 0))     # column-normalised weights (:733)
        res.update(Ts_=Ts, U_=U, V_=V)
        # R_, beta_ (:737-738), W_b, P_ on the device; their D2H (five p x K arrays) runs on the copy stream
        # in a helper thread while the block-score passes of calc_all stream X on the compute stream
        fetch_all = _finalise_device(eng, res, "UT", K, cn, np.sqrt(wnorm2), V, fetch=False)
        if calc_all and eng.has_device_helpers:
            import threading
            eng.sync()                                  # every slot the helper reads is final
            err = []

            def helper():
                try:
                    fetch_all(overlapped=True)
                    res.W_concat_ = eng.fetch("Wc", overlapped=True)
                except BaseException as e:              # re-raised on the calling thread
                    err.append(e)

            th = threading.Thread(target=helper)
            th.start()
            try:
                _kernel_extras_device(eng, res, Y, V, Ts, varxb, "UT", wnorm2)
            finally:
                th.join()
            if err:
                raise err[0]
            return res
        fetch_all()
        res.W_concat_ = eng.fetch("Wc")
        if calc_all:
            _kernel_extras_device(eng, res, Y, V, Ts, varxb, "UT", wnorm2)
        else:
            res.update(empty_blocks)
        return res
    # ---- n >= p (:580-692).  The reference deflates the p x p matrices with dense products, D^T VAR D with
    # D = I - w pl (:630-633): two p^3 GEMMs per component.  VAR is symmetric and pl = w^T VAR / den, so
    #     D^T VAR D = VAR - g g^T / den,   g = VAR w,   den = w^T g          (a symmetric rank-1 update)
    #     D^T COV   = COV - pl^T (w^T COV)
    # and VAR_k = VAR_0 - sum_i g_i g_i^T / den_i never has to be formed: g_k = VAR_0 w_k - sum_i g_i (g_i^T w_k) / den_i
    # costs ONE product with the p x p matrix X^T X where it lies in HBM (engine.xtx_operator: O(p^2), 3.2 GB at
    # p = 20 000) plus O(p K) on the host -- the n >= p analogue of the rank updates of the p > n branch.  Nothing p x p
    # visits the host.
    eng.slot("C", q)
    eng.xt_v(Y, 1.0, "C", 0)
    COV = eng.fetch("C")                                    # X^T Y (:587)
    var0 = eng.xtx_operator()                               # w -> (X^T X) w (:586)
    p = eng.p
    Wc = np.empty((p, 0)); P = np.empty((p, 0)); V = np.empty((q, 0)); U = np.empty((n, 0))
    Gd = np.empty((p, 0)); dens = np.empty(0)
    _plan(eng, solver="kernel n >= p: rank-1 deflation of X^T X, one p x p product per component")
    # (column shards: every p-length inner product below is a sum over the ranks' rows)
    ar = eng.comm.allreduce_host
    for k in range(K):
        w = _tlsv(COV, ar if eng.comm.is_distributed else None)    # S = COV COV^T stays true under :631,:633
        g = var0(w)                                         # VAR_0 w
        if k:
            g = g - Gd.dot(ar(Gd.T.dot(w)) / dens[:, None])   # ... - sum_i g_i (g_i^T w) / den_i  =  VAR_k w
        den = ar(w.T.dot(g)).item()                         # (:595)
        wC = ar(w.T.dot(COV))
        v = wC / den                                        # (:596)
        pl = g.T / den                                      # (:599)  w^T VAR / den, VAR symmetric
        if calc_all:
            u = Y.dot(v.T) / v.dot(v.T)                     # (:624-625)
            U = np.hstack((U, u / np.linalg.norm(u)))
        COV = COV - pl.T.dot(wC)                            # (:633)
        Gd = np.hstack((Gd, g)); dens = np.append(dens, den)
        V = np.hstack((V, v.T)); P = np.hstack((P, pl.T)); Wc = np.hstack((Wc, w))
    R = Wc.dot(np.linalg.pinv(ar(P.T.dot(Wc))))             # (:642)
    beta = R.dot(V.T)                                       # (:643)
    Ts = eng.x_w_host(R)                                    # (:644)
    nrm = np.linalg.norm(Ts, axis=0)
    V = V * nrm; P = P * nrm; Ts = Ts / nrm                 # (:646-650)
    res.update(Ts_=Ts, U_=U, V_=V, W_concat_=Wc)
    res.update(empty_blocks)
    if calc_all:
        offs = eng.block_offsets
        eng.put("Wblk", Wc)
        wnorm2 = ar(np.array([(Wc[offs[b]:offs[b + 1]] ** 2).sum(axis=0) for b in range(B)]))
        _kernel_extras(eng, res, Y, P, V, Ts, varxb, "Wblk", wnorm2)
    res.R_ = R
    res.beta_ = beta
    res.P_ = _split(P, eng.block_offsets)
    return res


# ----------------------------------------------------------------------------------- SIMPLS
def fit_simpls(eng, Y, K, standardize):
    """mbpls.py:995-1048 (de Jong 1993) with ONE pass over X per component.  X is never deflated: S = X^T Y, and per
    component the reference computes t = X r (:1006, r = S q) and p = X^T t (:1011) -- two reads of X.  Here the fused
    sweep pass (the NIPALS one: g = X^T v and X g in the same read) takes v = t: it returns the loadings p = X^T t AND
    a = X p.  Everything r is built from is a combination of vectors whose products with X are then known:
        S_0 = X^T Y, X S_0                (the first response with the statistics pass, the others by two products)
        v = p - V_s (V_s^T p)  (:1016)  ->  X v = a - (X V_s)(V_s^T p)
        S <- S - v (v^T S)     (:1019)  ->  X S <- X S - (X v)(v^T S)
    so t = X r = (X S) q needs no pass.  All p-length algebra (r = S q, the Gram-Schmidt step on v, the update of S)
    runs on the device in one p x (K+1+q) slot  Z = [V_s | p | S]; only n- and q-sized matrices visit the host.
    With standardised data nothing is written: the passes standardise in registers (engine: lazy / implicit mode with
    no component ever deflated)."""
    n = eng.n
    q = Y.shape[1]
    if K + 1 + q > 32:
        return _fit_simpls_hostvec(eng, Y, K, standardize)
    if not eng.sweep_reads_once():
        _plan(eng, solver="simpls: two products per component")
        return _fit_simpls_products(eng, Y, K, standardize)     # tall data: a fused sweep is two reads there as well
    _plan(eng, solver="simpls: one fused sweep per component (maintained products)")
    nz = K + 1 + q
    cS = K + 1                                              # first S column inside Z
    eng.slot("Z", nz); eng.slot("R", K); eng.slot("W", K); eng.slot("P", K)
    eng.use_lazy_deflation(K, "P", prefer="lazy")           # (never deflates: the never-writes-X passes when they apply)
    XS = np.empty((n, q)); XV = np.empty((n, K), order="F")
    _, tb, _ = eng.prepare(standardize, v=Y[:, 0:1], alpha=1.0, g_slot=("Z", cS))       # S[:, 0] = X^T y_0 (:998), X S[:, 0]
    XS[:, 0] = tb.sum(axis=1)
    if q == 2:
        tb, _ = eng.sweep(Y[:, 1:2], 1.0, ("Z", cS + 1))
        XS[:, 1] = tb.sum(axis=1)
    elif q > 2:                                             # the other responses: one X^T Y and one X W product
        eng.scaled_products()
        eng.xt_v(Y[:, 1:], 1.0, "Z", cS + 1)
        XS[:, 1:] = eng.x_w_slot("Z", cS + 1, q - 1)
    rm = getattr(eng, "rowmask", None)
    inc = None if rm is None else (np.asarray(rm).ravel() != 0)
    Ts = np.empty((n, K), order="F"); U = np.empty((n, K)); Q = np.empty((q, K))   # (F: contiguous leading columns)
    out = _ColumnsOut(eng, K, q)
    try:
        _simpls_fused_components(eng, Y, K, q, nz, cS, XS, XV, inc, Ts, U, Q, out)
        out.send_beta(Q)                                    # (:1044)
    finally:
        out.wait()              # (on the error path too: the columns in flight land before the slots are released)
    return FitResult(P_=_split(out.P_h, eng.block_offsets), Ts_=Ts, U_=U, R_=out.R_h, V_=Q, W_=out.W_h, beta_=out.beta_h)


def _simpls_fused_components(eng, Y, K, q, nz, cS, XS, XV, inc, Ts, U, Q, out):
    for comp in range(K):
        if q == 1:
            qv = np.array([[1.0]])                          # the 1 x 1 S^T S has the singular vector [1]
        else:
            StS = eng.cross(("Z", cS, q), ("Z", cS, q))
            qv = np.linalg.svd(StS, full_matrices=True)[0][:, 0:1]  # (:1001)
        eng.right_mul(("Z", cS, q), qv, ("W", comp))        # w = r = S q (:1004-1005)
        t = XS.dot(qv)                                      # X r (:1006), no pass
        t = t - np.mean(t) if inc is None else np.where(inc[:, None], t - t[inc].mean(), 0.0)   # (:1007)
        nt = np.sqrt(t.T.dot(t)).item()
        t = t / nt                                          # (:1008-1009)
        eng.right_mul(("W", comp, 1), np.array([[1.0 / nt]]), ("R", comp))   # r / ||t|| (:1010)
        # p = X^T t (:1011) into the scratch column of Z, and a = X p
        tb, _ = eng.sweep(t, 1.0, ("Z", K))
        a = tb.sum(axis=1, keepdims=True)
        eng.right_mul(("Z", K, 1), np.array([[1.0]]), ("P", comp))
        out.send(comp)                                      # w, r, p of this component are final
        qk = Y.T.dot(t)                                     # (:1012)
        u = Y.dot(qk)                                       # (:1013)
        coef = np.zeros((nz, 1))
        coef[K, 0] = 1.0
        xv = a
        if comp > 0:
            c = eng.cross(("Z", 0, comp), ("Z", K, 1))      # V_s^T p
            coef[:comp] = -c                                # v = p - V_s (V_s^T p) (:1016)
            xv = a - XV[:, :comp].dot(c)
            u = u - Ts[:, :comp].dot(Ts[:, :comp].T.dot(u))            # (:1017)
        eng.right_mul(("Z", 0, nz), coef, ("Z", comp))
        # one reduction for v^T v and v^T S (columns comp .. end of Z; the columns in between are not used)
        row = eng.cross(("Z", comp, 1), ("Z", comp, nz - comp))
        vv = row[0, 0]
        eng.right_mul(("Z", comp, 1), np.array([[1.0 / np.sqrt(vv)]]), ("Z", comp))     # (:1018)
        xv = xv / np.sqrt(vv)
        vS = row[:, cS - comp:] / np.sqrt(vv)               # v^T S of the unit-length v (1 x q)
        M = np.zeros((nz, q))
        M[cS:, :] = np.eye(q)
        M[comp, :] = -vS
        eng.right_mul(("Z", 0, nz), M, ("Z", cS))           # S <- S - v (v^T S) (:1019), in place
        XS -= xv.dot(vS)
        XV[:, comp:comp + 1] = xv
        Ts[:, comp:comp + 1] = t; U[:, comp:comp + 1] = u / np.linalg.norm(u); Q[:, comp:comp + 1] = qk


def _fit_simpls_products(eng, Y, K, standardize):
    """The two-reads-per-component form (tall data, n > 8192: the column passes are two streaming phases there, a fused
    sweep would read X twice as well, and the matrix-core products stream faster).
    mbpls.py:995-1048 (de Jong 1993).  X is never deflated: S = X^T Y once, then per component
    t = X r and p = X^T t (two reads of X).  All p-length algebra (r = S q, the Gram-Schmidt step on v,
    S <- S - v v^T S) runs on the device in one p x (K+1+q) slot  Z = [V_s | p | S]; only q x q and
    k x 1 matrices visit the host."""
    n = eng.n
    q = Y.shape[1]
    _prepare_products(eng, standardize)
    nz = K + 1 + q
    cS = K + 1                                              # first S column inside Z
    eng.slot("Z", nz); eng.slot("R", K); eng.slot("W", K); eng.slot("P", K)
    eng.xt_v(Y, 1.0, "Z", cS)                               # S = X^T Y (:998)
    rm = getattr(eng, "rowmask", None)
    inc = None if rm is None else (np.asarray(rm).ravel() != 0)
    Ts = np.empty((n, K), order="F"); U = np.empty((n, K)); Q = np.empty((q, K))   # (F: contiguous leading columns)
    out = _ColumnsOut(eng, K, q)
    try:
        _simpls_product_components(eng, Y, K, q, nz, cS, inc, Ts, U, Q, out)
        out.send_beta(Q)                                    # (:1044)
    finally:
        out.wait()
    return FitResult(P_=_split(out.P_h, eng.block_offsets), Ts_=Ts, U_=U, R_=out.R_h, V_=Q, W_=out.W_h, beta_=out.beta_h)


def _simpls_product_components(eng, Y, K, q, nz, cS, inc, Ts, U, Q, out):
    for comp in range(K):
        if q == 1:
            qv = np.array([[1.0]])                          # the 1 x 1 S^T S has the singular vector [1]
        else:
            StS = eng.cross(("Z", cS, q), ("Z", cS, q))
            qv = np.linalg.svd(StS, full_matrices=True)[0][:, 0:1]  # (:1001)
        eng.right_mul(("Z", cS, q), qv, ("W", comp))        # w = r = S q (:1004-1005)
        t = eng.x_w_slot("W", comp, 1)                      # (:1006)
        t = t - np.mean(t) if inc is None else np.where(inc[:, None], t - t[inc].mean(), 0.0)   # (:1007)
        nt = np.sqrt(t.T.dot(t)).item()
        t = t / nt                                          # (:1008-1009)
        eng.right_mul(("W", comp, 1), np.array([[1.0 / nt]]), ("R", comp))   # r / ||t|| (:1010)
        eng.xt_v(t, 1.0, "Z", K)                            # p = X^T t (:1011) into the scratch column
        eng.right_mul(("Z", K, 1), np.array([[1.0]]), ("P", comp))
        out.send(comp)                                      # w, r, p of this component are final
        qk = Y.T.dot(t)                                     # (:1012)
        u = Y.dot(qk)                                       # (:1013)
        coef = np.zeros((nz, 1))
        coef[K, 0] = 1.0
        if comp > 0:
            coef[:comp] = -eng.cross(("Z", 0, comp), ("Z", K, 1))     # v = p - V_s (V_s^T p) (:1016)
            u = u - Ts[:, :comp].dot(Ts[:, :comp].T.dot(u))            # (:1017)
        eng.right_mul(("Z", 0, nz), coef, ("Z", comp))
        # one reduction for v^T v and v^T S (columns comp .. end of Z; the columns in between are not used)
        row = eng.cross(("Z", comp, 1), ("Z", comp, nz - comp))
        vv = row[0, 0]
        eng.right_mul(("Z", comp, 1), np.array([[1.0 / np.sqrt(vv)]]), ("Z", comp))     # (:1018)
        vS = row[:, cS - comp:] / np.sqrt(vv)               # v^T S of the unit-length v (1 x q)
        M = np.zeros((nz, q))
        M[cS:, :] = np.eye(q)
        M[comp, :] = -vS
        eng.right_mul(("Z", 0, nz), M, ("Z", cS))           # S <- S - v (v^T S) (:1019), in place
        Ts[:, comp:comp + 1] = t; U[:, comp:comp + 1] = u / np.linalg.norm(u); Q[:, comp:comp + 1] = qk


def _fit_simpls_hostvec(eng, Y, K, standardize):
    """SIMPLS with the p-length vectors on the host (used when K + 1 + q exceeds the 32-column limit of the
    device helpers)."""
    n = eng.n
    q = Y.shape[1]
    _plan(eng, solver="simpls: two products per component, p-length vectors on the host (K + 1 + q > 32)")
    _prepare_products(eng, standardize)
    eng.slot("S", q); eng.slot("r", 1); eng.slot("p", 1)
    eng.xt_v(Y, 1.0, "S", 0)
    S = eng.fetch("S")                                      # (:998)
    cols = {k: [] for k in ("R", "T", "P", "Q", "U", "Vs", "W")}
    for comp in range(K):
        StS = eng.comm.allreduce_host(S.T.dot(S))
        qv = np.array([[1.0]]) if q == 1 else np.linalg.svd(StS, full_matrices=True)[0][:, 0:1]  # (:1001)
        r = S.dot(qv)                                       # (:1004)
        w = r.copy()
        eng.put("r", r)
        t = eng.x_w_slot("r", 0, 1)                         # (:1006)
        rm = getattr(eng, "rowmask", None)
        if rm is None:
            t = t - np.mean(t)
        else:                                               # row-subset fit: centre over the included samples
            inc = np.asarray(rm).ravel() != 0
            t = np.where(inc[:, None], t - t[inc].mean(), 0.0)
        nt = np.sqrt(t.T.dot(t))
        t = t / nt; r = r / nt                              # (:1008-1010)
        eng.xt_v(t, 1.0, "p", 0)
        pl = eng.fetch("p")                                 # (:1011)
        qk = Y.T.dot(t)                                     # (:1012)
        u = Y.dot(qk)
        v = pl
        if comp > 0:
            Vs = np.hstack(cols["Vs"]); Tm = np.hstack(cols["T"])
            v = v - Vs.dot(eng.comm.allreduce_host(Vs.T.dot(pl)))   # (:1016)
            u = u - Tm.dot(Tm.T.dot(u))                     # (:1017)
        v = v / np.sqrt(eng.comm.allreduce_host(v.T.dot(v)))        # (:1018)
        S = S - v.dot(eng.comm.allreduce_host(v.T.dot(S)))          # (:1019)
        for key, val in zip(("R", "T", "P", "Q", "U", "Vs", "W"), (r, t, pl, qk, u / np.linalg.norm(u), v, w)):
            cols[key].append(val)
    P = np.hstack(cols["P"])
    res = FitResult(P_=_split(P, eng.block_offsets), Ts_=np.hstack(cols["T"]), U_=np.hstack(cols["U"]),
                    R_=np.hstack(cols["R"]), V_=np.hstack(cols["Q"]), W_=np.hstack(cols["W"]))
    res.beta_ = res.R_.dot(res.V_.T)                        # (:1044)
    return res
