"""BASELINE-size configurations asserted on the GPU.

The CPU oracle cannot run these sizes (C2 alone is 32.8 GB of X and minutes of numpy), so the fitted
model is checked through size-independent properties that go through code paths DIFFERENT from the fused
column passes that produced it:

(a) the generic multi-right-hand-side products (``x_w`` / ``xw_mfma`` / ``axpy_cols`` behind ``transform`` and
    ``predict``):  Z R_ == Ts_,  Z beta_ == Ts_ V_^T,  X_b W_b - Ts triu(P_b^T W_b, 1) == T_[b];
(b) ~2000 sampled columns (first / last column, chunk and block boundaries, the far end of the 4.1e9-element
    address range) downloaded and pushed through the reference recurrence on the CPU with the GPU's Ts_ / U_:
    mean_/scale_ (mbpls.py:314), w~_j = z_j^T u / u^T u (:856), p_j = z_j^T ts / ts^T ts (:928),
    z_j <- z_j - ts p_j (:969);
(c) Ts^T Ts = I, NIPALS iteration counts [2]*K, explained variances consistent with the loadings;
(d) lazy vs explicit deflation (MBPLS_DEFLATION) and implicit vs stored standardisation agree at full size.

Sizes: C2 (n=4096, p=1e6, K=10 NIPALS), C3 (3 blocks 300k/300k/400k UNIPALS), C4 (n=8192, p=2e6, K=20 KERNEL,
131 GB, fitted in place as the 288 GB of one MI355X cannot hold a second copy).  Tolerance 1e-6 relative
(north_star), asserted at 1e-8.
"""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

TOL = 1e-8


def planted(n, q, r, seed, decay=0.85, noise_y=0.5):
    rng = np.random.RandomState(seed)
    T = rng.standard_normal((n, r))
    C = rng.standard_normal((r, q))
    Y = T.dot(C) + noise_y * rng.standard_normal((n, q))
    return T * decay ** np.arange(r), Y


def device_problem(ctx, n, p, offs, q=1, seed=0):
    """SURVEY 8d synthetic matrix generated in HBM (planted latent structure), as bench.py does."""
    from mbpls_amd import DeviceMatrix
    Ts, Y = planted(n, q, 32, seed)
    X = DeviceMatrix(ctx, n, p, offs)
    dT = ctx.from_host(Ts)
    X.fill_latent(dT, 32, 0.5, 1234, 0)
    ctx.sync()
    dT.free()
    return X, Y


def sample_ranges(X, width=48, extra=()):
    """column ranges to download: both ends, every block boundary, a few chunk boundaries, a spread of interior
    points and the last columns (byte offsets beyond 2^32 elements)"""
    p = X.p
    starts = {0, p - width}
    for o in X.block_offsets[1:-1]:
        starts.update((o - width // 2, o - width, o))
    per = max(1, p // max(1, X.nchunks))
    for c in (1, 2, X.nchunks // 2, X.nchunks - 1):
        starts.add(min(max(0, c * per - width // 2), p - width))
    rng = np.random.RandomState(5)
    starts.update(int(s) for s in rng.randint(0, p - width, size=26))
    starts.update(extra)
    return sorted(s for s in starts if 0 <= s <= p - width), width


def download_sample(X, ranges, width):
    cols = np.concatenate([np.arange(s, s + width) for s in ranges])
    data = np.hstack([X.download(s, width, order="F") for s in ranges])
    return cols, data


def scaler_vectors(mdl):
    return (np.concatenate([s.mean_ for s in mdl.x_scalers_]), np.concatenate([s.scale_ for s in mdl.x_scalers_]))


def check_statistics(mdl, cols, data):
    mean, scale = scaler_vectors(mdl)
    assert util.relerr(mean[cols], data.mean(axis=0)) < 1e-12
    assert util.relerr(scale[cols], data.std(axis=0)) < 1e-12
    return (data - data.mean(axis=0)) / data.std(axis=0)


def recurrence(Z, Ts, U, weights_from_u=True):
    """reference recurrence on sampled standardised columns: returns (W~ sample, P sample), each len(cols) x K"""
    K = Ts.shape[1]
    Z = Z.copy()
    Wn = np.empty((Z.shape[1], K)); P = np.empty((Z.shape[1], K))
    for k in range(K):
        u, ts = U[:, k], Ts[:, k]
        Wn[:, k] = Z.T.dot(u) / u.dot(u)                  # mbpls.py:856 / :503
        P[:, k] = Z.T.dot(ts) / ts.dot(ts)                # :928 / :520
        Z -= np.outer(ts, P[:, k])                        # :969 / :535
    return Wn, P


def check_generic_products(mdl, X, Y, blocks=True, check_u=True):
    """(a): the fitted model through transform / predict (generic x_w kernels, on-the-fly standardisation)"""
    out = mdl.transform(X, Y, return_block_scores=True) if blocks else mdl.transform(X, Y)
    Ts_new, U_new = out[0], out[-1]
    s = np.sign(np.sum(Ts_new * mdl.Ts_, axis=0))
    assert np.all(s == 1.0)                                # same model, same data: no sign freedom
    assert util.relerr(Ts_new, mdl.Ts_) < TOL, util.relerr(Ts_new, mdl.Ts_)
    if check_u:     # (KERNEL takes U_ from the deflated Y, transform from Y itself: not comparable there)
        assert util.relerr(U_new, mdl.U_) < TOL
    if blocks:
        for b, Tb in enumerate(out[1]):
            assert util.relerr(Tb, mdl.T_[b]) < TOL, (b, util.relerr(Tb, mdl.T_[b]))
    yhat = mdl.predict(X)
    fitted = mdl.y_scaler_.inverse_transform(mdl.Ts_.dot(mdl.V_.T))    # Z beta_ = Z R_ V_^T = Ts_ V_^T
    assert util.relerr(yhat, fitted) < TOL
    return yhat


def check_orthonormal(Ts):
    K = Ts.shape[1]
    assert np.max(np.abs(Ts.T.dot(Ts) - np.eye(K))) < 1e-10


# ---------------------------------------------------------------------------------------------- C2
def test_c2_nipals_full_size(gpu_required, monkeypatch):
    """BASELINE config[1]: single-block NIPALS, n=4096, p=1,000,000, K=10, q=1 (4.1e9 elements)."""
    import mbpls_amd
    from mbpls_amd import MBPLS
    ctx = mbpls_amd.get_context()
    ctx.trim_pool()
    n, p, K = 4096, 1_000_000, 10
    X, Y = device_problem(ctx, n, p, [0, p])
    try:
        mdl = MBPLS(n_components=K, method="NIPALS").fit(X, Y)
        assert mdl.nipals_iterations_ == [1] * K                          # (c) one response: one pass per component
        check_orthonormal(mdl.Ts_)
        yhat = check_generic_products(mdl, X, Y)                          # (a)
        assert 0.5 < 1.0 - np.sum((Y - yhat) ** 2) / np.sum((Y - Y.mean()) ** 2) < 1.0    # a sensible model
        ranges, width = sample_ranges(X)
        cols, data = download_sample(X, ranges, width)                    # (b)
        assert cols.size >= 1500 and cols[-1] == p - 1
        Z = check_statistics(mdl, cols, data)
        Wn, P = recurrence(Z, mdl.Ts_, mdl.U_)
        assert util.relerr(mdl.W_non_normal_[0][cols], Wn) < TOL
        assert util.relerr(mdl.P_[0][cols], P) < TOL
        nu = np.linalg.norm(mdl.W_non_normal_[0], axis=0)
        assert util.relerr(mdl.W_[0][cols], Wn / nu) < TOL
        # explained variance from the fitted loadings (analytic form of mbpls.py:933): ||ts p^T||^2 / ||Z||^2
        evx = (mdl.Ts_ ** 2).sum(axis=0) * (mdl.P_[0] ** 2).sum(axis=0) / (float(n) * p)
        assert util.relerr(np.asarray(mdl.explained_var_x_), evx) < TOL
        # R_ = W (P^T W)^+ (:988) recomputed on the host from the fetched attributes
        Wc = mdl.W_non_normal_[0] / nu
        R = Wc.dot(np.linalg.pinv(mdl.P_[0].T.dot(Wc)))
        assert util.relerr(mdl.R_, R) < TOL
        assert util.relerr(mdl.beta_, R.dot(mdl.V_.T)) < TOL
        # (d) explicit in-place deflation of a stored standardised copy == the never-writes-X fit
        monkeypatch.setenv("MBPLS_DEFLATION", "explicit")
        m2 = MBPLS(n_components=K, method="NIPALS").fit(X, Y)
        monkeypatch.delenv("MBPLS_DEFLATION")
        assert m2.nipals_iterations_ == [1] * K
        w = util.compare(util.model_to_flat(m2), util.model_to_flat(mdl), TOL, what="explicit vs lazy")
        assert w < TOL
        del m2
        # (e) the reference's iteration structure -- two passes per component, the second re-deriving the first until
        # ||ts_old - ts|| <= 1e-14 (mbpls.py:841) -- gives the same model, and meets its stop rule on every component
        monkeypatch.setenv("MBPLS_NIPALS_Q1_FIXED_POINT", "0")
        m3 = MBPLS(n_components=K, method="NIPALS").fit(X, Y)
        monkeypatch.delenv("MBPLS_NIPALS_Q1_FIXED_POINT")
        assert m3.nipals_iterations_ == [2] * K and all(m3.nipals_converged_)
        assert "fixed point" in mdl.fit_plan_["solver"] and "fixed point" not in m3.fit_plan_["solver"]
        w = util.compare(util.model_to_flat(m3), util.model_to_flat(mdl), TOL, what="two passes vs one per component")
        assert w < 1e-9, w
        del m3
        cols2, data2 = download_sample(X, ranges[:3], width)
        assert np.array_equal(data2, data[:, :cols2.size])                # copy=True: the caller's matrix is intact
    finally:
        X.close()
        ctx.trim_pool()


# ---------------------------------------------------------------------------------------------- C3
def test_c3_unipals_multiblock_full_size(gpu_required, monkeypatch):
    """BASELINE config[2]: 3 blocks p=[300k, 300k, 400k], n=4096, K=10, UNIPALS."""
    import mbpls_amd
    from mbpls_amd import MBPLS
    ctx = mbpls_amd.get_context()
    ctx.trim_pool()
    n, K = 4096, 10
    offs = [0, 300_000, 600_000, 1_000_000]
    X, Y = device_problem(ctx, n, offs[-1], offs)
    try:
        mdl = MBPLS(n_components=K, method="UNIPALS").fit(X, Y)
        check_orthonormal(mdl.Ts_)
        check_generic_products(mdl, X, Y)
        ranges, width = sample_ranges(X)
        cols, data = download_sample(X, ranges, width)
        Z = check_statistics(mdl, cols, data)
        Wraw, P = recurrence(Z, mdl.Ts_, mdl.U_)
        Pfit = np.vstack(mdl.P_)
        assert util.relerr(Pfit[cols], P) < TOL
        # block weights are w_b / ||w_b|| (:510-512): proportional to X_b^T u on every block, unit length, and
        # a_b = ||w_b||^2 of the unit-length concatenated w (:512)
        Wfit = np.vstack(mdl.W_)
        for b in range(3):
            sel = (cols >= offs[b]) & (cols < offs[b + 1])
            assert sel.sum() >= 100
            for k in range(K):
                c = Wfit[cols[sel], k].dot(Wraw[sel, k]) / Wraw[sel, k].dot(Wraw[sel, k])
                assert np.max(np.abs(Wfit[cols[sel], k] - c * Wraw[sel, k])) < TOL * np.max(np.abs(Wfit[cols[sel], k]))
            assert np.max(np.abs(np.linalg.norm(mdl.W_[b], axis=0) - 1.0)) < 1e-12
        assert np.max(np.abs(mdl.A_.sum(axis=0) - 1.0)) < 1e-12
        # the same fit with X (X^T y) taken by the explicit passes on a stored copy
        monkeypatch.setenv("MBPLS_DEFLATION", "explicit")
        m2 = MBPLS(n_components=K, method="UNIPALS").fit(X, Y)
        monkeypatch.delenv("MBPLS_DEFLATION")
        assert util.compare(util.model_to_flat(m2), util.model_to_flat(mdl), TOL, what="explicit vs lazy") < TOL
        # multiblock NIPALS on the same matrix converges to the same subspace for one response
        m3 = MBPLS(n_components=K, method="NIPALS").fit(X, Y)
        assert m3.nipals_iterations_ == [1] * K
        assert util.relerr(np.abs(m3.Ts_), np.abs(mdl.Ts_)) < 1e-6
        assert util.relerr(m3.beta_, mdl.beta_) < 1e-6
    finally:
        X.close()
        ctx.trim_pool()


# ---------------------------------------------------------------------------------------------- C4
def test_c4_kernel_full_size(gpu_required, monkeypatch):
    """BASELINE config[3]: KERNEL method, n=8192, p=2,000,000, K=20 (131 GB; 1.34e14 flop Gram on the f64 matrix
    cores).  First with the default copy=True: statistics-only preparation, every product standardises its operand in
    registers, the caller's matrix is untouched and no second 131 GB buffer exists.  Then with the stored path and
    copy=False (MBPLS_IMPLICIT_STD=0): X is standardised in place (the KERNEL branch never deflates X), so afterwards
    the resident matrix IS Z and the generic products can be applied to it directly."""
    import mbpls_amd
    from mbpls_amd import MBPLS
    ctx = mbpls_amd.get_context()
    ctx.trim_pool()
    n, p, K = 8192, 2_000_000, 20
    info = ctx.info()
    if info["total_bytes"] < 200e9:
        pytest.skip("needs the 288 GB of an MI355X")
    X, Y = device_problem(ctx, n, p, [0, p])
    try:
        ranges, width = sample_ranges(X)
        cols, raw = download_sample(X, ranges, width)                     # before the in-place standardisation
        from mbpls_amd.engine import HipEngine
        used, real = [], HipEngine.prepare_stats
        monkeypatch.setattr(HipEngine, "prepare_stats", lambda self, *a, **k: (used.append(True), real(self, *a, **k))[1])
        monkeypatch.setenv("MBPLS_KERNEL_IMPLICIT", "1")  # (by itself only when a second 131 GB matrix does not fit)
        m0 = MBPLS(n_components=K, method="KERNEL").fit(X, Y)             # copy=True, nothing written, no working copy
        monkeypatch.delenv("MBPLS_KERNEL_IMPLICIT")
        assert used == [True]
        _, again = download_sample(X, ranges, width)
        assert np.array_equal(again, raw)
        check_orthonormal(m0.Ts_)
        check_generic_products(m0, X, Y, check_u=False)                   # transform / predict standardise on the fly
        monkeypatch.setenv("MBPLS_IMPLICIT_STD", "0")
        mdl = MBPLS(n_components=K, method="KERNEL", copy=False).fit(X, Y)
        monkeypatch.delenv("MBPLS_IMPLICIT_STD")
        assert util.compare(util.model_to_flat(m0), util.model_to_flat(mdl), TOL, what="implicit vs stored") < TOL
        check_orthonormal(mdl.Ts_)
        Z = check_statistics(mdl, cols, raw)
        _, zdev = download_sample(X, ranges, width)
        assert util.relerr(zdev, Z) < 1e-12                               # X now holds Z, also beyond 2^32 elements
        # (a) generic products on the resident Z:  Z R_ = Ts_,  Z W_ - Ts triu(P^T W, 1) = T_
        dR = ctx.from_host(mdl.R_)
        out = ctx.empty(n * K)
        X.x_w(dR, K, out)
        assert util.relerr(out.get().reshape(K, n).T, mdl.Ts_) < TOL
        dW = ctx.from_host(mdl.W_[0])
        X.x_w(dW, K, out)
        Tb = out.get().reshape(K, n).T - mdl.Ts_.dot(np.triu(mdl.P_[0].T.dot(mdl.W_[0]), 1))
        assert util.relerr(Tb, mdl.T_[0]) < TOL
        # (b) sampled columns: W_concat_ = colnorm(X^T U_) (:731-733), P_ = X^T Ts (Ts^T Ts)^+ (:734)
        Wraw = Z.T.dot(mdl.U_)
        cn = np.linalg.norm(mdl.W_concat_, axis=0)
        assert np.max(np.abs(cn - 1.0)) < 1e-12
        for k in range(K):
            c = mdl.W_concat_[cols, k].dot(Wraw[:, k]) / Wraw[:, k].dot(Wraw[:, k])
            assert np.max(np.abs(mdl.W_concat_[cols, k] - c * Wraw[:, k])) < TOL * np.max(np.abs(mdl.W_concat_[cols, k]))
        P = Z.T.dot(mdl.Ts_).dot(np.linalg.pinv(mdl.Ts_.T.dot(mdl.Ts_)))
        assert util.relerr(mdl.P_[0][cols], P) < TOL
        # the association matrix itself at full size: three columns of Z Z^T through xt_v (c = 1) and x_w (c = 1),
        # against the matrix-core Gram kernel; and the first score vector ts_1 ~ A y (:712-715 with one response)
        ldg = n + (n & 1)
        G = ctx.empty((ldg, n), zero=True)
        X.gram(G, ldg)
        Gh = G.get()[:n, :n]
        G.free()
        assert np.array_equal(Gh, Gh.T)
        row = ctx.empty(p)
        vec = ctx.empty(n)
        for i in (0, 4097, n - 1):
            e = np.zeros(n); e[i] = 1.0
            vec.set(e)
            X.xt_v(vec, 1, 1.0, row)                                     # row i of Z as a p-vector
            X.x_w(row, 1, out)                                           # Z z_i = column i of the Gram
            gi = out.get()[:n]
            assert util.relerr(Gh[:, i], gi) < 1e-10, i
        ys = mdl.y_scaler_.transform(Y)
        a = Gh.dot(ys)
        ts1 = (a / np.linalg.norm(a)).ravel()
        assert util.relerr(ts1 * np.sign(ts1.dot(mdl.Ts_[:, 0])), mdl.Ts_[:, 0]) < TOL
        for d in (dR, dW, out, row, vec):
            d.free()
    finally:
        X.close()
        ctx.trim_pool()


# ------------------------------------------------------------------------------- several responses at C2 size
def test_c2_several_responses_full_size(gpu_required, monkeypatch):
    """The secondary q = 10 configuration (paper protocol at BASELINE size; n=4096, p=1,000,000, 2 blocks,
    K=10) for the solvers that keep X^T (P Y) and X_b S_b up to date instead of passing over X (one fused sweep per
    component): NIPALS and UNIPALS with ten responses, SIMPLS.  Checked through (a) the generic products behind
    transform / predict, (b) sampled columns through the reference recurrence, (c) orthonormal scores, and (d) the
    same fit through a DIFFERENT solver route (NIPALS on per-block association matrices; UNIPALS on the explicitly
    deflated matrix)."""
    import mbpls_amd
    from mbpls_amd import MBPLS
    ctx = mbpls_amd.get_context()
    ctx.trim_pool()
    n, p, K, q = 4096, 1_000_000, 10, 10
    offs = [0, 500_000, p]
    X, Y = device_problem(ctx, n, p, offs, q=q)
    try:
        ranges, width = sample_ranges(X)
        cols, data = download_sample(X, ranges, width)
        # ---- NIPALS
        mdl = MBPLS(n_components=K, method="NIPALS").fit(X, Y)
        assert "never written" in mdl.fit_plan_["standardisation"]
        check_orthonormal(mdl.Ts_)
        check_generic_products(mdl, X, Y)
        Z = check_statistics(mdl, cols, data)
        _, P = recurrence(Z, mdl.Ts_, mdl.U_)
        assert util.relerr(np.vstack(mdl.P_)[cols], P) < TOL
        Wn = np.vstack(mdl.W_non_normal_)
        for b in range(2):
            nu = np.linalg.norm(mdl.W_non_normal_[b], axis=0)
            assert util.relerr(mdl.W_[b], mdl.W_non_normal_[b] / nu) < 1e-12
        Wc = Wn / np.linalg.norm(Wn, axis=0)
        R = Wc.dot(np.linalg.pinv(np.vstack(mdl.P_).T.dot(Wc)))                     # (:988)
        assert util.relerr(mdl.R_, R) < TOL and util.relerr(mdl.beta_, R.dot(mdl.V_.T)) < TOL
        monkeypatch.setenv("MBPLS_NIPALS_SUBSPACE", "0")                            # per-block association matrices
        m2 = MBPLS(n_components=K, method="NIPALS").fit(X, Y)
        monkeypatch.delenv("MBPLS_NIPALS_SUBSPACE")
        assert m2.fit_plan_["solver"] != mdl.fit_plan_["solver"], m2.fit_plan_
        assert all(abs(a - b) <= 2 for a, b in zip(m2.nipals_iterations_, mdl.nipals_iterations_))
        assert util.compare(util.model_to_flat(m2), util.model_to_flat(mdl), 1e-7, what="Gram route vs maintained") < 1e-7
        del m2
        # ---- UNIPALS
        mu = MBPLS(n_components=K, method="UNIPALS").fit(X, Y)
        check_orthonormal(mu.Ts_)
        # (no Z R_ == Ts_ check here: with several responses the reference takes ts as the first LEFT SINGULAR vector of
        # the non-symmetric X X^T Y Y^T (:489-493), which is not X w / ||X w|| for its weights w = X^T u (:503))
        Wraw, P = recurrence(Z, mu.Ts_, mu.U_)
        assert util.relerr(np.vstack(mu.P_)[cols], P) < TOL
        Wfit = np.vstack(mu.W_)
        for b in range(2):
            sel = (cols >= offs[b]) & (cols < offs[b + 1])
            for k in range(K):
                c = Wfit[cols[sel], k].dot(Wraw[sel, k]) / Wraw[sel, k].dot(Wraw[sel, k])
                assert np.max(np.abs(Wfit[cols[sel], k] - c * Wraw[sel, k])) < TOL * np.max(np.abs(Wfit[cols[sel], k]))
        monkeypatch.setenv("MBPLS_UNIPALS_NODEFLATE", "0")                          # explicitly deflated working copy
        m2 = MBPLS(n_components=K, method="UNIPALS").fit(X, Y)
        monkeypatch.delenv("MBPLS_UNIPALS_NODEFLATE")
        assert util.compare(util.model_to_flat(m2), util.model_to_flat(mu), 1e-7, what="deflating vs maintained") < 1e-7
        del m2, mu
        # ---- SIMPLS (never deflates: P = Z^T Ts, Ts = Z R_, orthonormal scores, beta_ = R_ V_^T)
        ms = MBPLS(n_components=K, method="SIMPLS").fit(X, Y)
        check_orthonormal(ms.Ts_)
        check_generic_products(ms, X, Y, blocks=False, check_u=False)
        Pfit = np.vstack(ms.P_) if isinstance(ms.P_, list) else ms.P_
        assert util.relerr(Pfit[cols], Z.T.dot(ms.Ts_)) < TOL
        Ys = (Y - Y.mean(axis=0)) / Y.std(axis=0)
        assert util.relerr(ms.V_, Ys.T.dot(ms.Ts_)) < TOL
        assert util.relerr(ms.beta_, ms.R_.dot(ms.V_.T)) < 1e-12
        cols2, data2 = download_sample(X, ranges[:3], width)
        assert np.array_equal(data2, data[:, :cols2.size])                # the caller's matrix is intact
    finally:
        X.close()
        ctx.trim_pool()
