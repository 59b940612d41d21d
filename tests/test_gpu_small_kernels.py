"""Kernel-level checks of the round-3 helpers against numpy, through the C ABI: the K x K end-of-fit algebra on the
device (one-wave Jacobi pseudo-inverse with numpy.linalg.pinv's cutoff), the strided p x K products, and the fold
statistics / fold algebra of the batched cross-validation."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(gpu_required):
    import mbpls_amd
    return mbpls_amd.get_context()


@pytest.mark.parametrize("K,B,q,kind", [(1, 1, 1, "regular"), (5, 2, 1, "regular"), (10, 1, 3, "triangular"),
                                        (32, 3, 2, "regular"), (8, 1, 1, "rank deficient"), (6, 2, 1, "ill conditioned")])
def test_finalise_small_is_numpy_pinv(ctx, K, B, q, kind):
    """M = D (C D)^+ and MV = M V^T as numpy computes them (mbpls.py:988-989): regular, (nearly) upper triangular like
    P^T W of a PLS fit, rank deficient (singular values below 1e-15 sigma_max are dropped, as numpy.linalg.pinv does)
    and ill conditioned"""
    from mbpls_amd import _capi
    rng = np.random.RandomState(K * 7 + B)
    Cm = rng.standard_normal((K, K))
    if kind == "triangular":
        Cm = np.triu(Cm) + np.eye(K) * 3 + 1e-16 * rng.standard_normal((K, K))
    elif kind == "rank deficient":
        Cm[:, -1] = Cm[:, 0] * 2.0 - Cm[:, 1]
    elif kind == "ill conditioned":
        u, _, vt = np.linalg.svd(Cm)
        Cm = (u * np.logspace(0, -10, K)).dot(vt)
    scal = np.zeros((K, 4, B)); scal[:, 0, :] = rng.rand(K, B) + 0.5
    V = rng.standard_normal((q, K))
    D = 1.0 / np.sqrt(scal[:, 0, :].sum(axis=1))
    want_M = D[:, None] * np.linalg.pinv(Cm * D[None, :])
    dC, dS, dV = ctx.from_host(Cm.ravel()), ctx.from_host(scal.ravel()), ctx.from_host(np.asfortranarray(V))
    dM, dMV = ctx.empty(K * K), ctx.empty(K * q)
    _capi.check(ctx.lib.mbpls_hip_finalise_small(ctx.handle, K, B, q, dC.ptr, dS.ptr, dV.ptr, dM.ptr, dMV.ptr))
    M, MV = dM.get().reshape(K, K), dMV.get().reshape(K, q)
    tol = 1e-11 if kind != "ill conditioned" else 1e-5       # (relative to a 1e10 condition number)
    assert np.max(np.abs(M - want_M)) <= tol * np.max(np.abs(want_M)), kind
    assert np.max(np.abs(MV - want_M.dot(V.T))) <= tol * max(np.max(np.abs(want_M.dot(V.T))), 1e-300)


def test_strided_products_of_fold_stacks(ctx):
    """cross_product_dev / right_multiply_dev on p x K matrices whose columns are F p apart (fold f of a
    component-major stack): what the batched cross-validation's end-of-fit algebra uses"""
    from mbpls_amd import _capi
    p, F, K = 4001, 5, 6
    rng = np.random.RandomState(3)
    P, W = rng.standard_normal((p, K * F)), rng.standard_normal((p, K * F))      # column k F + f
    dP, dW = ctx.from_host(np.asfortranarray(P)), ctx.from_host(np.asfortranarray(W))
    dCm, dMv, dB = ctx.empty(K * K), ctx.from_host(np.arange(1.0, K + 1)), ctx.empty(p)
    for f in (0, 3, 4):
        _capi.check(ctx.lib.mbpls_hip_cross_product_dev(ctx.handle, p, dP.ptr + 8 * p * f, p * F, K, dW.ptr + 8 * p * f,
                                                        p * F, K, dCm.ptr))
        want = P[:, f::F].T.dot(W[:, f::F])
        assert np.max(np.abs(dCm.get().reshape(K, K) - want)) < 1e-11 * np.max(np.abs(want))
        _capi.check(ctx.lib.mbpls_hip_right_multiply_dev(ctx.handle, p, dW.ptr + 8 * p * f, p * F, K, dMv.ptr, 1, dB.ptr, p))
        wb = W[:, f::F].dot(np.arange(1.0, K + 1))
        assert np.max(np.abs(dB.get() - wb)) < 1e-12 * np.max(np.abs(wb))


@pytest.mark.parametrize("n,p,blocks", [(64, 700, 1), (150, 3001, 2), (1100, 513, 1), (9000, 300, 2)])
def test_fold_statistics_and_fold_algebra(ctx, n, p, blocks):
    """mbpls_hip_xt_v_centred (centred sums and sums of squares over row sets), mbpls_hip_cv_fold_stats (StandardScaler
    of every fold's training rows, constant columns, entries flagged for cancellation) and mbpls_hip_cv_weights (both
    modes) against numpy, for the few-samples, register-resident and long-column kernel families"""
    from mbpls_amd import DeviceMatrix, _capi
    from oracle import mbpls_oracle as orc
    rng = np.random.RandomState(n + p)
    X = rng.standard_normal((n, p)) * (1 + rng.rand(p)) + 3 * rng.rand(p)
    X[:, 4] = 2.5                                       # constant everywhere
    X[:, 7] = 1.0; X[3, 7] = 9.0                        # constant on the training rows of a fold that holds out row 3
    F = 6
    folds = [np.arange(n)[np.arange(n) % F == f] for f in range(F)]
    M = np.ones((n, F))
    for f, te in enumerate(folds):
        M[te, f] = 0.0
    offs = [0, p] if blocks == 1 else [0, p // 3, p]
    Xd = DeviceMatrix(ctx, n, p, offs).upload(X)
    mu0, var0 = X.mean(axis=0), X.var(axis=0)
    dmu0, dvar0 = ctx.from_host(mu0), ctx.from_host(var0)
    dH, dnf = ctx.from_host(1.0 - M), ctx.from_host(M.sum(axis=0))
    h1, h2, dmean, drs = (ctx.empty((p, F)) for _ in range(4))
    drisky = ctx.empty(1)
    _capi.check(ctx.lib.mbpls_hip_xt_v_centred(Xd.handle, dmu0.ptr, 0, dH.ptr, n, F, h1.ptr, p))
    _capi.check(ctx.lib.mbpls_hip_xt_v_centred(Xd.handle, dmu0.ptr, 1, dH.ptr, n, F, h2.ptr, p))
    Xc = X - mu0
    assert np.max(np.abs(h1.get() - Xc.T.dot(1.0 - M))) < 1e-10 * np.max(np.abs(Xc)) * n
    assert np.max(np.abs(h2.get() - (Xc * Xc).T.dot(1.0 - M))) < 1e-10 * np.max(Xc * Xc) * n
    _capi.check(ctx.lib.mbpls_hip_cv_fold_stats(Xd.handle, F, dmu0.ptr, dvar0.ptr, h1.ptr, h2.ptr, dnf.ptr, dmean.ptr,
                                                drs.ptr, drisky.ptr))
    mean, rs, risky = dmean.get(), drs.get(), int(drisky.get().view(np.uint64)[0])
    f3 = 3 % F
    assert risky >= 1 and rs[7, f3] == -1.0             # the cancellation case is flagged, not silently wrong
    for f in range(F):
        sc = orc.ColumnScaler().fit(X[M[:, f] != 0.0])
        ok = rs[:, f] > 0.0
        assert ok.sum() >= p - 2
        np.testing.assert_allclose(mean[ok, f], sc.mean_[ok], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(1.0 / rs[ok, f], sc.scale_[ok], rtol=1e-9)
        assert rs[4, f] == 1.0                           # constant column: scale 1
    # the fold algebra, sweep mode and prediction mode
    rs = np.where(rs > 0.0, rs, 1.0)
    drs.set(rs)
    G, su, alpha = rng.standard_normal((p, F)), rng.standard_normal(F), rng.rand(F) + 0.5
    dG, dsu, dal = ctx.from_host(G), ctx.from_host(su), ctx.from_host(alpha)
    dg, dC, dsums = ctx.empty((p, F)), ctx.empty((p, F)), ctx.empty((len(offs) - 1) * F * 2)
    _capi.check(ctx.lib.mbpls_hip_cv_weights(Xd.handle, F, 0, dG.ptr, dmean.ptr, drs.ptr, dsu.ptr, dal.ptr, dg.ptr, dC.ptr,
                                             dsums.ptr))
    g = alpha * (G - mean * su) * rs
    np.testing.assert_allclose(dg.get(), g, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(dC.get(), g * rs, rtol=1e-12, atol=1e-13)
    sums = dsums.get().reshape(len(offs) - 1, F, 2)
    for b in range(len(offs) - 1):
        r = slice(offs[b], offs[b + 1])
        np.testing.assert_allclose(sums[b, :, 0], (g[r] ** 2).sum(axis=0), rtol=1e-11)
        np.testing.assert_allclose(sums[b, :, 1], (mean[r] * g[r] * rs[r]).sum(axis=0), rtol=1e-9, atol=1e-9)
    _capi.check(ctx.lib.mbpls_hip_cv_weights(Xd.handle, F, 1, dG.ptr, dmean.ptr, drs.ptr, None, None, dg.ptr, dC.ptr,
                                             dsums.ptr))
    np.testing.assert_allclose(dC.get(), G * rs, rtol=1e-12, atol=1e-13)
    sums = dsums.get().reshape(len(offs) - 1, F, 2)
    np.testing.assert_allclose(sums[:, :, 1].sum(axis=0), (mean * G * rs).sum(axis=0), rtol=1e-9, atol=1e-9)
    Xd.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,p", [(1, 1), (33, 7), (64, 96), (257, 1000), (1000, 257)])
def test_device_transpose_and_xtx_through_the_gram_kernel(n, p):
    """mbpls_hip_mat_transpose: the p x n image of an n x p matrix, bit for bit; the Gram kernel applied to it is
    X^T X (the n >= p branches, mbpls.py:586)."""
    import mbpls_amd
    from mbpls_amd import DeviceMatrix
    ctx = mbpls_amd.get_context()
    rng = np.random.default_rng(n * 131 + p)
    X = rng.standard_normal((n, p))
    Xd = DeviceMatrix(ctx, n, p)
    Xd.upload(X)
    Xt = Xd.transposed()
    assert (Xt.n, Xt.p) == (p, n)
    assert np.array_equal(Xt.download(), X.T)
    ldg = p + (p & 1)
    G = ctx.empty(ldg * p, zero=True)
    Xt.gram(G, ldg)
    got = G.get().reshape(p, ldg)[:, :p]
    want = X.T.dot(X)
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
    assert np.array_equal(got, got.T)
    G.free(); Xt.close(); Xd.close()


@pytest.mark.gpu
@pytest.mark.parametrize("p,K", [(1, 1), (1000, 5), (70_001, 12), (600_000, 3)])
def test_r_update_builds_w_times_inverse_of_a_triangular_matrix(p, K):
    """mbpls_hip_queue_r_update / mbpls_hip_flush_r_update: R[:, k] = (W[:, k] - sum_{i<k} R[:, i] C[i, k]) / C[k, k] and
    beta += R[:, k] v_k -- R_ = W (P^T W)^+ (mbpls.py:988) for the upper-triangular P^T W of a PLS fit, one column per
    call.  Here each update goes by itself (no column pass to carry it)."""
    import mbpls_amd
    from mbpls_amd import _capi
    ctx = mbpls_amd.get_context()
    lib = ctx.lib
    rng = np.random.default_rng(p + K)
    W = np.asfortranarray(rng.standard_normal((p, K)))
    Cm = np.triu(rng.standard_normal((K, K))) + 3.0 * np.eye(K)
    v = rng.standard_normal(K)
    dW, dC, dv = ctx.from_host(W.ravel(order="F")), ctx.from_host(np.asfortranarray(Cm).ravel(order="F")), ctx.from_host(v)
    dR, dbeta = ctx.empty(p * K), ctx.empty(p)
    for k in range(K):
        _capi.check(lib.mbpls_hip_queue_r_update(ctx.handle, p, dW.ptr + 8 * p * k, dR.ptr, p, k, dC.ptr + 8 * K * k,
                                                 dv.ptr + 8 * k, dbeta.ptr))
        _capi.check(lib.mbpls_hip_flush_r_update(ctx.handle))
        _capi.check(lib.mbpls_hip_flush_r_update(ctx.handle))          # nothing queued: a no-op
    R = dR.get().reshape(K, p).T
    want = W.dot(np.linalg.inv(Cm))
    assert np.abs(R - want).max() <= 1e-12 * np.abs(want).max()
    assert np.abs(dbeta.get() - want.dot(v)).max() <= 1e-12 * max(1.0, np.abs(want.dot(v)).max())
    for a in (dW, dC, dv, dR, dbeta):
        a.free()
