"""Host-side solver logic (mbpls_amd/solvers.py) against the reference fixtures, with the
test-only numpy engine standing in for the HIP kernels (no GPU here).  The GPU parity tests
(test_gpu_parity.py) run the same drivers on the real HipEngine."""
import numpy as np
import pytest

from mbpls_amd import solvers
from mbpls_amd.scaler import ColumnScaler
from tests import util
from tests.numpy_engine import NumpyEngine

TOL = 1e-9


def run_driver(X, Y, method, K, standardize=True, calc_all=True, norm=2, comm=None, sparse=False):
    blocks = X if isinstance(X, list) else [X]
    Y = np.asarray(Y, dtype=np.float64)
    if Y.ndim == 1:
        Y = Y.reshape(-1, 1)
    if standardize:
        Y = ColumnScaler().fit(Y).transform(Y)
    eng = NumpyEngine(comm).load_host_blocks(blocks)
    eng.sparse = sparse
    if method == "NIPALS":
        return solvers.fit_nipals(eng, Y, K, standardize, 1e-14, norm, calc_all)
    if method == "UNIPALS":
        return solvers.fit_unipals(eng, Y, K, standardize)
    if method == "KERNEL":
        return solvers.fit_kernel(eng, Y, K, standardize, calc_all)
    return solvers.fit_simpls(eng, Y, K, standardize)


FIT_ONLY_SKIP = ("predict", "Ts_new", "T_new", "U_new")


@pytest.mark.parametrize("fixture", ["matrix_pgtn", "matrix_ngtp"])
def test_matrix(fixture):
    g = util.load(fixture)
    splits = np.cumsum(g["splits"])[:-1]
    for q in (1, 3):
        X, Y = g[f"in/q{q}/X"], g[f"in/q{q}/Y"]
        for blocks in ("single", "multi"):
            Xin = X if blocks == "single" else list(np.split(X, splits, axis=1))
            for m in ("NIPALS", "UNIPALS", "KERNEL", "SIMPLS"):
                for std in (True, False):
                    for ca in (True, False):
                        pre = f"ref/q{q}/{blocks}/{m}/std{int(std)}/ca{int(ca)}"
                        res = run_driver(Xin, Y, m, 3, std, ca)
                        util.compare(util.model_to_flat(res), util.sub(g, pre), TOL, skip=FIT_ONLY_SKIP, what=pre)


def test_reference_csv_inputs():
    for fixture, methods in (("ref_golden_pgtn", ["UNIPALS", "NIPALS", "KERNEL", "SIMPLS"]),
                             ("ref_golden_ngtp", ["UNIPALS", "KERNEL"])):
        g = util.load(fixture)
        X = [g["x1_train"], g["x2_train"]]
        for m in methods:
            res = run_driver(X, g["y_train"], m, 2)
            util.compare(util.model_to_flat(res), util.sub(g, f"ref/{m}/full1"), TOL, skip=FIT_ONLY_SKIP, what=m)


def test_nipals_iterations_q1(monkeypatch):
    """the reference's iteration structure (two passes per component for one response), selected by the switch"""
    g = util.load("quickstart_c1")
    monkeypatch.setenv("MBPLS_NIPALS_Q1_FIXED_POINT", "0")
    res = run_driver(g["x"], g["y"], "NIPALS", 2)
    assert res.nipals_iterations_ == [2, 2]
    util.compare(util.model_to_flat(res), util.sub(g, "ref/NIPALS"), TOL, skip=FIT_ONLY_SKIP)


def test_q1_fixed_point_is_the_default(monkeypatch):
    """For one response the second NIPALS pass only re-derives the first (u is y / v whatever ts was): the default route
    makes one pass per component; every attribute must be what the reference's two passes give."""
    g = util.load("quickstart_c1")
    monkeypatch.delenv("MBPLS_NIPALS_Q1_FIXED_POINT", raising=False)
    res = run_driver(g["x"], g["y"], "NIPALS", 2)
    assert res.nipals_iterations_ == [1, 1]
    util.compare(util.model_to_flat(res), util.sub(g, "ref/NIPALS"), TOL, skip=FIT_ONLY_SKIP)
    g = util.load("matrix_pgtn")
    splits = np.cumsum(g["splits"])[:-1]
    X, Y = g["in/q1/X"], g["in/q1/Y"]
    res = run_driver(list(np.split(X, splits, axis=1)), Y, "NIPALS", 3)
    util.compare(util.model_to_flat(res), util.sub(g, "ref/q1/multi/NIPALS/std1/ca1"), TOL, skip=FIT_ONLY_SKIP)


def test_prefault_leaves_contents_alone():
    """engine._prefault maps the pages of a fresh attribute buffer in a helper thread while a copy may already be writing
    into it: whatever is in the array stays (madvise(MADV_POPULATE_WRITE), or nothing at all where unsupported)."""
    import time
    from mbpls_amd import engine
    a = np.empty((300_000, 10), order="F")
    a[:] = np.arange(10.0)
    a[::4096, 0] = -1.0
    engine._prefault(a)
    b = np.empty((1 << 16,))                # below the size limit: left alone
    engine._prefault(b)
    for _ in range(3):                      # keep writing while the helper runs
        a[::1000, 3] = 7.0
        time.sleep(0.01)
    q = engine._PREFAULT.get("queue")
    if q is not None:
        t0 = time.time()
        while not q.empty() and time.time() - t0 < 10:
            time.sleep(0.01)
        time.sleep(0.05)
    assert np.all(a[::4096, 0] == -1.0) and np.all(a[1::1000, 3] == 3.0) and np.all(a[::1000, 3] == 7.0)
    assert np.all(a[:, 9] == 9.0)


@pytest.mark.parametrize("n,q,seed", [(9000, 2, 0), (20000, 3, 1), (12000, 5, 2)])
def test_subspace_loop_in_q_space_matches_the_n_vector_loop(n, q, seed):
    """solvers._subspace_loop_qspace (tall data, several responses: the NIPALS inner loop on q x q matrices, stop rule on
    the difference of the coefficient vectors) against the n-vector form of the same loop: iteration counts and iterates."""
    rng = np.random.default_rng(seed)
    T = rng.standard_normal((n, q + 1))
    Y = T[:, :q] @ rng.standard_normal((q, q)) + 0.05 * rng.standard_normal((n, q))
    Y = (Y - Y.mean(0)) / Y.std(0)
    X = T @ rng.standard_normal((q + 1, 40)) + 0.3 * rng.standard_normal((n, 40))
    X -= X.mean(0)
    G = X @ (X.T @ Y)
    c0 = np.zeros((q, 1)); c0[0] = 1.0
    a = solvers._subspace_loop_host(G, Y, c0, 2, 1e-14, 10000)
    b = solvers._subspace_loop_qspace(G, Y, c0, 1e-14, 10000)
    assert not a[1] and not b[1]
    assert abs(a[0] - b[0]) <= 2, (a[0], b[0])               # the stop test sits at the rounding floor of both forms
    for i in (2, 3, 4, 5):
        assert np.abs(a[i] - b[i]).max() <= 1e-11 * np.abs(a[i]).max()


def test_scalers_are_sklearn_standard_scalers():
    """x_scalers_ / y_scaler_ of the reference are sklearn StandardScaler objects (mbpls.py:307,316,325): the product's
    carry the statistics of the HIP passes but ARE StandardScalers (isinstance, clone, pickle) with the same fitted
    attributes and the same numbers as sklearn's own fit."""
    import pickle
    from sklearn.base import clone
    from sklearn.preprocessing import StandardScaler
    from mbpls_amd.scaler import ColumnScaler
    rng = np.random.RandomState(0)
    X = rng.rand(40, 5) * 3 + 7
    X[:, 2] = 4.0                                   # constant: scale_ 1, var_ 0
    mine, ref = ColumnScaler().fit(X), StandardScaler().fit(X)
    assert isinstance(mine, StandardScaler)
    for attr in ("mean_", "scale_", "var_"):
        np.testing.assert_allclose(getattr(mine, attr), getattr(ref, attr), rtol=1e-13, atol=1e-15)
    assert mine.n_samples_seen_ == ref.n_samples_seen_ and mine.n_features_in_ == ref.n_features_in_
    np.testing.assert_allclose(mine.transform(X), ref.transform(X), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(mine.inverse_transform(mine.transform(X)), X, rtol=1e-12)
    carried = ColumnScaler(ref.mean_, ref.scale_, 40, ref.var_)          # statistics produced elsewhere (the GPU passes)
    np.testing.assert_allclose(pickle.loads(pickle.dumps(carried)).transform(X), ref.transform(X), rtol=1e-12, atol=1e-13)
    assert isinstance(clone(carried), ColumnScaler)


def test_one_response_noise_floor_is_a_stagnation_rule(two_pass_nipals):
    """advisor finding (round 3): a fixed cut-off at four iterations also stopped components the reference finishes in
    five.  The one-response loop gives up only when ||ts_old - ts|| has STOPPED shrinking; iteration counts equal the
    oracle's wherever the oracle converges, and a difference that still shrinks is never cut."""
    import warnings
    from oracle import mbpls_oracle as orc
    st, pd = 0, None
    for run, d in [(2, 1e-10), (3, 1e-12), (4, 3e-14), (5, 1.2e-14)]:       # shrinking: left to converge
        give_up, st = solvers.q1_noise_floor(run, d, pd, st)
        pd = d
        assert not give_up
    st, pd, stops = 0, None, []
    for run, d in [(2, 3e-14), (3, 2.5e-14), (4, 4e-14), (5, 3e-14)]:       # a plateau: stopped at the fourth iteration
        give_up, st = solvers.q1_noise_floor(run, d, pd, st)
        pd = d
        stops.append(give_up)
    assert stops == [False, False, True, True]
    g, st = solvers.q1_noise_floor(3, np.array([1e-14, 3e-14]), np.array([1e-10, 2e-14]), np.zeros(2, dtype=int))
    assert list(st) == [0, 1] and not g.any()                               # per cross-validation fold
    rng = np.random.RandomState(1)
    X, Y = rng.standard_normal((60, 400)), rng.standard_normal((60, 1))     # p >> n: y is exhausted after five components
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = orc.fit(X, Y, n_components=6, method="NIPALS", faithful=False, max_iter=60)
        res = run_driver(X, Y, "NIPALS", 6)
    for it_ref, it, conv in zip(ref.nipals_iterations_, res["nipals_iterations_"], res["nipals_converged_"]):
        if it_ref < 60:
            assert it == it_ref and conv
        else:
            assert not conv and it <= solvers.NIPALS_Q1_NOISE_FLOOR_MAX


def test_leading_singular_vector_of_a_tall_thin_matrix():
    """solvers._tlsv: the leading left singular vector through the q x q Gram matrix (tall thin operands of the n >= p
    branches, mbpls.py:396-400 / :590) equals LAPACK's to rounding -- also with a wide spread of singular values -- and a
    nearly double leading singular value takes the SVD itself."""
    from mbpls_amd import solvers
    rng = np.random.RandomState(3)
    for m, q, spread in ((500, 3, 1.0), (20000, 10, 1e-6), (4096, 16, 1e-10)):
        A = rng.standard_normal((m, q)) * np.logspace(0, np.log10(spread), q)
        want = np.linalg.svd(A, full_matrices=False)[0][:, 0:1]
        got = solvers._tlsv(A)
        s = np.sign((want * got).sum())
        assert np.abs(got * s - want).max() < 1e-13
        assert abs(np.linalg.norm(got) - 1.0) < 1e-14
    # two equal leading singular values: the Gram route is not taken (whatever LAPACK returns is returned)
    Q = np.linalg.qr(rng.standard_normal((300, 4)))[0]
    A = Q * np.array([2.0, 2.0, 1.0, 0.5])
    assert np.array_equal(solvers._tlsv(A), np.linalg.svd(A, full_matrices=False)[0][:, 0:1])
    # one column: just normalised
    a = rng.standard_normal((50, 1))
    assert np.allclose(solvers._tlsv(a), a / np.linalg.norm(a))


def test_host_thread_limits_are_set_and_put_back():
    """runtime.host_threads: BLAS / OpenMP pools larger than the CPU quota are limited for the duration of a fit and
    restored afterwards (what threadpoolctl.limit does, without its per-call inventory)."""
    from mbpls_amd import runtime
    runtime._thread_limits()
    tp = runtime._state.get("tp")
    if tp is None or not tp.lib_controllers:
        pytest.skip("no thread pools to control here")
    before = [lc.num_threads for lc in tp.lib_controllers]
    old = runtime._state["tp_limit"]
    runtime._state["tp_limit"] = 1
    try:
        with runtime.host_threads() as h:
            assert all(lc.num_threads == 1 for lc in tp.lib_controllers if (lc.num_threads is not None))
            assert len(h.changed) == sum(1 for b in before if b is not None and b > 1)
    finally:
        runtime._state["tp_limit"] = old
    assert [lc.num_threads for lc in tp.lib_controllers] == before


def test_random_problems_through_the_host_drivers(monkeypatch):
    """A randomised sweep of the host drivers (solvers.py on the numpy engine -- no GPU here): shapes, block splits, the
    four methods, both routes of one-response NIPALS, calc_all, +- standardisation -- against the oracle.  (The GPU suite
    runs the same kind of sweep on the HIP engine, tests/fuzz.py.)"""
    from oracle import mbpls_oracle as orc
    rng = np.random.RandomState(12)
    worst = 0.0
    for c in range(60):
        n = int(rng.choice([rng.randint(6, 40), rng.randint(40, 200)]))
        p = int(rng.choice([rng.randint(2, 30), rng.randint(30, 400)]))
        q = int(rng.choice([1, 1, 2, 3]))
        method = str(rng.choice(["NIPALS", "NIPALS", "UNIPALS", "KERNEL", "SIMPLS"]))
        K = int(min(rng.randint(1, 5), max(1, min(n // 4, p))))
        std, calc_all = bool(rng.rand() < 0.7), bool(rng.rand() < 0.8)
        nb = int(min(p, rng.randint(1, 4)))
        cuts = np.sort(rng.choice(np.arange(1, p), nb - 1, replace=False)) if nb > 1 else np.array([], dtype=int)
        X, Y = orc.make_latent(n, p, q, r=min(4, p), seed=3000 + c)
        X = X + rng.uniform(-3, 3, size=p) * (rng.rand() < 0.5)
        Xin = X if nb == 1 else [np.ascontiguousarray(b) for b in np.split(X, cuts, axis=1)]
        monkeypatch.setenv("MBPLS_NIPALS_Q1_FIXED_POINT", "0" if rng.rand() < 0.5 else "1")
        ref = orc.fit(Xin, Y, n_components=K, method=method, standardize=std, full_svd=True, calc_all=calc_all, max_iter=3000)
        if method == "NIPALS" and max(ref.nipals_iterations_) >= 3000:
            continue                                       # (the reference's own loop does not converge: nothing to compare)
        res = run_driver(Xin, Y, method, K, std, calc_all)
        what = f"case {c}: n={n} p={p} q={q} K={K} {method} std={std} calc_all={calc_all} blocks={nb}"
        worst = max(worst, util.compare(util.model_to_flat(res), util.model_to_flat(ref), 1e-6, what=what))
    assert worst < 1e-7, worst


def test_missing_values_reference_fixtures():
    """sparse_data=True through the host drivers (masked passes of csrc/colpass_nan.hip restated in the numpy engine):
    the attributes captured from the imported reference, missing values in X (sparse.npz) and in Y as well
    (sparse_y.npz: start vector :832-836, masked Y loadings :891-897, masked Y scores :900-906)."""
    worst = 0.0
    for fixture, qs in (("sparse", (1, 2)), ("sparse_y", (2, 3))):
        g = util.load(fixture)
        for q in qs:
            X, Y = g[f"in/q{q}/X"], g[f"in/q{q}/Y"]
            for blocks in ("single", "multi"):
                Xin = X if blocks == "single" else [np.ascontiguousarray(X[:, :40]), np.ascontiguousarray(X[:, 40:])]
                for std in (True, False):
                    pre = f"ref/q{q}/{blocks}/std{int(std)}"
                    res = run_driver(Xin, Y, "NIPALS", 3, std, True, sparse=True)
                    skip = FIT_ONLY_SKIP + ("y_mean", "y_scale")
                    worst = max(worst, util.compare(util.model_to_flat(res), util.sub(g, pre), TOL, skip=skip,
                                                    what=f"{fixture} {pre}"))
    assert worst < 1e-9, worst


def test_random_problems_with_missing_values_through_the_host_drivers(monkeypatch):
    """... and a randomised sweep against the NaN-aware oracle: shapes, block splits, +- standardisation, one response
    on both routes (at its fixed point after one pass, solvers.fit_nipals; or with the reference's confirming pass),
    several responses"""
    import warnings
    from oracle import mbpls_oracle as orc
    from oracle import mbpls_oracle_sparse as ors
    rng = np.random.RandomState(21)
    worst, done = 0.0, 0
    for c in range(40):
        n, p = int(rng.randint(20, 120)), int(rng.randint(20, 200))
        q = int(rng.choice([1, 1, 2, 3]))
        K = int(min(rng.randint(1, 4), n // 4))
        std = bool(rng.rand() < 0.7)
        nb = int(rng.randint(1, 4))
        cuts = np.sort(rng.choice(np.arange(1, p), nb - 1, replace=False)) if nb > 1 else np.array([], dtype=int)
        X, Y = orc.make_latent(n, p, q, r=4, seed=7000 + c)
        X[rng.rand(n, p) < rng.choice([0.01, 0.05])] = np.nan
        Xin = X if nb == 1 else [np.ascontiguousarray(b) for b in np.split(X, cuts, axis=1)]
        one_pass = bool(rng.rand() < 0.5)
        monkeypatch.setenv("MBPLS_NIPALS_Q1_FIXED_POINT", "1" if one_pass else "0")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = ors.fit(Xin, Y, n_components=K, standardize=std, max_iter=3000)
            if max(ref.nipals_iterations_) >= 3000:
                continue                                   # (the reference's own loop does not converge here)
            res = run_driver(Xin, Y, "NIPALS", K, std, True, sparse=True)
        what = f"case {c}: n={n} p={p} q={q} K={K} std={std} blocks={nb} one_pass={one_pass}"
        worst = max(worst, util.compare(util.model_to_flat(res), util.model_to_flat(ref), 1e-6, what=what))
        if q == 1:
            assert list(res.nipals_iterations_) == ([1] * K if one_pass else list(ref.nipals_iterations_)), what
        done += 1
    assert done >= 30 and worst < 1e-7, (done, worst)


def test_adjacent_copies_are_merged():
    """_ResidentQueue._merged: columns k, k + 1 of the same device slot and the same F-ordered host array become one
    copy; anything not adjacent on BOTH sides stays apart"""
    from mbpls_amd.engine import _ResidentQueue
    A = np.zeros((100, 4), order="F")
    B = np.zeros((100, 2), order="F")
    base = 1 << 20
    copies = [(A[:, 1], base + 800), (A[:, 0], base), (A[:, 2], base + 1600), (B[:, 0], base + 2400), (A[:, 3], base + 9000),
              (B[:, 1], base + 2400 + 800)]
    merged = _ResidentQueue._merged(copies)
    sizes = sorted((m[0].nbytes, m[1] - base) for m in merged)
    assert sizes == [(800, 9000), (1600, 2400), (2400, 0)], sizes
    assert merged[0][0].ctypes.data == A.ctypes.data


def test_large_host_buffers_are_handed_out_again_only_when_nobody_refers_to_them():
    """engine._pooled_host_array: the p x K attributes of a fit are views of owners the context keeps; an owner is handed
    out again once every array, slice and view of it is gone (numpy collapses the base chain onto the owner) -- a caller
    that fits in a loop then neither pays the munmap of the previous model nor the page faults of the next."""
    from mbpls_amd import engine

    class Ctx:
        pass

    ctx = Ctx()
    small, mapped = engine._pooled_host_array(ctx, (1000, 3))
    assert not mapped and small.flags.f_contiguous and not getattr(ctx, "_host_pool", [])       # below HOST_POOL_MIN
    shape = (200_000, 4)                                                                         # 6.4 MB
    a, mapped = engine._pooled_host_array(ctx, shape)
    assert not mapped and a.shape == shape and a.flags.f_contiguous and a.dtype == np.float64
    a[:] = 7.0
    addr = a.ctypes.data
    b, mapped = engine._pooled_host_array(ctx, shape)
    assert not mapped and b.ctypes.data != addr                    # the first one is in use
    col = a[:, 2]                                                  # what a caller keeps of a model it has let go
    del a
    c, mapped = engine._pooled_host_array(ctx, shape)
    assert not mapped and c.ctypes.data != addr and np.all(col == 7.0)
    del col
    d, mapped = engine._pooled_host_array(ctx, shape)
    assert mapped and d.ctypes.data == addr                        # nobody refers to it any more: handed out again
    e, mapped = engine._pooled_host_array(ctx, (190_000, 4))       # a slightly smaller request takes a free owner too
    assert not mapped or e.ctypes.data != addr
    del b, c, d, e
    f, mapped = engine._pooled_host_array(ctx, (190_000, 4))
    assert mapped and f.shape == (190_000, 4) and f.flags.f_contiguous
    g, mapped = engine._pooled_host_array(ctx, (20_000, 4))        # far smaller: a new owner, the big ones stay whole
    assert not mapped
    assert sum(r.nbytes for r in ctx._host_pool) <= engine.HOST_POOL_BYTES


def test_a_full_pool_of_other_sizes_makes_room(monkeypatch):
    """... and a caller that moves on to another problem size is not stuck with a pool full of the old one: owners at
    rest go, oldest first, owners in use stay (wide few-sample fits after larger ones paid the page faults of fresh
    arrays in every fit: 32 ms instead of 15)"""
    from mbpls_amd import engine

    class Ctx:
        pass

    ctx = Ctx()
    monkeypatch.setattr(engine, "HOST_POOL_BYTES", 40 << 20)
    big = (1_000_000, 2)                                                     # 16 MB
    a, _ = engine._pooled_host_array(ctx, big)
    b, _ = engine._pooled_host_array(ctx, big)
    kept_addr = b.ctypes.data
    del a                                                                    # a's owner is at rest, b's is in use
    assert len(ctx._host_pool) == 2
    other = (700_000, 2)                                                     # 11.2 MB: fits no owner, 43 MB in all
    c, mapped = engine._pooled_host_array(ctx, other)
    assert not mapped and len(ctx._host_pool) == 2                           # a's owner went, c's came
    assert sorted(r.nbytes for r in ctx._host_pool) == [8 * 700_000 * 2, 8 * 1_000_000 * 2]
    del c
    d, mapped = engine._pooled_host_array(ctx, other)
    assert mapped                                                            # the new size is pooled now
    assert b.ctypes.data == kept_addr and sum(r.nbytes for r in ctx._host_pool) <= engine.HOST_POOL_BYTES
    # nothing at rest and no room: handed out unpooled, the pool stays as it is
    e, mapped = engine._pooled_host_array(ctx, big)
    assert not mapped and len(ctx._host_pool) == 2


def test_send_cols_of_the_plain_engine_copies_and_normalises_like_the_copier():
    """EngineBase.send_cols / send_slot / wait_sent: what the solvers call for the finished columns of a component -- on the
    HIP engine a job of the copier thread (csrc/copier.hip), here plain copies; ``unit`` forms w_b = w~_b / ||w~_b||
    (mbpls.py:860) block by block, in place or into a second array, and leaves every other column alone"""
    from tests.numpy_engine import NumpyEngine

    rng = np.random.RandomState(3)
    eng = NumpyEngine().load_host_blocks([rng.randn(9, 4), rng.randn(9, 1), rng.randn(9, 6)])
    K = 3
    eng.slot("W", K)[...] = rng.randn(eng.p, K)
    eng.slot("P", K)[...] = rng.randn(eng.p, K)
    nu = np.array([2.0, 0.5, 3.0])
    want = eng.slots["W"][:, 1].copy()
    for b in range(3):
        want[eng.block_offsets[b]:eng.block_offsets[b + 1]] /= nu[b]
    for in_place in (False, True):
        raw = np.full((eng.p, K), np.nan, order="F")
        unit = raw if in_place else np.full((eng.p, K), np.nan, order="F")
        P_h = np.full((eng.p, K), np.nan, order="F")
        ticket = eng.send_cols([("W", 1, raw), ("P", 0, P_h)], unit=(raw, unit, 1, nu))
        eng.wait_sent(ticket)
        assert np.array_equal(unit[:, 1], want)
        if not in_place:
            assert np.array_equal(raw[:, 1], eng.slots["W"][:, 1])
        assert np.array_equal(P_h[:, 0], eng.slots["P"][:, 0])
        assert np.isnan(raw[:, [0, 2]]).all() and np.isnan(P_h[:, 1:]).all()
    whole = np.empty((eng.p, K), order="F")
    eng.wait_sent(eng.send_slot("P", whole))
    assert np.array_equal(whole, eng.slots["P"])
