"""The copier thread of the C ABI (csrc/copier.hip, mbpls_hip_copier_submit / _wait): finished device columns are carried
to host arrays behind an event, small ones through pinned bounce buffers and out of them piece by piece on several
threads, with the block normalisation w_b = w~_b / ||w~_b|| (mbpls.py:860) done on the way."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx():
    import mbpls_amd
    return mbpls_amd.get_context()


@pytest.mark.parametrize("p", [1, 7, 8191, 8192, 8193, 125_000, 600_001])
def test_copies_arrive_bit_exact_in_submission_order(p):
    ctx = _ctx()
    rng = np.random.default_rng(p)
    K = 5
    src = np.asfortranarray(rng.standard_normal((p, K)))
    dev = ctx.from_host(src.ravel(order="F"))
    tickets, outs = [], []
    for rep in range(3):
        out = np.full((p, K), np.nan, order="F")
        # column copies (every offset: columns of an F-ordered array are only 8-byte aligned) ...
        t = ctx.copier_submit([(out[:, k], dev.ptr + 8 * p * k) for k in range(K - 1)])
        # ... and a second job for the last column
        t2 = ctx.copier_submit([(out[:, K - 1], dev.ptr + 8 * p * (K - 1))])
        assert t2 == t + 1
        tickets.append(t2)
        outs.append(out)
    ctx.copier_wait(tickets[-1])
    for out in outs:
        assert np.array_equal(out, src)


@pytest.mark.parametrize("in_place", [False, True])
@pytest.mark.parametrize("p,offs", [(1000, [0, 1000]), (125_000, [0, 40_000, 40_001, 125_000]), (70_000, [0, 8192, 65536, 70_000])])
def test_block_normalisation_rides_on_the_copy(p, offs, in_place):
    ctx = _ctx()
    rng = np.random.default_rng(p + in_place)
    B = len(offs) - 1
    w = rng.standard_normal(p)
    dev = ctx.from_host(w)
    nu2_h, _ = ctx.mapped_scalars(64)
    nu2 = rng.uniform(0.5, 4.0, B)
    nu2_h[8:8 + B] = nu2
    raw = np.full((p, 2), np.nan, order="F")
    unit = raw if in_place else np.full((p, 2), np.nan, order="F")
    offs64 = np.asarray(offs, dtype=np.int64)
    t = ctx.copier_submit([(raw[:, 1], dev.ptr)], scale=(raw[:, 1], unit[:, 1], offs64, nu2_h.ctypes.data + 8 * 8))
    ctx.copier_wait(t)
    want = w.copy()
    for b in range(B):
        want[offs[b]:offs[b + 1]] = w[offs[b]:offs[b + 1]] / np.sqrt(nu2[b])
    assert np.array_equal(unit[:, 1], want)               # the same IEEE division
    if not in_place:
        assert np.array_equal(raw[:, 1], w)
    assert np.isnan(raw[:, 0]).all()


def test_large_copies_take_the_direct_path():
    ctx = _ctx()
    p = 700_000                                            # 5.6 MB: above the bounce size
    src = np.arange(p, dtype=np.float64)
    dev = ctx.from_host(src)
    out = np.empty(p)
    small = np.empty(10)
    t = ctx.copier_submit([(small, dev.ptr + 8 * 5), (out, dev.ptr), (small, dev.ptr + 8 * 100)])
    ctx.copier_wait(t)
    assert np.array_equal(out, src) and np.array_equal(small, src[100:110])


@pytest.mark.parametrize("in_place", [False, True])
@pytest.mark.parametrize("position", [0, 1, 2])
def test_block_normalisation_of_a_large_column(position, in_place):
    """columns above the bounce size arrive straight in the caller's pages; their normalisation is shared out to the
    helper threads -- beside the copy of the job's next large column when there is one (position 0, 1), after the last
    copy otherwise (position 2) -- and must be the same IEEE division, with the other columns untouched"""
    ctx = _ctx()
    p, offs = 900_001, [0, 1, 300_000, 300_007, 900_001]                      # 7.2 MB columns, ragged blocks
    rng = np.random.default_rng(position + 10 * in_place)
    B = len(offs) - 1
    src = np.asfortranarray(rng.standard_normal((p, 3)))
    dev = ctx.from_host(src.ravel(order="F"))
    nu2 = np.ascontiguousarray(rng.uniform(0.5, 4.0, B))                       # (plain host memory serves as well)
    for rep in range(3):
        out = np.full((p, 3), np.nan, order="F")
        unit = out if in_place else np.full((p, 3), np.nan, order="F")
        small = np.empty(7)
        copies = [(out[:, k], dev.ptr + 8 * p * k) for k in range(3)]
        copies.insert(1, (small, dev.ptr + 8 * 11))                            # a bounced copy between the large ones
        t = ctx.copier_submit(copies, scale=(out[:, position], unit[:, position], np.asarray(offs, dtype=np.int64),
                                             nu2.ctypes.data))
        ctx.copier_wait(t)
        want = src[:, position].copy()
        for b in range(B):
            want[offs[b]:offs[b + 1]] = src[offs[b]:offs[b + 1], position] / np.sqrt(nu2[b])
        assert np.array_equal(unit[:, position], want)
        assert np.array_equal(small, src[11:18, 0])
        for k in range(3):
            if k != position or not in_place:
                assert np.array_equal(out[:, k], src[:, k])
            if k != position and not in_place:
                assert np.isnan(unit[:, k]).all()


@pytest.mark.parametrize("method,q", [("NIPALS", 1), ("UNIPALS", 1), ("SIMPLS", 1), ("NIPALS", 2), ("UNIPALS", 3),
                                      ("SIMPLS", 2), ("KERNEL", 1)])
def test_few_samples_and_columns_above_the_bounce_size(method, q):
    """n = 24, p = 600 000 in three ragged blocks: every attribute column is a large copy (4.8 MB), a pass is shorter
    than the copies of its columns -- the resident queue's jobs, the several-response routes' send_cols with the block
    normalisation, SIMPLS' columns and device-formed beta_ -- against the oracle"""
    import warnings

    from mbpls_amd import MBPLS
    from oracle import mbpls_oracle as orc
    from tests import util

    rng = np.random.default_rng(77 + q)
    n, sizes, K = 24, [1, 250_007, 349_992], 3
    T = rng.standard_normal((n, 4))
    blocks = [T.dot(rng.standard_normal((4, w))) + 0.5 * rng.standard_normal((n, w)) + 3.0 for w in sizes]
    Y = T[:, :q] + 0.1 * rng.standard_normal((n, q))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = util.model_to_flat(orc.fit(blocks, Y, n_components=K, method=method))
        for rep in range(2):                                  # (the second fit takes pooled host buffers)
            m = MBPLS(n_components=K, method=method).fit(blocks, Y)
            assert util.compare(util.model_to_flat(m), ref, 1e-8, what=f"{method} q={q} fit {rep}") < 1e-8
            if method == "KERNEL":      # (from 2^18 columns on whatever n: the route that streams its columns)
                assert "maintained thin products" in m.fit_plan_["solver"], m.fit_plan_


def test_a_job_can_wait_for_a_value_in_host_mapped_memory_instead_of_an_event():
    """``flag``: no event on the stream; the job starts once the double at the address reaches the value (in a fit a
    step kernel queued behind the producers stores it, ``mbpls_nipals_step.host_seq``; here the host does)."""
    import threading
    import time
    ctx = _ctx()
    src = np.arange(1000.0)
    dev = ctx.from_host(src)
    ctx.sync()
    flags, _ = ctx.mapped_scalars(64)
    flags[40] = 0.0
    out = np.full(1000, np.nan)
    t = ctx.copier_submit([(out, dev.ptr)], flag=(flags.ctypes.data + 8 * 40, 3.0))
    later = ctx.copier_submit([(np.empty(4), dev.ptr)])          # queued behind it (jobs run in order)
    time.sleep(0.05)
    assert np.isnan(out).all()                                     # still waiting
    flags[40] = 2.0
    time.sleep(0.02)
    assert np.isnan(out).all()
    threading.Timer(0.02, lambda: flags.__setitem__(40, 3.0)).start()
    ctx.copier_wait(later)
    assert np.array_equal(out, src)
    assert t < later


def test_destroying_the_context_releases_a_job_that_waits_for_its_flag():
    """a job whose flag is never stored must not keep mbpls_hip_destroy from returning"""
    import time
    from mbpls_amd import _capi
    ctx = _capi.Context(0)
    src = np.arange(64.0)
    dev = ctx.from_host(src)
    ctx.sync()
    flags, _ = ctx.mapped_scalars(16)
    flags[3] = 0.0
    out = np.full(64, np.nan)
    ctx.copier_submit([(out, dev.ptr)], flag=(flags.ctypes.data + 8 * 3, 1.0))
    time.sleep(0.02)
    t = time.perf_counter()
    dev.free()
    ctx.close()
    assert time.perf_counter() - t < 2.0
    assert np.isnan(out).all()


@pytest.mark.parametrize("n,p,blocks,K,method", [(4096, 60000, 3, 6, "NIPALS"), (300, 5000, 2, 6, "NIPALS"),
                                                 (1000, 40000, 4, 8, "UNIPALS")])
def test_repeated_fits_are_bit_identical(n, p, blocks, K, method):
    """Every attribute of a device-resident fit travels through the copier thread (pieces of a column moved by several
    threads, the block normalisation fused, jobs started by events or by a step kernel's flag): forty fits of the same
    data must agree bit for bit."""
    import hashlib
    import mbpls_amd
    from mbpls_amd import MBPLS, DeviceMatrix
    ctx = _ctx()
    rng = np.random.default_rng(n)
    T = rng.standard_normal((n, 8))
    Y = T[:, :1] + 0.3 * rng.standard_normal((n, 1))
    offs = [int(round(b * p / blocks)) for b in range(blocks + 1)]
    X = DeviceMatrix(ctx, n, p, offs)
    X.fill_latent(ctx.from_host(T), 8, 0.5, 99, 0)
    ctx.sync()

    def digest(m):
        h = hashlib.sha1()
        for a in [m.R_, m.beta_, m.Ts_, m.U_, m.V_] + list(m.W_) + list(m.P_) + list(m.T_):
            h.update(np.ascontiguousarray(a).tobytes())
        return h.hexdigest()

    first = digest(MBPLS(n_components=K, method=method).fit(X, Y))
    assert all(digest(MBPLS(n_components=K, method=method).fit(X, Y)) == first for _ in range(40))
    X.close()


@pytest.mark.parametrize("method,blocks,K", [("NIPALS", 3, 5), ("UNIPALS", 1, 4), ("NIPALS", 2, 1)])
def test_small_fits_copy_their_attributes_whole(monkeypatch, method, blocks, K):
    """Small problems do not stream their columns out one by one (a copy costs the copier ~30 us whatever its size: 36
    copies kept a 0.45 ms queue waiting 1.1 ms): engine._ResidentQueue collects them and submits ONE job of merged
    copies behind the queue, the block normalisation of the weights follows on the host.  Same attributes, bit for
    bit, as the streamed form (BULK_BYTES = 0), on both routes of one-response NIPALS."""
    from mbpls_amd import MBPLS
    from mbpls_amd import engine
    from oracle import mbpls_oracle as orc
    from tests import util
    n, p = 150, 3000
    X, Y = orc.make_latent(n, p, 1, r=6, seed=17 + K)
    cuts = [p * (b + 1) // blocks for b in range(blocks - 1)]
    Xs = [np.ascontiguousarray(b) for b in np.split(X, cuts, axis=1)] if blocks > 1 else X
    submitted = []
    real = _ctx().copier_submit.__func__

    def spy(self, copies, **kw):
        submitted.append(len(copies))
        return real(self, copies, **kw)

    monkeypatch.setattr(type(_ctx()), "copier_submit", spy)
    for q1 in ("1", "0"):
        monkeypatch.setenv("MBPLS_NIPALS_Q1_FIXED_POINT", q1)
        submitted.clear()
        bulk = MBPLS(n_components=K, method=method).fit(Xs, Y)
        jobs_bulk = list(submitted)
        monkeypatch.setattr(engine._ResidentQueue, "BULK_BYTES", 0)
        submitted.clear()
        streamed = MBPLS(n_components=K, method=method).fit(Xs, Y)
        jobs_streamed = list(submitted)
        monkeypatch.undo()
        monkeypatch.setattr(type(_ctx()), "copier_submit", spy)
        assert len(jobs_bulk) <= 3 < len(jobs_streamed) or K == 1, (jobs_bulk, jobs_streamed)
        a, b = util.model_to_flat(bulk), util.model_to_flat(streamed)
        assert set(a) == set(b)
        for k in a:
            assert np.array_equal(a[k], b[k]), k
        ref = orc.fit(Xs, Y, n_components=K, method=method)
        assert util.compare(a, util.model_to_flat(ref), 1e-6, what="bulk copies") < 1e-9



def test_upload_matrices_of_small_fits_are_kept_and_overwritten(monkeypatch):
    """Fits from host arrays of a small shape reuse the device matrix of the previous fit of that shape
    (engine.HipEngine.load_host_blocks / close: seven allocations and uploads less per fit).  What the previous fit
    left in it -- another data set, standardised and deflated in place by a storing route -- must not show: same
    attributes, bit for bit, as a fit into a fresh matrix; the four latest shapes are kept, `trim_pool` returns them."""
    from mbpls_amd import MBPLS
    from oracle import mbpls_oracle as orc
    from tests import util
    ctx = _ctx()
    ctx.trim_pool()
    shapes = [(40, 200, [100]), (60, 333, []), (25, 1000, [10, 500]), (150, 300, [7]), (33, 64, [])]

    def data(i, seed):
        n, p, cuts = shapes[i]
        X, Y = orc.make_latent(n, p, 1, r=4, seed=seed)
        return ([np.ascontiguousarray(b) for b in np.split(X, cuts, axis=1)] if cuts else X), Y

    def flat(i, seed, method="NIPALS"):
        X, Y = data(i, seed)
        return util.model_to_flat(MBPLS(n_components=3, method=method).fit(X, Y))

    fresh = flat(0, 1)                                   # into a new matrix (nothing kept yet)
    cache = ctx._mat_cache
    assert len(cache) == 1
    kept = next(iter(cache.values()))
    monkeypatch.setenv("MBPLS_DEFLATION", "explicit")
    monkeypatch.setenv("MBPLS_IMPLICIT_STD", "0")
    flat(0, 2)                                           # another data set, stored standardised, deflated in place
    monkeypatch.undo()
    assert next(iter(cache.values())) is kept            # (the same matrix object went out and came back)
    again = flat(0, 1)
    assert set(again) == set(fresh)
    for k in fresh:
        assert np.array_equal(again[k], fresh[k]), k
    ref = orc.fit(*data(0, 1), n_components=3)
    assert util.compare(again, util.model_to_flat(ref), 1e-6, what="kept upload matrix") < 1e-9
    for i in range(1, 5):                                # five shapes: the oldest one is dropped
        flat(i, 10 + i, method=["UNIPALS", "KERNEL", "SIMPLS", "NIPALS"][i - 1])
    assert len(cache) == 4 and (40, (0, 100, 200)) not in cache, list(cache)
    big = orc.make_latent(40, 200_000, 1, r=4, seed=3)   # 64 MB: beyond engine.MAT_CACHE_BYTES, not kept
    MBPLS(n_components=2).fit(*big)
    assert len(cache) == 4 and all(m.nbytes <= 32 << 20 for m in cache.values())
    ctx.trim_pool()
    assert not ctx._mat_cache and cache is not ctx._mat_cache
    final = flat(0, 1)
    for k in fresh:
        assert np.array_equal(final[k], fresh[k]), k


def test_attributes_of_a_model_that_was_let_go_are_reused_and_kept_ones_are_not():
    """The p x K attributes are views of host buffers the context hands out again once nobody refers to them
    (engine._pooled_host_array): a loop of fits at BASELINE size pays neither munmap nor page faults (20-26 ms of a 77 ms
    iteration).  What the caller keeps of an old model -- a slice is enough -- stays untouched."""
    import mbpls_amd
    from mbpls_amd import MBPLS, DeviceMatrix
    from bench import planted
    ctx = _ctx()
    ctx.trim_pool()
    n, p, K = 512, 300_000, 4                        # attributes of 9.6 MB
    Ts, Y = planted(n, 1, 8, 0)
    X = DeviceMatrix(ctx, n, p, [0, p // 2, p])
    X.fill_latent(ctx.from_host(Ts), 8, 0.5, 7, 0)
    m1 = MBPLS(n_components=K).fit(X, Y)
    kept = m1.W_[1][:, 2]                            # a column of one block of an attribute
    kept_copy, kept_addr = kept.copy(), kept.ctypes.data
    addrs1 = {m1.R_.ctypes.data, m1.P_[0].ctypes.data}
    ref_beta = m1.beta_.copy()
    del m1
    m2 = MBPLS(n_components=K).fit(X, Y)
    addrs2 = {m2.R_.ctypes.data, m2.P_[0].ctypes.data, m2.W_[0].ctypes.data}
    assert addrs1 & addrs2, (addrs1, addrs2)         # buffers of the model that was let go serve the new one
    lo = m2.W_[0].ctypes.data
    assert not (lo <= kept_addr < lo + 8 * p * K) or m2.W_[0].base is not kept.base
    assert np.array_equal(kept, kept_copy)           # ... the kept column's buffer was not among them
    assert np.array_equal(m2.beta_, ref_beta)
    X.close()
    ctx.trim_pool()
    assert not ctx._host_pool
