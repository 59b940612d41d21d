"""GPU parity for LONG columns (n beyond the register-resident column passes): csrc/colpass_cluster.hip -- clusters of
workgroups that share a column, ONE read of X per pass (the never-writes-X forms) --, csrc/colpass_long.hip -- the
two-phase passes in their storing and their never-writes-X forms --, the multi-workgroup step kernel of csrc/resident.hip.

Tall data -- many samples, few features -- is what the reference's n >= p branches exist for
(mbpls.py:388-479 UNIPALS, :580-692 KERNEL); NIPALS / SIMPLS take the same passes.  Checked (a) with the
streamed-column kernels forced on small problems (every fixture case of the n >= p and p > n matrices,
row-subset fits) and (b) at n = 100 000 against the oracle.  Tolerance 1e-6 relative, sign-aligned."""
import numpy as np
import pytest

from oracle import mbpls_oracle as orc
from tests import util
from tests.test_gpu_parity import _fit_flat

pytestmark = pytest.mark.gpu
TOL = 1e-6


def _long_mode(ctx, mode):
    """"cluster": one-read passes with 128-row slabs (several workgroups per column even on small problems);
    "two_phase": the streamed passes of colpass_long.hip; None: the defaults"""
    ctx.set_tunable("cluster_passes", 0 if mode == "two_phase" else 1)
    ctx.set_tunable("cluster_slab_rows", 128 if mode in ("cluster", "cluster_shared") else 8192)
    # "cluster": a dedicated exchange wave per workgroup; "cluster_shared": wave 0 computes and exchanges; None: by shape
    ctx.set_tunable("cluster_exchange", {"cluster": 1, "cluster_shared": 2}.get(mode, 0))


@pytest.fixture(params=["cluster", "two_phase"])
def forced_long(gpu_required, request):
    import mbpls_amd
    ctx = mbpls_amd.get_context()
    ctx.set_tunable("long_columns_from", 0)
    _long_mode(ctx, request.param)
    yield ctx
    ctx.set_tunable("long_columns_from", 8192)
    _long_mode(ctx, None)
    assert ctx.cluster_status() == 0


def _oracle_flat(ref, Xn, Yn, method):
    flat = util.model_to_flat(ref)
    flat["predict"] = orc.predict(ref, Xn)
    if method != "SIMPLS":
        Ts, T, U = orc.transform(ref, Xn, Yn, return_block_scores=True)
        for b, blk in enumerate(T):
            flat[f"T_new/{b}"] = blk
    else:
        Ts, U = orc.transform(ref, Xn, Yn)
    flat["Ts_new"], flat["U_new"] = Ts, U
    return flat


@pytest.mark.parametrize("fixture", ["matrix_pgtn", "matrix_ngtp"])
def test_fixture_matrix_with_streamed_columns(forced_long, fixture):
    from mbpls_amd import MBPLS
    g = util.load(fixture)
    splits = np.cumsum(g["splits"])[:-1]
    worst = 0.0
    for q in (1, 3):
        X, Y = g[f"in/q{q}/X"], g[f"in/q{q}/Y"]
        Xn, Yn = g[f"in/q{q}/Xnew"], g[f"in/q{q}/Ynew"]
        for blocks in ("single", "multi"):
            sp = (lambda A: A) if blocks == "single" else (
                lambda A: [np.ascontiguousarray(b) for b in np.split(A, splits, axis=1)])
            for m in ("NIPALS", "UNIPALS", "KERNEL", "SIMPLS"):
                for std in (True, False):
                    pre = f"ref/q{q}/{blocks}/{m}/std{int(std)}/ca1"
                    ref = util.sub(g, pre)
                    mdl = MBPLS(n_components=3, method=m, standardize=std, full_svd=(m == "SIMPLS")).fit(sp(X), Y)
                    has_T_new = any(k.startswith("T_new") for k in ref)
                    worst = max(worst, util.compare(_fit_flat(mdl, sp(Xn), Yn, block_scores=has_T_new), ref, TOL,
                                                    what=pre))
    assert worst < 1e-8, worst


@pytest.mark.parametrize("method,n,p,q,K,blocks,mode", [
    ("NIPALS", 1001, 3000, 2, 3, [1000, 2000], "cluster"),        # 8 workgroups per column
    ("NIPALS", 1001, 3000, 2, 3, [1000, 2000], "two_phase"),
    ("UNIPALS", 777, 2500, 1, 3, [2500], "cluster"),
    ("UNIPALS", 777, 2500, 1, 3, [2500], "two_phase"),
    ("NIPALS", 1500, 1300, 1, 4, [300, 400, 600], "cluster"),     # one response: lazy deflation, three blocks
    ("NIPALS", 1500, 1300, 1, 4, [300, 400, 600], "cluster_shared"),
    ("UNIPALS", 999, 1100, 3, 3, [500, 600], "cluster_shared"),
    ("SIMPLS", 1203, 700, 1, 3, [700], "cluster"),
    ("NIPALS", 100_000, 300, 1, 3, [100, 200], None),       # genuinely long columns (n > 65536), n >= p
    ("NIPALS", 100_000, 300, 1, 3, [100, 200], "two_phase"),
    ("NIPALS", 12_000, 260, 3, 2, [100, 160], None),         # several responses: the inner loop in q x q algebra
    ("UNIPALS", 100_000, 240, 2, 3, [240], None),
    ("KERNEL", 70_001, 200, 1, 3, [80, 120], None),
    ("SIMPLS", 90_000, 150, 2, 3, [150], None),
    ("NIPALS", 20_000, 1000, 1, 3, [1000], None),            # the paper protocol's shape: three workgroups per column
    ("UNIPALS", 16_384, 2100, 1, 3, [700, 1400], None),      # two full slabs
])
def test_vs_oracle(gpu_required, method, n, p, q, K, blocks, mode):
    import mbpls_amd
    from mbpls_amd import MBPLS
    ctx = mbpls_amd.get_context()
    ctx.set_tunable("long_columns_from", 0 if n <= 8192 else 8192)
    _long_mode(ctx, mode)
    try:
        X, Y = orc.make_latent(n, p, q, seed=n + p)
        Xn, Yn = orc.make_latent(64, p, q, seed=1)
        cuts = np.cumsum(blocks)[:-1]
        sp = (lambda A: A) if len(blocks) == 1 else (
            lambda A: [np.ascontiguousarray(b) for b in np.split(A, cuts, axis=1)])
        ref = orc.fit(sp(X), Y, n_components=K, method=method, full_svd=True)
        mdl = MBPLS(n_components=K, method=method, full_svd=True).fit(sp(X), Y)
        w = util.compare(_fit_flat(mdl, sp(Xn), Yn), _oracle_flat(ref, sp(Xn), Yn, method), TOL,
                         what=f"{method} n={n} p={p}")
        assert w < 1e-8, w
        assert ctx.cluster_status() == 0
        if mode != "two_phase":
            assert "clusters of workgroups" in mdl.fit_plan_["kernels"], mdl.fit_plan_
    finally:
        ctx.set_tunable("long_columns_from", 8192)
        _long_mode(ctx, None)


def test_row_subset_fit_with_streamed_columns(forced_long):
    """cross-validation folds (row mask in the standardise pass) through the streamed-column kernels"""
    from mbpls_amd import MBPLS, DeviceMatrix
    X, Y = orc.make_latent(301, 900, 1, seed=11)
    mask = np.ones(301, dtype=bool)
    mask[::7] = False
    dX = DeviceMatrix(forced_long, 301, 900, [0, 400, 900])
    dX.upload(X, 0)
    for method in ("NIPALS", "UNIPALS"):
        mdl = MBPLS(n_components=3, method=method)._fit(dX, Y, mask)
        ref = orc.fit([X[mask][:, :400], X[mask][:, 400:]], Y[mask], n_components=3, method=method)
        w = util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what=method)
        assert w < 1e-9, w
    dX.close()


def test_register_resident_configs_above_8192(gpu_required):
    """long_columns_from = 65536 keeps the register-resident passes (R = 32 / 64 rows per thread, spilling;
    also what lazy deflation needs) for 8192 < n <= 65536 -- slower than the streamed passes, same numbers."""
    import mbpls_amd
    from mbpls_amd import MBPLS
    ctx = mbpls_amd.get_context()
    X, Y = orc.make_latent(10_000, 2000, 1, seed=12)
    a = MBPLS(n_components=3).fit(X, Y)                      # clusters of two workgroups per column (never write X)
    ctx.set_tunable("long_columns_from", 65536)
    try:
        b = MBPLS(n_components=3).fit(X, Y)                  # register-resident, lazy deflation
    finally:
        ctx.set_tunable("long_columns_from", 8192)
    ctx.set_tunable("cluster_passes", 0)
    try:
        c = MBPLS(n_components=3).fit(X, Y)                  # two-phase long-column passes
    finally:
        ctx.set_tunable("cluster_passes", 1)
    ref = orc.fit(X, Y, n_components=3)
    for mdl in (a, b, c):
        w = util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what="n=10000")
        assert w < 1e-8, w


@pytest.mark.parametrize("method,norm,offset", [
    ("NIPALS", 2, 0.0),
    ("NIPALS", np.inf, 0.0),       # the max-norm stop rule reduces over the step kernel's workgroups as well
    ("NIPALS", 2, 300.0),          # offset data: matrix-core dots and row pass on x - mean (centring flavour)
    ("UNIPALS", 2, 0.0),           # tall one-response UNIPALS: the one-pass-per-component queue
    ("UNIPALS", 2, 300.0),
])
def test_tall_one_response_fit_is_one_queue_and_never_writes_x(gpu_required, monkeypatch, q1_route, method, norm, offset):
    """n = 70 000 (> 2 x 32768: the step kernel runs on several workgroups with grid-wide reductions), two blocks,
    one response: statistics-only first pass, lazy deflation, everything queued on the device -- against the oracle,
    with the caller's resident matrix bit-identical afterwards and the host-driven loop never entered."""
    import mbpls_amd
    from mbpls_amd import MBPLS, DeviceMatrix
    from mbpls_amd.engine import HipEngine
    n, p, K = 70_000, 260, 3
    X, Y = orc.make_latent(n, p, 1, seed=5)
    X = X + offset * (1.0 + np.arange(p) % 3)
    blocks = [X[:, :100], X[:, 100:]]
    ref = orc.fit(blocks, Y, n_components=K, method=method, nipals_convergence_norm=norm)
    ctx = mbpls_amd.get_context()
    Xd = DeviceMatrix(ctx, n, p, [0, 100, p])
    Xd.upload(X, 0)
    host_passes, flavours = [], []
    real_sweep, real_close = HipEngine.sweep, HipEngine.close

    def spy_sweep(self, *a, **k):
        host_passes.append("sweep")
        return real_sweep(self, *a, **k)

    def spy_close(self):
        flavours.append((self._istd, self._cen, self._lazy_pslot))
        return real_close(self)

    monkeypatch.setattr(HipEngine, "sweep", spy_sweep)
    monkeypatch.setattr(HipEngine, "close", spy_close)
    mdl = MBPLS(n_components=K, method=method, nipals_convergence_norm=norm).fit(Xd, Y)
    assert host_passes == [], host_passes                       # no host-driven pass: the queue held
    assert flavours == [(True, offset > 0.0, "P")], flavours    # implicit standardisation (+ centring), lazy deflation
    assert np.array_equal(Xd.download(), X)                     # never written
    if method == "NIPALS":
        util.check_nipals_iterations(mdl, ref, K)
    w = util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what=f"{method} norm={norm} offset={offset}")
    assert w < (1e-8 if offset == 0.0 else 1e-6), w
    Xd.close()


def test_many_lazy_corrections_through_the_finishing_kernel(forced_long):
    """kprev up to 29 in k_fin_lazy (the corrections sum_i P[j,i] cu_i / ce_i and the d sums of the two-phase passes)"""
    from mbpls_amd import MBPLS
    rng = np.random.default_rng(4)
    n, p, K = 500, 900, 30
    X = rng.standard_normal((n, p))
    Y = X[:, :6].sum(axis=1, keepdims=True) + 0.1 * rng.standard_normal((n, 1))
    # (UNIPALS: no iteration -- NIPALS on deep components of noise never meets its stop rule, here or in the reference)
    ref = orc.fit(X, Y, n_components=K, method="UNIPALS")
    mdl = MBPLS(n_components=K, method="UNIPALS").fit(X, Y)
    assert util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what=f"UNIPALS K={K}") < 1e-6


@pytest.mark.parametrize("method,q,route", [
    ("NIPALS", 1, "device-resident queue"),      # the time-out reaches the host when the queue returns
    ("NIPALS", 2, "host-driven loop"),           # ... behind the first pass of the loop
    ("UNIPALS", 1, "device-resident queue"),
])
def test_cluster_handoff_timeout_reruns_the_fit_on_two_phase_passes(gpu_required, method, q, route):
    """A hand-off between the workgroups of a column cluster that gives up (another tenant keeps partners off the
    device; forced here with a spin limit of one poll) must not reach the caller as garbage nor be iterated on: the
    fit warns, switches the context to the two-phase passes and runs again -- same numbers as the oracle."""
    import mbpls_amd
    from mbpls_amd import MBPLS
    ctx = mbpls_amd.get_context()
    n, p, K = 2100, 900, 3
    X, Y = orc.make_latent(n, p, q, seed=21)
    ref = orc.fit([X[:, :400], X[:, 400:]], Y, n_components=K, method=method, full_svd=(method == "SIMPLS"))
    ctx.set_tunable("long_columns_from", 0)
    _long_mode(ctx, "cluster")
    ctx.set_tunable("cluster_spin_limit", 1)
    try:
        assert ctx.cluster_columns(n)
        with pytest.warns(RuntimeWarning, match="two-phase long-column passes"):
            mdl = MBPLS(n_components=K, method=method, full_svd=(method == "SIMPLS")).fit([X[:, :400], X[:, 400:]], Y)
        assert ctx.cluster_passes == 0 and not ctx.cluster_columns(n)
        assert "two streaming phases" in mdl.fit_plan_["kernels"], mdl.fit_plan_
        assert ctx.cluster_status() == 0                       # the flag was consumed by the fit
        w = util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what=f"{method} q={q} after a time-out")
        assert w < 1e-8, w
    finally:
        ctx.set_tunable("cluster_spin_limit", 1 << 22)
        ctx.set_tunable("long_columns_from", 8192)
        _long_mode(ctx, None)
        ctx.cluster_status()


def _two_phase_fits_do_not_stall(ctx):
    """Regression guard for the fallback route itself.  Round 5 measured a tall NIPALS fit on the two-phase passes
    (n = 20 000, p = 1 000) at 60 ms -- two `deflate_sweep` composites of 29 ms each, 0.16 ms of kernels in them -- where
    UNIPALS beside it took 1.7 ms: host-to-device copies of a few hundred bytes out of PAGEABLE memory inside the pass
    (the small matrices of the device helpers, the block offsets of the matrix-core X^T V) are staged by the runtime and
    now and then executed tens of milliseconds late.  They go through the context's pinned ring now (csrc/small.hip)
    or are not made at all (csrc/xtv_mfma.hip); the timeline of such a fit is profiles/r06_two_phase_*.  Here: eight
    fits on the two-phase passes, none may take more than 20 ms, no pass more than 2 ms -- fits during which the box's
    cgroup ran out of CPU quota (util.cpu_throttled_periods) are not counted: those stand still wherever they are."""
    import time
    from mbpls_amd import MBPLS, DeviceMatrix
    from bench import planted
    n, p = 20000, 1000
    Ts, Y = planted(n, 1, 32, 0)
    X = DeviceMatrix(ctx, n, p)
    dT = ctx.from_host(Ts)
    X.fill_latent(dT, 32, 0.5, 1234, 0)
    ctx.sync()
    dT.free()
    try:
        walls, stalled = [], 0
        for i in range(40):
            thr0 = util.cpu_throttled_periods()
            ctx.profile(True); ctx.profile_reset()
            t0 = time.perf_counter()
            m = MBPLS(n_components=3).fit(X, Y)
            ctx.sync()
            wall = time.perf_counter() - t0
            prof = ctx.profile_get(); ctx.profile(False)
            assert "two streaming phases" in m.fit_plan_["kernels"], m.fit_plan_
            if i < 2:                                               # (the first fits of a shape build scratch and pools)
                continue
            if thr0 >= 0 and util.cpu_throttled_periods() != thr0:  # the box took the CPU away during this fit
                stalled += 1
                continue
            walls.append(wall)
            worst = max(prof, key=lambda k: prof[k]["avg_ms"])
            assert prof[worst]["avg_ms"] < 2.0, (i, worst, prof[worst], m.fit_timings_, walls)
            assert m.fit_timings_["total"] < 0.020, (i, m.fit_timings_, walls)
            if len(walls) == 8:
                break
        assert len(walls) >= 4, (walls, stalled)
        assert sorted(walls)[len(walls) // 2] < 0.010, walls
    finally:
        X.close()


def test_refused_cooperative_launch_falls_through_to_two_phase_passes(gpu_required):
    """The cluster passes are launched cooperatively with a grid sized from the runtime's occupancy figure.  A grid the
    device cannot hold at once (forced: capacity set to 4096 workgroups) is refused by the runtime; the pass then runs
    on the two-phase kernels in the same call and the context stays there."""
    import mbpls_amd
    from mbpls_amd import MBPLS
    ctx = mbpls_amd.get_context()
    n, p, K = 1500, 5000, 3
    X, Y = orc.make_latent(n, p, 1, seed=22)
    ref = orc.fit(X, Y, n_components=K)
    ctx.set_tunable("long_columns_from", 0)
    _long_mode(ctx, "cluster")
    ctx.set_tunable("cluster_cap", 4096)
    try:
        assert ctx.cluster_columns(n)
        mdl = MBPLS(n_components=K).fit(X, Y)
        assert not ctx.cluster_columns(n)                      # capacity 0 after the refusal
        w = util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what="refused cooperative launch")
        assert w < 1e-8, w
        assert "cannot hold a cluster" in mdl.fit_plan_["cluster_passes_off"], mdl.fit_plan_
        _two_phase_fits_do_not_stall(ctx)
        ctx.set_tunable("cluster_cap", -1)                     # ask the runtime again
        assert ctx.cluster_columns(n)
        again = MBPLS(n_components=K).fit(X, Y)
        assert "clusters of workgroups" in again.fit_plan_["kernels"]
        assert util.compare(util.model_to_flat(again), util.model_to_flat(ref), TOL, what="cluster passes back") < 1e-8
    finally:
        ctx.set_tunable("cluster_cap", -1)
        ctx.set_tunable("long_columns_from", 8192)
        _long_mode(ctx, None)


def test_cluster_timeout_under_column_shards_is_a_collective_decision(gpu_required):
    """round-5 advisor (medium): a hand-off time-out on ONE rank must not send that rank alone into the re-run of the fit
    -- see tests/_cluster_timeout_worker.py"""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "_cluster_timeout_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("re-ran the fit on both, == oracle") == 2, r.stdout
