"""Shared helpers for the parity tests: fixture loading and sign-aligned comparison."""
import contextlib
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# attributes whose column k flips with the sign of component k
SIGNED = ("Ts_", "U_", "V_", "R_", "T_", "W_", "P_", "W_non_normal_", "W_concat_", "Ts_new", "T_new", "U_new")
UNSIGNED = ("beta_", "A_", "A_corrected_", "explained_var_x_", "explained_var_y_",
            "explained_var_xblocks_", "predict")


def load(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def sub(store, prefix):
    """All entries below ``prefix/`` with the prefix stripped."""
    pre = prefix.rstrip("/") + "/"
    return {k[len(pre):]: v for k, v in store.items() if k.startswith(pre)}


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if b.size == 0:
        return 0.0
    den = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))


def model_to_flat(m, n_blocks=None):
    """Flatten a fitted estimator / OracleModel into the fixture's key scheme."""
    get = (lambda k: m[k]) if isinstance(m, dict) else (lambda k: getattr(m, k))
    has = (lambda k: k in m) if isinstance(m, dict) else (lambda k: hasattr(m, k))
    out = {}
    for a in ("Ts_", "U_", "V_", "R_", "beta_", "A_", "A_corrected_", "explained_var_x_",
              "explained_var_y_", "explained_var_xblocks_", "W_concat_"):
        if has(a):
            out[a] = np.asarray(get(a), dtype=np.float64)
    for a in ("T_", "W_", "P_", "W_non_normal_"):
        if has(a):
            v = get(a)
            if isinstance(v, list):
                for b, blk in enumerate(v):
                    out[f"{a}/{b}"] = np.asarray(blk)
            else:
                out[a] = np.asarray(v)
    return out


def compare(got, ref, tol, skip=(), only=None, what=""):
    """Sign-aligned comparison of two flat attribute dicts.  Returns worst rel. error.

    Every key present in ``ref`` (minus ``skip``) must be present in ``got``."""
    s = np.sign(np.sum(ref["Ts_"] * got["Ts_"], axis=0))
    s[s == 0] = 1.0
    worst = 0.0
    for k, rv in ref.items():
        base = k.split("/")[0]
        if base in skip or k in skip:
            continue
        if only is not None and base not in only:
            continue
        if base in ("x_mean", "x_scale", "y_mean", "y_scale"):
            continue
        if np.asarray(rv).size == 0 and k not in got:
            continue  # the reference leaves empty placeholders (e.g. W_concat_ outside KERNEL)
        assert k in got, f"{what}: missing {k}"
        gv = np.asarray(got[k], dtype=np.float64)
        rv = np.asarray(rv, dtype=np.float64)
        if gv.ndim == 1 and rv.ndim == 1:
            pass
        if base in SIGNED and gv.ndim == 2 and gv.shape[1] == s.size:
            gv = gv * s
        e = relerr(gv.reshape(rv.shape), rv)
        assert e <= tol, f"{what}: {k} rel err {e:.3e} > {tol:.1e}"
        worst = max(worst, e)
    return worst


def check_nipals_iterations(mdl, ref, K=None, two=True):
    """Iteration counts of a NIPALS fit against the oracle's.  The default route of ONE-RESPONSE NIPALS makes one pass per
    component (the reference's second pass only re-derives the first, solvers._fit_nipals_q1_fixed_point) and reports 1;
    the premise -- the reference converges in two -- is checked on the oracle (``two``).  Every other route repeats the
    reference's loop and must stop where the oracle stops."""
    got, want = list(mdl.nipals_iterations_), list(ref.nipals_iterations_)
    plan = getattr(mdl, "fit_plan_", None) or {}
    if "fixed point after one pass" in str(plan.get("solver", "")):
        assert got == [1] * len(want), got
        if two:
            assert want == [2] * len(want), want
    else:
        assert got == want, (got, want)
    if K is not None:
        assert len(got) == K


@contextlib.contextmanager
def forced_routes(**forms):
    """Force fallback forms of the solvers (mbpls_amd.plan.FALLBACK_FORMS: nipals_subspace, nipals_qspace, nipals_gram,
    unipals_nodeflate, r_incremental, stream_stats) for the fits inside the block, e.g. ``nipals_subspace=False``.  These
    are not switches of the product (no environment variable): a fit takes them when its shape needs them; the tests
    compare each with the oracle."""
    from mbpls_amd.plan import FitOptions
    old = dict(FitOptions.forced)
    FitOptions.forced.update(forms)
    try:
        yield
    finally:
        FitOptions.forced.clear()
        FitOptions.forced.update(old)


def cpu_throttled_periods():
    """Periods in which this process's cgroup ran out of CPU quota so far (cgroup v2 `cpu.stat`, else v1; -1 when the
    file is not there).  The GPU boxes of the pool show 256 CPUs under a 16-CPU quota; a BLAS pool sized from the first
    number spins the quota of a 100 ms period away within 12 ms (profiles/r06_stall_throttled.txt; tests/conftest.py caps
    the pool): a wall-clock guard has to leave out the fits a throttled period fell into -- the whole process, copier
    thread included, stands still for up to 75 ms there."""
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            with open(path) as f:
                d = dict(line.split() for line in f)
            return int(d.get("nr_throttled", 0))
        except (OSError, ValueError):
            pass
    return -1
