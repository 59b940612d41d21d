"""``MBPLS`` -- drop-in for ``mbpls.mbpls.MBPLS`` (reference mbpls/mbpls.py:22-1437) whose
``fit`` / ``predict`` / ``transform`` run on an MI355X through ``libmbpls_hip.so``.

Same constructor (verbatim, sklearn ``clone``/``get_params`` depend on it), same fitted
attribute names, types and shapes, same errors.  There is no CPU path: without
the HIP library or a GPU every method raises.

Deliberate deviations from the reference (documented in DESIGN.md):
* ``sparse_data=True`` (NaN-aware NIPALS, per-row/column Python loops in the reference, mbpls.py:846-926)
  runs on masked column passes (csrc/colpass_nan.hip) for missing values in X; missing responses are handled on the
  host (n x q) for EVERY row, where the reference only patches the sparse rows of the last X block and fails on any
  other (mbpls.py:903): same numbers wherever the reference produces any.
* NIPALS has an iteration cap (``solvers.NIPALS_MAX_ITER``) with a warning; the reference loops
  forever on data without latent structure (mbpls.py:841).  With ONE complete response the loop also gives up once
  ``||ts_old - ts||`` STAGNATES above ``max_tol`` (``solvers.q1_noise_floor``: from the fourth iteration on, two
  iterations in a row without halving it; twelve at most) -- always with a warning, ``nipals_converged_[k] = False``:
  the iteration is at its fixed point after the first pass, what is left is rounding noise of a response that is
  exhausted, and the reference does not terminate there either.  A difference that still shrinks is left to converge.
* SIMPLS with one response and ``full_svd=False`` works (the reference raises inside scipy ``svds``).
* X may also be a ``DeviceMatrix`` already resident in HBM (feature-major); ``copy`` then decides
  whether it is deflated in place.
"""
from __future__ import annotations

import warnings

import numpy as np

from . import _capi, runtime, solvers
from .engine import HipEngine
from .scaler import ColumnScaler

try:  # sklearn is what the reference builds on; use its mixins / validators when present
    from sklearn.base import BaseEstimator, MultiOutputMixin, RegressorMixin, TransformerMixin
    from sklearn.exceptions import NotFittedError
    from sklearn.utils import check_array, check_consistent_length
    _HAVE_SKLEARN = True
except Exception:  # pragma: no cover
    _HAVE_SKLEARN = False

    class BaseEstimator:  # minimal stand-ins
        def get_params(self, deep=True):
            import inspect
            return {k: getattr(self, k) for k in inspect.signature(self.__init__).parameters}

        def set_params(self, **kw):
            for k, v in kw.items():
                setattr(self, k, v)
            return self

    class TransformerMixin: pass
    class RegressorMixin: pass
    class MultiOutputMixin: pass

    class NotFittedError(ValueError, AttributeError): pass

    def check_array(a, dtype=np.float64, ensure_2d=True, copy=False, **kw):
        a = np.array(a, dtype=dtype, copy=copy) if copy else np.asarray(a, dtype=dtype)
        if ensure_2d and a.ndim != 2:
            raise ValueError(f"Expected 2D array, got {a.ndim}D array instead")
        if not np.all(np.isfinite(a)):
            raise ValueError("Input contains NaN, infinity or a value too large for dtype('float64').")
        return a

    def check_consistent_length(*arrays):
        ls = {len(a) for a in arrays}
        if len(ls) > 1:
            raise ValueError(f"Found input variables with inconsistent numbers of samples: {sorted(ls)}")

_METHODS = ("UNIPALS", "KERNEL", "NIPALS", "SIMPLS")


def _check_array_no_scan(a):
    """check_array (2-D, numeric, float64, non-empty) without the finite-ness scan."""
    try:
        return check_array(a, dtype=np.float64, copy=False, ensure_all_finite=False)
    except TypeError:      # scikit-learn < 1.6 spells it force_all_finite
        return check_array(a, dtype=np.float64, copy=False, force_all_finite=False)


def _require_finite(out):
    """NaN/inf in new data shows up in every product with it (same error type as sklearn's check_array)."""
    if not np.all(np.isfinite(out)):
        raise ValueError("Input X contains NaN, infinity or a value too large for dtype('float64').")


def _global_block_sizes(eng):
    """block widths of the whole problem (summed over column shards)"""
    return solvers._global_sizes(eng)


def _is_block_list(X):
    return isinstance(X, list) and not isinstance(X[0], list)       # mbpls.py:301


CLUSTER_REARM_FITS = 8


def _rearm_cluster_passes(ctx):
    """A hand-off time-out switches the (process-wide) context to the two-phase long-column passes; that is recorded on
    the context with its reason and undone after CLUSTER_REARM_FITS further fits."""
    down = getattr(ctx, "_cluster_downgrade", None)
    if down is None:
        return
    if getattr(ctx, "cluster_passes", 0):          # (someone switched them on again by hand)
        del ctx._cluster_downgrade
        return
    down["fits_left"] -= 1
    if down["fits_left"] <= 0:
        ctx.set_tunable("cluster_passes", 1)
        del ctx._cluster_downgrade



def _one_at_a_time(method):
    """fit / predict / transform of estimators that share a context take turns (Context.lock): the reference is plain
    numpy and can be driven from several threads at once -- joblib's threading backend does --, the passes of a fit
    share the context's parameter block, workspace and copier"""
    import functools

    @functools.wraps(method)
    def locked(self, *args, **kwargs):
        with runtime.get_context().lock:
            return method(self, *args, **kwargs)
    return locked


class MBPLS(TransformerMixin, RegressorMixin, MultiOutputMixin, BaseEstimator):
    """(Multiblock) PLS regression, MI355X-native.  See the module docstring."""

    def __init__(self, n_components=2, full_svd=False, method='NIPALS', standardize=True, max_tol=1e-14,
                 nipals_convergence_norm=2, calc_all=True, sparse_data=False, copy=True):
        self.n_components = n_components
        self.full_svd = full_svd
        self.method = method
        self.standardize = standardize
        self.max_tol = max_tol
        self.nipals_convergence_norm = nipals_convergence_norm
        self.calc_all = calc_all
        self.sparse_data = sparse_data
        self.copy = copy

    # ------------------------------------------------------------------ input handling
    def _host_blocks(self, X, Y=None):
        """Validate like the reference (mbpls.py:301-323); host arrays are never modified, the
        device upload *is* the copy, so no deepcopy / check_array(copy=True) is needed."""
        raw = X if _is_block_list(X) else [X]
        blocks = []
        for b in raw:
            if Y is not None:
                check_consistent_length(b, Y)                        # :309
            # :310.  The NaN/inf scan of sklearn's check_array is a full single-threaded host pass over X
            # (1.7 s at 32.8 GB); here non-finite input is detected on the GPU instead: it makes the column
            # sums of squares of the first pass non-finite, and the engine raises the same ValueError.
            blocks.append(_check_array_no_scan(b))
        return blocks

    def _engine(self):
        return HipEngine(ctx=runtime.get_context(), comm=runtime.get_comm())

    # ------------------------------------------------------------------ fit
    def fit(self, X, Y):
        """Fit model to given data (reference mbpls.py:273).

        X : 2-D array, list of 2-D arrays (blocks), or a ``DeviceMatrix`` resident in HBM
        Y : 1-D or 2-D array"""
        return self._fit(X, Y, None)

    @_one_at_a_time
    def _fit(self, X, Y, row_mask):
        """fit() on the samples where ``row_mask`` is True (None = all): the engine zeroes the excluded rows
        of its working copy, so the matrix never has to be re-uploaded or sliced for cross-validation folds
        (``mbpls_amd.model_selection``).  n-row attributes are returned for the included samples only."""
        sparse = self.sparse_data is True
        if sparse and self.method != 'NIPALS':                                      # :285-289
            warnings.warn("The parameter sparse data was set to 'True', but the chosen method is not 'NIPALS'."
                          "The method will be set to 'NIPALS'")
            self.method = 'NIPALS'
        if sparse:
            # missing values are allowed in Y as well (:293, force_all_finite=not sparse_data); inf is not
            Y = _check_array_no_scan(np.asarray(Y, dtype=np.float64).reshape(len(Y), -1))
            if np.isinf(Y).any():
                raise ValueError("Input Y contains infinity or a value too large for dtype('float64').")
            Y = np.array(Y, copy=True) if self.copy else Y
        elif (type(Y) is np.ndarray and Y.dtype == np.float64 and Y.ndim in (1, 2) and Y.size
              and np.isfinite(Y).all()):
            # (the common case without sklearn's 0.1 ms of validation machinery -- anything else goes through it,
            # with its errors)
            Y = Y.copy() if self.copy else Y
        else:
            Y = check_array(Y, dtype=np.float64, ensure_2d=False, copy=self.copy)  # :293
        if Y.ndim == 1:
            Y = Y.reshape(-1, 1)                                                    # :297-298
        device_input = isinstance(X, _capi.DeviceMatrix)
        if device_input:
            if X.n != Y.shape[0]:
                raise ValueError(f"Found input variables with inconsistent numbers of samples: [{X.n}, {Y.shape[0]}]")
            blocks = None
        else:
            blocks = self._host_blocks(X, Y)
        if self.method not in _METHODS:
            raise NameError('Method you called is unknown')                         # :1050
        if self.method == 'SIMPLS':
            warnings.warn("Method 'SIMPLS' does not calculate A_ and T_!")          # :996
        if row_mask is not None:
            row_mask = np.asarray(row_mask, dtype=bool).ravel()
            if row_mask.size != Y.shape[0] or not row_mask.any():
                raise ValueError("row_mask must select at least one of the n samples")
            if not device_input or not self.copy:
                raise ValueError("row-subset fits need a DeviceMatrix and copy=True (the resident matrix is kept)")
        if self.standardize:
            self.y_scaler_ = ColumnScaler().fit(Y if row_mask is None else Y[row_mask])   # :325-326
            Ys = self.y_scaler_._standardised(Y)
        else:
            Ys = Y
        if row_mask is not None:
            Ys = np.where(row_mask[:, None], Ys, 0.0)      # excluded samples take no part in any sum
        eng = self._engine()
        eng.sparse = sparse
        eng.rowmask = None if row_mask is None else row_mask.astype(np.float64)
        _rearm_cluster_passes(eng.ctx)
        # A refit replaces the previous p x K attributes; releasing ~0.4 GB of numpy memory (page-table
        # teardown) costs ~15 ms of host time at p = 1e6 -- more than any single pass hides.  The engine drops
        # one stale object per column pass, right after the launch, while the host would only wait anyway.
        stale = [self.__dict__.pop(k) for k in ("P_", "W_", "W_non_normal_", "R_", "beta_", "W_concat_", "T_")
                 if k in self.__dict__]
        old_scalers = self.__dict__.pop("x_scalers_", None)
        if old_scalers:      # their mean_/scale_/var_ are columns of one p x 3 array (engine.stream_stats): recyclable too
            stale.append([a for sc in old_scalers for a in (sc.mean_, sc.scale_, getattr(sc, "var_", sc.mean_))])
            del old_scalers
        eng.deferred_host_work = [stale.pop for _ in stale]
        import time as _time
        t_fit0 = _time.perf_counter()
        threads = runtime.host_threads()
        threads.__enter__()
        try:
            K = self.n_components
            for attempt in (0, 1):
                try:
                    with eng._timed("load"):
                        if device_input:
                            eng.load_device(X, copy=self.copy)
                        else:
                            eng.load_host_blocks(blocks)
                    if self.method == 'NIPALS':
                        res = solvers.fit_nipals(eng, Ys, K, self.standardize, self.max_tol,
                                                 self.nipals_convergence_norm, self.calc_all)
                    elif self.method == 'UNIPALS':
                        res = solvers.fit_unipals(eng, Ys, K, self.standardize)
                    elif self.method == 'KERNEL':
                        res = solvers.fit_kernel(eng, Ys, K, self.standardize, self.calc_all)
                    else:
                        res = solvers.fit_simpls(eng, Ys, K, self.standardize)
                    eng.check_cluster_passes()              # long columns: a hand-off between workgroups gave up -> raise
                    break
                except _capi.ClusterTimeout as err:
                    # Long columns: the workgroups that share a column wait for each other inside the launch, which
                    # needs all of them on the device at once.  The launch is cooperative and sized from the runtime's
                    # occupancy figure, but a second tenant (another rank or process on this GPU, a CU mask) can still
                    # keep partners off the device until the bounded waits give up.  Those passes never write X: the
                    # context takes the two-phase passes from now on and the fit runs again -- unless this fit has
                    # already overwritten the caller's own device matrix (copy=False with a storing pass).
                    wrote_callers = (device_input and not self.copy and
                                     (eng.plan.get("deflation") == "explicit"
                                      or str(eng.plan.get("standardisation", "")).startswith("stored")))
                    if attempt or wrote_callers:
                        raise
                    warnings.warn(f"{err}; this context now uses the two-phase long-column passes (two reads of X per "
                                  "pass) and the fit is run again", RuntimeWarning)
                    eng.ctx.set_tunable("cluster_passes", 0)
                    # (not for good: the context takes the cluster passes again after CLUSTER_REARM_FITS fits -- a
                    # co-tenant that stays costs one failed attempt in so many fits, one that has left costs nothing)
                    eng.ctx._cluster_downgrade = dict(reason=str(err), fits_left=CLUSTER_REARM_FITS)
                    pending = eng.deferred_host_work
                    eng.close()
                    eng = self._engine()
                    eng.sparse, eng.rowmask, eng.deferred_host_work = sparse, None if row_mask is None else row_mask.astype(np.float64), pending
            if row_mask is not None:                        # n-row outputs: included samples only
                for key in ("Ts_", "U_"):
                    if key in res:
                        res[key] = res[key][row_mask]
                if isinstance(res.get("T_"), list):
                    res["T_"] = [t[row_mask] for t in res["T_"]]
            if sparse:
                self._sparsity_info(eng, Y, row_mask)
            with eng._timed("store_attributes"):
                self._store(eng, res)
        finally:
            eng.run_deferred(all_of_it=True)
            threads.__exit__(None, None, None)
            with eng._timed("release"):
                eng.close()
            eng.timings["total"] = _time.perf_counter() - t_fit0
            self.fit_timings_ = dict(eng.timings)   # seconds per host-side phase (not in the reference)
            # the route this fit took (not in the reference): solver form, deflation, standardisation, kernel family,
            # centring of offset data, column shards -- decided in engine / solvers from plan.FitOptions, recorded here
            n_ = getattr(eng, "n", 0)
            long_from = getattr(eng.ctx, "long_columns_from", 8192)
            fam = (("long columns (clusters of workgroups, one read per pass)" if getattr(eng, "cluster_passes", False)
                    else "long columns (two streaming phases)") if n_ > long_from else
                   "few samples (column groups per wave)" if n_ <= 1024 else "register-resident columns")
            self.fit_plan_ = dict(method=self.method, **eng.plan, kernels=fam, centred_in_registers=bool(getattr(eng, "_cen", False)),
                                  column_shards=int(getattr(eng.comm, "world_size", 1)), sparse=bool(sparse))
            if n_ > long_from and not sparse and not getattr(eng, "cluster_passes", False):
                # why a long-column fit is on the two-read passes (half the throughput of the cluster passes)
                down = getattr(eng.ctx, "_cluster_downgrade", None)
                self.fit_plan_["cluster_passes_off"] = (
                    f"a hand-off timed out earlier ({down['reason']}); re-armed in {down['fits_left']} fit(s)" if down
                    else "switched off (tunable / MBPLS_HIP_CLUSTER_PASSES=0)" if not getattr(eng.ctx, "cluster_passes", 0)
                    else "the device cannot hold a cluster of this shape at once (occupancy query / refused cooperative launch)")
        return self

    def check_sparsity_level(self, data):
        """(sparse_rows, sparse_columns, dense_rows, dense_columns) index arrays of a HOST array, with the
        reference's warning when more than half of the rows or columns have missing entries (mbpls.py:255-271).
        ``fit`` derives the same lists for the device-resident blocks from kernel outputs instead."""
        miss = np.isnan(np.asarray(data, dtype=np.float64))
        rows, cols = miss.any(axis=1), miss.any(axis=0)
        for flags in (cols, rows):
            if flags.size and flags.sum() / flags.size > 0.5:
                warnings.warn("The sparsity of your data is likely to high for this algorithm. This can cause "
                              "either convergenceproblems or crash the algorithm.")
        return np.where(rows)[0], np.where(cols)[0], np.where(~rows)[0], np.where(~cols)[0]

    def plot(self, num_components=2):
        """The reference's matplotlib overview figure (mbpls.py:1439) is presentation, not part of the accelerated
        path: every quantity it draws is a fitted attribute of this estimator (``A_``, ``P_``, ``Ts_``,
        ``explained_var_xblocks_`` ...)."""
        raise NotImplementedError("MBPLS.plot is not provided by mbpls_amd; plot the fitted attributes directly")

    def _sparsity_info(self, eng, Y, row_mask=None):
        """``sparse_X_info_[b]`` / ``sparse_Y_info_['Y']`` = (sparse_rows, sparse_columns, dense_rows,
        dense_columns) index arrays and the high-sparsity warnings of check_sparsity_level (mbpls.py:255-271,
        :295-296, :313, :321), from the per-column counts of the standardise pass and one indicator pass.  For a
        row-subset fit the indices refer to the included samples, as if the sliced arrays had been passed."""
        def info(sparse_rows, sparse_cols):
            for flags in (sparse_cols, sparse_rows):                                  # :265-270
                if flags.size and flags.sum() / flags.size > 0.5:
                    warnings.warn("The sparsity of your data is likely to high for this algorithm. This can cause "
                                  "either convergenceproblems or crash the algorithm.")
            return (np.where(sparse_rows)[0], np.where(sparse_cols)[0],
                    np.where(~sparse_rows)[0], np.where(~sparse_cols)[0])
        col_present = eng.count.get()
        row_present = eng.present_per_row()
        n_rows = eng.n
        if row_mask is not None:            # excluded samples are zero rows of the working copy: leave them out
            row_present, Y, n_rows = row_present[row_mask], Y[row_mask], int(np.count_nonzero(row_mask))
        sizes = _global_block_sizes(eng)
        self.sparse_X_info_ = {}
        for b, (s, e) in enumerate(zip(eng.block_offsets[:-1], eng.block_offsets[1:])):
            self.sparse_X_info_[b] = info(row_present[:, b] < sizes[b], col_present[s:e] < n_rows)
        self.sparse_Y_info_ = {'Y': info(np.isnan(Y).any(axis=1), np.isnan(Y).any(axis=0))}

    def _store(self, eng, res):
        B = eng.nblocks
        sizes = eng.block_sizes
        self.num_blocks_ = B
        self._block_offsets = list(eng.block_offsets)
        if self.standardize:
            streamed = eng.wait_stats() if hasattr(eng, "wait_stats") else None
            if streamed is not None:                  # left through the copier thread while the fit was running
                mean, scale, var = streamed
            else:
                mean, scale = eng.mean.get(), eng.scale.get()
                var = eng.var.get() if getattr(eng, "var", None) is not None else None
            seen = eng.count.get().astype(np.int64) if eng.sparse else None    # per-column present samples
            rm = getattr(eng, "rowmask", None)
            n_seen = eng.n if rm is None else int(np.count_nonzero(rm))        # row-subset fit: the included samples
            self.x_scalers_ = [ColumnScaler(mean[s:e], scale[s:e], n_seen if seen is None else seen[s:e],
                                            None if var is None else var[s:e])
                               for s, e in zip(eng.block_offsets[:-1], eng.block_offsets[1:])]
        # containers exactly as the reference leaves them (mbpls.py:359-382)
        self.W_non_normal_ = [np.empty((sz, 0)) for sz in sizes]
        self.W_concat_ = np.empty((eng.p, 0))
        for k, v in res.items():
            setattr(self, k, v)
        if self.method == 'SIMPLS':
            self.explained_var_xblocks_ = np.empty((B, 0))
            self.explained_var_x_ = []
            self.explained_var_y_ = []

    # ------------------------------------------------------------------ predict / transform
    def _check_fitted(self):
        if not hasattr(self, 'beta_'):
            raise NotFittedError("This MBPLS instance is not fitted yet. Call 'fit' with appropriate "
                                 "arguments before using this estimator.")

    def _device_new(self, eng, X):
        """Upload new data; returns (matrix, mean_dev, scale_dev).  Standardisation with the fitted
        scalers (mbpls.py:1369,:1372,:1097,:1103) is applied on the fly inside the x_w kernel."""
        if isinstance(X, _capi.DeviceMatrix):
            mat = X
        else:
            blocks = self._host_blocks(X)
            if len(blocks) != self.num_blocks_ or any(b.shape[1] != e - s for b, s, e in zip(
                    blocks, self._block_offsets[:-1], self._block_offsets[1:])):
                raise ValueError("X has a different block structure than the data the model was fitted on")
            eng.load_host_blocks(blocks)
            mat = eng.X
        mean = scale = None
        if self.standardize:
            mean = eng.ctx.from_host(np.concatenate([s.mean_ for s in self.x_scalers_]))
            scale = eng.ctx.from_host(np.concatenate([s.scale_ for s in self.x_scalers_]))
        return mat, mean, scale

    @_one_at_a_time
    def predict(self, X, copy=True):
        """Predict y based on the fitted model (reference mbpls.py:1337): (X_std beta) sigma_y + mu_y."""
        self._check_fitted()
        eng = self._engine()
        try:
            mat, mean, scale = self._device_new(eng, X)
            dB = eng.ctx.from_host(self.beta_)
            y = eng._xw(dB, self.beta_.shape[1], False, mean, scale, mat=mat,      # :1386 / :1408
                        nan="skip" if self.sparse_data is True else None)           # :1376-1384
            _require_finite(y)
        finally:
            eng.close()
        if self.standardize:
            y = self.y_scaler_.inverse_transform(y)
        return y

    @_one_at_a_time
    def transform(self, X, Y=None, return_block_scores=False, copy=True):
        """Scores of new data (reference mbpls.py:1052): Ts = X_std R_; block scores by the
        deflation-free identity T_b = X_b W_b - Ts triu(P_b^T W_b, 1) (== the sequential deflation
        of :1129-1155); Y scores U = colnorm(Y_std V_) (:1166)."""
        self._check_fitted()
        eng = self._engine()
        try:
            mat, mean, scale = self._device_new(eng, X)
            K = self.R_.shape[1]
            dR = eng.ctx.from_host(self.R_)
            nan = "skip" if self.sparse_data is True else None
            Ts = eng._xw(dR, K, False, mean, scale, mat=mat, nan=nan)               # :1117 / :1250, :1106-1115
            _require_finite(Ts)
            T = None
            if self.method != 'SIMPLS' and (return_block_scores) and np.vstack(self.W_).shape[1] == 0:
                # KERNEL fitted with calc_all=False has no block weights (:740 is skipped): the reference's loop
                # (:1129-1155) then stacks K products with empty slices -- block scores without columns
                T = [np.empty((Ts.shape[0], 0)) for _ in range(self.num_blocks_)]
            elif self.method != 'SIMPLS' and (return_block_scores):
                W = np.vstack(self.W_)
                dW = eng.ctx.from_host(W)
                XW = eng._xw(dW, K, True, mean, scale, mat=mat, nan=nan)
                T = []
                if nan is None:
                    for b in range(self.num_blocks_):
                        PtW = eng.comm.allreduce_host(self.P_[b].T.dot(self.W_[b]))
                        T.append(XW[b] - Ts.dot(np.triu(PtW, 1)))
                else:
                    # missing values: the deflated row i is x_i - sum_l Ts[i,l] P[:,l] on its PRESENT columns only
                    # (:1133-1153), so P_b[:,l]^T W_b[:,k] becomes a per-row sum over the present columns:
                    # one indicator product with the columns P[:,l] * W[:,k], l < k
                    pairs = [(l, k) for k in range(K) for l in range(k)]
                    T = [xw.copy() for xw in XW]
                    if pairs:
                        P = np.vstack(self.P_)
                        C = np.empty((W.shape[0], len(pairs)), order="F")
                        for i, (l, k) in enumerate(pairs):
                            np.multiply(P[:, l], W[:, k], out=C[:, i])
                        dC = eng.ctx.from_host(C)
                        M = eng._xw(dC, len(pairs), True, None, None, mat=mat, nan="indicator")
                        for b in range(self.num_blocks_):
                            for i, (l, k) in enumerate(pairs):
                                T[b][:, k] -= Ts[:, l] * M[b][:, i]
        finally:
            eng.close()
        out = [Ts]
        if T is not None:
            out.append(T)
        if Y is not None:
            if self.sparse_data is True:                                            # :1120, NaN allowed
                # (numpy >= 2: copy=False raises when a conversion is needed; check_array(copy=False) copies only then)
                Y = np.array(Y, dtype=np.float64) if copy else np.asarray(Y, dtype=np.float64)
                if np.isinf(Y).any():
                    raise ValueError("Input Y contains infinity or a value too large for dtype('float64').")
            else:
                Y = check_array(Y, dtype=np.float64, ensure_2d=False, copy=copy)   # :1120
            if Y.ndim == 1:
                Y = Y.reshape(-1, 1)
            if self.standardize:
                Y = self.y_scaler_.transform(Y)                                     # :1125
            if self.sparse_data is True:
                Y = np.where(np.isnan(Y), 0.0, Y)           # :1157-1163: missing responses do not contribute
            YV = Y.dot(self.V_)
            out.append(YV / np.linalg.norm(YV, axis=0))                             # :1166
        return out[0] if len(out) == 1 else tuple(out)

    def fit_transform(self, X, y=None, **fit_params):
        """Fit the model, then transform the given data (reference mbpls.py:1412)."""
        return self.fit(X, y, **fit_params).transform(X, y)

    def fit_predict(self, X, Y, **fit_params):
        """Fit to data, then predict it (reference mbpls.py:1418)."""
        return self.fit(X, Y, **fit_params).predict(X)

    def r2_score(self, X, Y):
        """reference mbpls.py:1424 (variance-weighted when the data is not standardised)."""
        from sklearn import metrics
        if self.standardize:
            return metrics.r2_score(Y, self.predict(X))
        return metrics.r2_score(Y, self.predict(X), sample_weight=None, multioutput='variance_weighted')

    def explained_variance_score(self, X, Y):
        """reference mbpls.py:1431."""
        from sklearn import metrics
        if self.standardize:
            return metrics.explained_variance_score(Y, self.predict(X))
        return metrics.explained_variance_score(Y, self.predict(X), sample_weight=None,
                                                multioutput='variance_weighted')
