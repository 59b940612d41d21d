"""Device engine: owns the (deflating) feature-major X in HBM and issues the HIP kernels.

The solvers in ``solvers.py`` are written against this small interface; every quantity that
is a sum over feature columns passes through ``comm`` so the same solver code runs on one
GPU and column-sharded over several.  Small results come back as host numpy
arrays (the tiny n x B / n x q algebra and the loop-exit decision stay on the host, as in
the reference); p-length results stay in named device *slots* (p x K, column-major) until
``fetch``.
"""
from __future__ import annotations

import contextlib
import math
import sys
import time

import numpy as np

from . import _capi
from .comm import LocalComm
from .engine_base import EngineBase


class HipEngine(EngineBase):
    has_association_matrices = True
    has_device_helpers = True
    KD = 40                     # == KD of csrc/colpass.hip (max lazily deflated components + 1)
    MAT_CACHE_BYTES = 32 << 20  # uploaded matrices up to this size are kept (the four latest shapes) for the next fit
    # Deflation strategy of the NIPALS / UNIPALS(p>n) drivers.  "lazy": X in HBM is never rewritten, the
    # rank-1 updates X <- X - ts p^T (mbpls.py:969) are applied algebraically inside the next passes
    # (one READ of X per loadings+deflation+sweep instead of read+write).  "explicit": in-place update.
    deflation = "lazy"

    def __init__(self, ctx=None, comm=None, device=0):
        self.ctx = ctx if ctx is not None else _capi.Context(device)
        self.comm = comm if comm is not None else LocalComm()
        self.X = None            # working matrix (standardised, deflated in place)
        self._owns_X = True
        self.slots = {}
        self._xchg = None
        self._xw_buf = None
        self.mean = None
        self.scale = None
        self.G = None
        self._init_base()
        # missing values (sparse_data=True): NaN-aware column passes (csrc/colpass_nan.hip); set before load_*
        self.sparse = False

    @contextlib.contextmanager
    def _timed(self, name):
        t0 = time.perf_counter()
        try:
            yield
        finally:
            self.timings[name] = self.timings.get(name, 0.0) + time.perf_counter() - t0

    # ------------------------------------------------------------------ data
    def load_host_blocks(self, blocks):
        """Upload host blocks (each n x p_b) side by side into one feature-major matrix.
        Replaces deepcopy / check_array(copy) / hstack of the reference (mbpls.py:303,310,379):
        the caller's arrays are only ever read."""
        n = blocks[0].shape[0]
        offs = np.concatenate(([0], np.cumsum([b.shape[1] for b in blocks]))).astype(np.int64)
        # small problems: the matrix object of the previous fit of this very shape, if there was one -- creating one
        # costs seven allocations and uploads for the block offsets and chunk tables (70 us of a 0.6 ms fit at the
        # README quick-start size); every column is overwritten by the uploads below
        self._mat_key = (int(n), tuple(int(o) for o in offs))
        X = self.ctx._mat_cache.pop(self._mat_key, None)
        if X is None:
            X = _capi.DeviceMatrix(self.ctx, n, int(offs[-1]), offs)
        for b, o in zip(blocks, offs[:-1]):
            X.upload(b, int(o))
        self._set_matrix(X, owns=True)
        self._uploaded = X
        self._hint_key = self._mat_key              # (offset hint of host data: by shape, see _offset_hint)
        return self

    def load_device(self, dmat, copy=True):
        """Adopt a matrix already resident in HBM.  copy=True keeps the caller's matrix intact
        (the standardise pass writes into a second matrix); copy=False deflates it in place."""
        # copy=True: the working copy is allocated by prepare() -- and only if the fit writes X at all (with lazy
        # deflation and implicit standardisation it never does, and the caller's matrix is used as it is)
        self._src = dmat if copy else None
        self._set_matrix(dmat, owns=False)
        self._hint_owner = dmat                     # (offset hint of a resident matrix: kept on the matrix object)
        return self

    # Raw spectra, counts, intensities: the columns of real data sit many standard deviations away from zero, and the
    # raw-space passes of a never-writes-X fit then have to centre in registers (CENTRE_FROM_OFFSET).  A device-resident
    # queue only learns that from its own statistics pass: it stops, and the fit is queued again with centring passes --
    # one statistics pass and a queue thrown away, 0.3 ms of a 0.55 ms fit on the reference's FTIR data, a whole pass
    # over X at any size.  The measure of the last fit of a resident matrix (or of host data of this shape) is therefore
    # remembered: the next fit starts with the flavour that one needed, and drops it again when the step kernel reports
    # a measure that no longer asks for it.  Never under column shards: every rank must queue the same passes.
    OFFSET_HINTS = 32
    OFFSET_ASK_BYTES = 1 << 30      # matrices from this size on: an unknown matrix waits for its statistics pass instead

    def _offset_hint(self):
        """True / False: what the last fit of this matrix (host data: of this shape) measured; None: nothing known"""
        if self.comm.is_distributed:
            return None
        owner = getattr(self, "_hint_owner", None)
        if owner is not None:
            return getattr(owner, "_offset_hint", None)
        return self.ctx._offset_hints.get(getattr(self, "_hint_key", None))

    def _set_offset_hint(self, needed):
        if self.comm.is_distributed:
            return
        owner = getattr(self, "_hint_owner", None)
        if owner is not None:
            owner._offset_hint = bool(needed)
            return
        key = getattr(self, "_hint_key", None)
        if key is None:
            return
        hints = self.ctx._offset_hints
        hints.pop(key, None)
        while len(hints) >= self.OFFSET_HINTS:
            hints.pop(next(iter(hints)))
        hints[key] = bool(needed)

    def _measured_offset(self):
        """max_j (mean_j / scale_j)^2 as the statistics pass left it in the exchange buffer (waits for that pass)"""
        n, B = self.n, self.nblocks
        self.ctx.sync()
        return float(np.max(_capi.DeviceArray(self.ctx, (B,), ptr=self._xptr(n * B + 3 * B)).get()))

    def _working_copy(self):
        """the matrix the standardise pass writes into: X itself when it is ours to overwrite, else a new one"""
        if self._src is not None:
            W = _capi.DeviceMatrix(self.ctx, self._src.n, self._src.p, self._src.block_offsets)
            self.X, self._owns_X = W, True
        return self.X

    def _set_matrix(self, X, owns):
        self.X = X
        self._owns_X = owns
        self.n, self.p, self.nblocks = X.n, X.p, X.nblocks
        self.block_offsets = list(X.block_offsets)
        self.block_sizes = [self.block_offsets[b + 1] - self.block_offsets[b] for b in range(self.nblocks)]
        B, n = self.nblocks, self.n
        # exchange buffer: [t: n*B][ssq_g, ssq_p, ssq_x: 3*B][lazy-deflation sums: B*KD][sparse only, d: n*B]
        self._d_off = n * B + 3 * B + B * self.KD
        # (the last word: under column shards the count of timed-out cluster hand-offs of every rank rides on the
        # all-reduce of the buffer -- _stage_cluster_flag / check_cluster_passes)
        self._flag_off = self._d_off + (n * B if self.sparse else 0)
        self._xchg = self.comm.exchange_buffer(self.ctx, self._flag_off + 1)
        self._vec_n = [self.ctx.empty(n), self.ctx.empty(n)]
        self._coef = [self.ctx.empty(self.KD), self.ctx.empty(self.KD)]   # <ts_i, v>, <ts_i, ts>
        self._lazy_ts = []          # score vectors of the components deflated so far (lazy mode)
        self._lazy_pslot = None
        self._istd = False          # implicit standardisation: X in HBM stays raw (see prepare)
        self._cen = False           # ... and the raw-space passes subtract the column means (offset data)
        self._kscale = False        # statistics-only preparation: X stays raw, the products standardise on the fly
        self.host_stats = None      # (mean, scale, var) on the host when they were streamed out during the fit
        self.rs = self.r2 = None
        if not hasattr(self, "_src"):
            self._src = None

    def close(self):
        if self.ctx.handle and getattr(self, "_sent", 0):
            # a fit that failed with columns on their way: they land before the slots they come from are released
            try:
                self.wait_sent(self._sent)
            except Exception:       # noqa: BLE001 -- the failure that brought us here is the one to report
                self._sent, self._sent_keep = 0, []
        if self.ctx.handle:
            self.ctx.lib.mbpls_hip_set_true_space_outputs(self.ctx.handle, None, None)
        for s in self.slots.values():
            s.free()
        self.slots = {}
        if self.X is not None and self._owns_X:
            cache = self.ctx._mat_cache if self.ctx.handle else None
            if (cache is not None and self.X is getattr(self, "_uploaded", None) and self.X.nbytes <= self.MAT_CACHE_BYTES
                    and self._mat_key not in cache):
                while len(cache) >= 4:                  # four shapes at most: the oldest one goes
                    cache.pop(next(iter(cache))).close()
                cache[self._mat_key] = self.X           # (kept for the next fit of this shape; Context.trim_pool drops it)
            else:
                self.X.close()
        self.X = self._uploaded = None
        if self.G is not None:
            self.G[0].close()
            if hasattr(self.G[1], "free"):
                self.G[1].free()
            self.G = None
        if getattr(self, "_xtx", None) is not None:
            self._xtx[0].close()
            self._xtx[1].free()
            self._xtx = None
        if getattr(self, "Gsum", None) is not None:
            self.Gsum[0].close()
            self.Gsum[1].free()
            self.Gsum = None
        if getattr(self, "Gb", None) is not None:
            self.Gb[0].close()
            for a in self.Gb[1:]:
                if hasattr(a, "free"):
                    a.free()
            self.Gb = None
        if getattr(self, "_ss", None) is not None:
            for key in ("Y", "inp", "out", "moments"):
                self._ss[key].free()
            self._ss = None
        if getattr(self, "_gp", None) is not None:
            for a in [self._gp[key] for key in ("Y", "a", "ts", "u", "v", "state")] + self._gp["W"]:
                a.free()
            self._gp = None

    # ------------------------------------------------------------------ slots (p x K device arrays)
    def slot(self, name, K, zero=True):
        s = self.slots.get(name)
        if s is None or s.shape != (self.p, K):
            if s is not None:
                s.free()
            s = self.ctx.empty((self.p, K), zero=zero)
            self.slots[name] = s
        return s

    def fetch_col(self, name, k, out):
        """D2H of column k of a slot into column k of the F-ordered host array ``out`` on the copy stream:
        it does not wait for the kernel currently running on the compute stream (the column must be final)."""
        S = self.slots[name]
        col = out[:, k]
        assert col.flags.c_contiguous and col.size == self.p
        with self._timed("d2h_overlapped"):
            _capi.check(self.ctx.lib.mbpls_hip_memcpy_d2h_overlap(self.ctx.handle, col.ctypes.data,
                                                                  S.ptr + 8 * self.p * k, 8 * self.p))

    def host_array(self, shape, reserve=None, prefault=True, prefault_tail_only=False):
        """F-ordered host array for a p x K attribute.  A device-to-host copy into a fresh numpy array runs at
        page-fault speed (23 GB/s measured; 55 GB/s once the pages are mapped), so large arrays are handed to a helper
        thread that maps their pages (madvise(MADV_POPULATE_WRITE): contents untouched, safe beside a copy that is
        already writing) while the GPU works -- every fit, also the first of a fresh estimator, then receives its
        attributes at PCIe speed.  (Round 2 recycled the previous fit's buffers behind a reference-count heuristic
        instead; that only helped refits of the same estimator.)  ``reserve``: keep the array for the later fetch of
        the slot of that name."""
        ready = getattr(self, "_host_ready", None)
        if ready is None:
            ready = self._host_ready = {}
        out, mapped = _pooled_host_array(self.ctx, shape)
        if mapped:                                  # a buffer an earlier model has let go of: its pages are there
            pass
        elif prefault:
            # (the LAST column first: it is the one that arrives behind the last pass, when nothing hides a page fault)
            _prefault(out, tail_bytes=8 * int(shape[0]) if len(shape) == 2 else 0)
        elif prefault_tail_only and len(shape) == 2:
            _prefault(out, tail_bytes=8 * int(shape[0]), only_tail=True)
        if reserve is not None:
            ready[reserve] = out
        return out

    def _host_for_slot(self, name, shape):
        a = getattr(self, "_host_ready", {}).pop(name, None)
        if a is not None and a.shape == tuple(shape):
            return a
        # not reserved ahead: the copy that follows at once maps the pages itself -- handing the array to the helper
        # thread as well would only keep that thread busy (80 MB: ~50 ms) when the NEXT fit wants its R_ / beta_ mapped
        return self.host_array(shape, prefault=False)

    def fetch(self, name, overlapped=False, out=None):
        """D2H of a whole slot.  overlapped=True: on the copy stream, not ordered behind the kernels queued on
        the compute stream (the slot must be final); safe to call from a helper thread.  ``out``: an F-ordered host
        array of the slot's shape to copy into (repeated fetches of a scratch slot)."""
        S = self.slots[name]
        if out is None:
            out = self._host_for_slot(name, S.shape)
        assert out.shape == S.shape and out.flags.f_contiguous
        if not overlapped:
            with self._timed("d2h_slots"):
                _capi.check(self.ctx.lib.mbpls_hip_memcpy_d2h(self.ctx.handle, out.ctypes.data, S.ptr, out.nbytes))
            return out
        _capi.check(self.ctx.lib.mbpls_hip_memcpy_d2h_overlap(self.ctx.handle, out.ctypes.data, S.ptr, out.nbytes))
        return out

    def fetch_async(self, name):
        """D2H of a whole slot by the copier thread, behind everything queued on the compute stream so far; returns
        (host array, wait) -- wait() blocks until the array is complete (and re-raises a copy error)."""
        S = self.slots[name]
        out = self._host_for_slot(name, S.shape)
        ticket = self.ctx.copier_submit([(out, S.ptr)])

        def wait():
            self.ctx.copier_wait(ticket)
            return out

        return out, wait

    def send_cols(self, cols, unit=None):
        """the copier thread carries the columns out behind everything queued on the compute stream so far, and forms
        the block-normalised column from the one that has just arrived (csrc/copier.hip)"""
        scale = None
        if unit is not None:
            src, dst, k, nu = unit
            offs = getattr(self, "_offs64", None)
            if offs is None:
                offs = self._offs64 = np.ascontiguousarray(self.block_offsets, dtype=np.int64)
            nu2 = np.ascontiguousarray(np.asarray(nu, dtype=np.float64).ravel() ** 2)     # (the thread divides by its root)
            self._sent_keep = getattr(self, "_sent_keep", []) + [nu2]
            scale = (src[:, k], dst[:, k], offs, nu2.ctypes.data)
        self._sent = self.ctx.copier_submit([(out[:, k], self.slots[name].ptr + 8 * self.p * k) for name, k, out in cols],
                                            scale=scale)
        return self._sent

    def send_slot(self, name, out):
        S = self.slots[name]
        assert out.shape == S.shape and out.flags.f_contiguous
        self._sent = self.ctx.copier_submit([(out, S.ptr)])
        return self._sent

    def wait_sent(self, ticket):
        with self._timed("wait_copies"):
            self.ctx.copier_wait(ticket)
        if ticket >= getattr(self, "_sent", 0):
            self._sent, self._sent_keep = 0, []

    def put(self, name, arr):
        arr = np.asarray(arr, dtype=np.float64)
        if arr.ndim == 1:
            arr = arr.reshape(-1, 1)
        return self.slot(name, arr.shape[1]).set(arr)

    def cross(self, a, b, rows=None):
        """(slot a[rows, ka:ka+ca])^T (slot b[rows, kb:kb+cb]) -> host ca x cb, summed over shards.
        a, b are (name, k0, ncols) triples; rows = (r0, r1) restricts the sum to a block of features."""
        (na, ka, ca), (nb, kb, cb) = a, b
        A, Bm = self.slots[na], self.slots[nb]
        r0, r1 = rows if rows is not None else (0, self.p)
        out = np.zeros((ca, cb), dtype=np.float64)
        if r1 > r0:
            _capi.check(self.ctx.lib.mbpls_hip_cross_product(self.ctx.handle, r1 - r0, A.ptr + 8 * (self.p * ka + r0),
                                                             self.p, ca, Bm.ptr + 8 * (self.p * kb + r0), self.p, cb,
                                                             out.ctypes.data))
        return self.comm.allreduce_host(out)

    def right_mul(self, a, M, c, rows=None):
        """slot c[rows, kc:kc+c2] = slot a[rows, ka:ka+c1] @ M  (M: c1 x c2 host).  rows = (r0, r1) or all."""
        (na, ka, c1), (nc, kc) = a, c
        M = np.ascontiguousarray(M, dtype=np.float64)
        assert M.shape[0] == c1
        r0, r1 = rows if rows is not None else (0, self.p)
        if r1 <= r0:
            return
        A, Cm = self.slots[na], self.slots[nc]
        _capi.check(self.ctx.lib.mbpls_hip_right_multiply(self.ctx.handle, r1 - r0, A.ptr + 8 * (self.p * ka + r0),
                                                          self.p, c1, M.ctypes.data, M.shape[1],
                                                          Cm.ptr + 8 * (self.p * kc + r0), self.p))
        self._keep_m = getattr(self, "_keep_m", [])[-32:] + [M]

    def put_col(self, name, k, vec):
        self.slots[name].col(k).set(np.ascontiguousarray(vec, dtype=np.float64).ravel())

    def _slot_col(self, spec):
        if spec is None:
            return None
        name, k = spec
        return self.slots[name].col(k)

    def sync(self):
        self.ctx.sync()

    def run_deferred(self, count=1, all_of_it=False):
        """Host chores queued by the estimator (releasing the attributes a refit replaces): a few per column
        pass, called right after a launch so they run while the GPU works."""
        work = getattr(self, "deferred_host_work", None)
        while work and (all_of_it or count > 0):
            work.pop()()
            count -= 1

    # ------------------------------------------------------------------ exchange helpers
    def _xptr(self, off):
        return self._xchg.ptr + 8 * off

    def use_lazy_deflation(self, K, pslot, prefer=None):
        """Called by a solver whose passes all go through sweep/deflate_sweep: enable lazy deflation
        when the component count fits the kernel's correction registers.  ``prefer`` is the solver's own
        choice ("lazy"/"explicit"); the MBPLS_DEFLATION environment variable overrides both."""
        mode = self.opts.deflation or prefer or self.deflation
        if self.sparse:
            mode = "explicit"      # the implicit correction would need every row's own missing-value mask
        if not self._raw_space_ok():
            mode = "explicit"      # register-resident columns above 8192 rows (spilling configurations): no lazy flavour
        self._lazy_pslot = pslot if (mode == "lazy" and K < self.KD - 1) else None
        self._lazy_ts = []
        self._lazy_K = K
        self.plan["deflation"] = "lazy" if self._lazy_pslot is not None else "explicit"
        return self._lazy_pslot is not None

    def _lazy_coef(self, idx, vec):
        """upload <ts_i, vec> for the deflated components; returns the device array (or None)"""
        if not self._lazy_ts:
            return None
        c = np.array([float(t.ravel().dot(vec.ravel())) for t in self._lazy_ts])
        buf = np.zeros(self.KD)
        buf[:c.size] = c
        return self._coef[idx].set(buf)

    def _lazy_project(self, v, extra_ts=None):
        """Sweep vector of a lazy pass minus its components along the scores deflated so far (and ``extra_ts``, the score
        the pass itself deflates).  X_k^T v is the same for both (X_k = (I - Ts Ts^T) X_0), but the pass takes it as
        X_0^T v - sum_i P[:, i] <ts_i, v>: with the projected vector the two terms are no longer large against their
        difference -- without it the rounding of deep components (y long since explained: <ts_i, y> big, <x_j^k, y>
        tiny) was amplified until ||ts_old - ts|| <= 1e-14 (mbpls.py:841) went out of reach beyond ~25 components.
        Same re-orthogonalisation as the device-resident step (csrc/step_reg.hip)."""
        cols = list(self._lazy_ts) + ([np.asarray(extra_ts, dtype=np.float64).ravel()] if extra_ts is not None else [])
        if not cols or v is None:
            return v
        T = np.column_stack(cols)
        v = np.array(v, dtype=np.float64).reshape(-1, 1)
        return v - T.dot(T.T.dot(v) / (T * T).sum(axis=0)[:, None])

    def _lazy_fix(self, t, d, extra_ts=None):
        """t~_b = A_b - Ts d_b  (A_b: raw accumulation over the undeflated columns)"""
        cols = list(self._lazy_ts) + ([extra_ts] if extra_ts is not None else [])
        if not cols or t is None:
            return t
        Tm = np.column_stack([c.ravel() for c in cols])            # n x k
        return t - Tm.dot(d[:, :Tm.shape[1]].T)                     # d: B x k

    def _finish_pass(self, want_t, want_g, want_p, want_x, nd=0):
        """all-reduce the exchange buffer (if sharded) and split it on the host."""
        n, B = self.n, self.nblocks
        self._last_d = None
        if self.comm.is_distributed:
            with self._timed("allreduce"):
                staged = self._stage_cluster_flag()
                self.comm.allreduce_device(self._xchg)
        with self._timed("wait_kernel+d2h_small"):
            h = self._xchg.get()
        self.check_cluster_passes(shared=h[self._flag_off] if (self.comm.is_distributed and staged) else None)
        t = h[: n * B].reshape(B, n).T.copy() if want_t else None   # n x B
        o = n * B
        if self.sparse and want_t:
            # t_i = nu A_i / D_i over the present columns (mbpls.py:869-872); handed back as nu^2 A_i / D_i so
            # that the caller's t / nu is the block score.  Rows without missing entries have D_i = nu^2.
            d = h[self._d_off:self._d_off + n * B].reshape(B, n).T
            ssq = h[o:o + B]
            with np.errstate(divide="ignore", invalid="ignore"):
                t = np.where(d > 0.0, t * (ssq[None, :] / d), 0.0)
        if nd:
            self._last_d = h[o + 3 * B:o + 3 * B + B * nd].reshape(B, nd).copy()
        return (t, h[o:o + B].copy() if want_g else None, h[o + B:o + 2 * B].copy() if want_p else None,
                h[o + 2 * B:o + 3 * B].copy() if want_x else None)

    def _put_vec(self, idx, v):
        d = self._vec_n[idx]
        d.set(np.ascontiguousarray(v, dtype=np.float64).ravel())
        return d

    # ------------------------------------------------------------------ hot path
    def prepare(self, standardize, v=None, alpha=1.0, g_slot=None, mean=None, scale=None, defer=False):
        """a1 (+ first sweep): standardise X (or just take its sums of squares) and optionally run the
        first sweep of component 0 in the same pass.  Returns (ssq_x[B], t[n,B] or None, ssq_g[B] or None).
        defer=True only queues the pass (device-resident fit: the exchange buffer is consumed on the device)."""
        n, B = self.n, self.nblocks
        self.var = None
        if standardize:
            self.mean = self.ctx.empty(self.p)
            self.scale = self.ctx.empty(self.p)
            mode = _capi.SCALE_FIT
            if mean is None:
                self.var = self.ctx.empty(self.p)      # StandardScaler.var_ of the fitted scalers
            if mean is not None:
                self.mean.set(mean)
                self.scale.set(scale)
                mode = _capi.SCALE_APPLY
        else:
            mode = _capi.SCALE_NONE
        src = self._src if self._src is not None else self.X
        _capi.check(self.ctx.lib.mbpls_hip_set_variance_output(self.ctx.handle, _capi._ptr(self.var)))
        self.host_stats = None
        try:
            out = self._prepare_passes(src, standardize, mode, v, alpha, g_slot, mean, defer)
        finally:
            self.ctx.lib.mbpls_hip_set_variance_output(self.ctx.handle, None)
        if standardize and mode == _capi.SCALE_FIT:
            self.stream_stats()
        return out

    def stream_stats(self):
        """mean / scale / var are final once the standardise pass has run: send them to the host through the copier
        thread now (behind an event), so the three p-length copies -- page faults of fresh arrays included -- overlap
        with the rest of the fit instead of sitting at its end.  ``host_stats`` / ``wait_stats`` hand them over."""
        if self.mean is None or self.scale is None or not self.opts.stream_stats:
            return
        arrs = [self.mean, self.scale] + ([self.var] if getattr(self, "var", None) is not None else [])
        S = self.host_array((self.p, 3), prefault=False)   # one F-ordered p x 3 array (copied right after the first pass)
        host = [S[:, i] for i in range(len(arrs))]
        self._stats_ticket = self.ctx.copier_submit([(h, d.ptr) for h, d in zip(host, arrs)])
        self.host_stats = tuple(host) if len(host) == 3 else (host[0], host[1], None)

    def wait_stats(self):
        """(mean, scale, var) on the host, or None when they were not streamed"""
        if self.host_stats is None:
            return None
        self.ctx.copier_wait(self._stats_ticket)
        return self.host_stats

    def _prepare_passes(self, src, standardize, mode, v, alpha, g_slot, mean, defer):
        n, B = self.n, self.nblocks
        if self._implicit_ok(v, mean):
            self.plan["standardisation"] = "implicit (X never written)" if standardize else "none (X never written)"
            out = self._prepare_implicit(src, standardize, mode, v, alpha, g_slot, defer)
            return None if defer else out
        self.plan["standardisation"] = ("stored working copy" if self._src is not None else "stored in place") \
            if standardize else "none"
        self._working_copy()
        dv = self._put_vec(0, v) if v is not None else None
        # row subset (cross-validation fold): excluded samples become zero rows of the working copy and the
        # column statistics are taken over the included ones only
        dm, n_eff = None, 0.0
        if self.sparse:
            self.count = self.ctx.empty(self.p)          # present samples per column
            if getattr(self, "rowmask", None) is not None:   # a fold of data with missing values
                dm = self.ctx.from_host(np.ascontiguousarray(self.rowmask, dtype=np.float64).ravel())
            src.nan_standardize_sweep(self.X, mode, self.mean, self.scale, self.count, self._xptr(n * B + 2 * B), dv,
                                      self._slot_col(g_slot), self._xptr(0) if v is not None else None,
                                      self._xptr(self._d_off) if v is not None else None,
                                      self._xptr(n * B) if v is not None else None, rowmask=dm)
        elif getattr(self, "rowmask", None) is not None:
            m = np.ascontiguousarray(self.rowmask, dtype=np.float64).ravel()
            assert m.size == n
            dm, n_eff = self.ctx.from_host(m), float(np.count_nonzero(m))
        if not self.sparse:
            src.standardize_sweep(self.X, mode, self.mean, self.scale, self._xptr(n * B + 2 * B), dv, alpha,
                                  self._slot_col(g_slot), self._xptr(0) if v is not None else None,
                                  self._xptr(n * B) if v is not None else None, rowmask=dm, n_eff=n_eff)
        self._src = None
        if defer:
            return None
        self.run_deferred(2)        # the standardise pass is the longest of the fit
        t, g, _, x = self._finish_pass(v is not None, v is not None, False, True)
        if not np.all(np.isfinite(x)):
            # any NaN / inf in a column poisons its sum of squares (and its mean / scale)
            raise ValueError("Input X contains NaN, infinity or a value too large for dtype('float64').")
        return x, t, g

    def sweep_reads_once(self):
        """SIMPLS takes its one-fused-sweep-per-component route (p = X^T t and X p in the same read) on register-resident
        columns.  On long columns the fused sweep is one read as well since round 4 (clusters of workgroups), but tall
        data is narrow and SIMPLS then does better with two many-right-hand-side products per component and no host
        round trip per pass (the paper's shapes, N = 20 000: 0.021 / 0.100 s against 0.036 / 0.105 s)."""
        return self.n <= min(8192, getattr(self.ctx, "long_columns_from", 8192)) or self.sparse

    @property
    def cluster_passes(self):
        """the never-writes-X passes of this matrix run on clusters of workgroups (long columns, one read of X)"""
        return (not self.sparse and self.n > getattr(self.ctx, "long_columns_from", 8192)
                and bool(getattr(self.ctx, "cluster_passes", 0)) and hasattr(self.ctx, "cluster_status")
                and self.ctx.cluster_columns(self.n))

    def _cluster_guarded(self):
        return bool(getattr(self.ctx, "cluster_passes", 0) and not self.sparse
                    and self.n > getattr(self.ctx, "long_columns_from", 8192) and hasattr(self.ctx, "cluster_status"))

    def _stage_cluster_flag(self):
        """column shards, in front of the all-reduce of the exchange buffer: this rank's count of timed-out hand-offs
        into the buffer's last word (on the stream), so that every rank reads the sum behind the pass"""
        if not self._cluster_guarded():
            return False
        _capi.check(self.ctx.lib.mbpls_hip_cluster_status_into(self.ctx.handle, self._xptr(self._flag_off)))
        return True

    def check_cluster_passes(self, shared=None):
        """Long columns: the workgroups of a cluster hand partial sums to each other inside a launch, every wait
        bounded; a wait that gave up leaves garbage behind -- never hand that to the caller, and never iterate on it:
        called behind every pass of a host-driven loop (the exchange buffer has just been read, so the check is a
        read of one host-mapped word), when a device-resident queue returns, and at the end of the fit.
        Column shards: the decision is COLLECTIVE -- a rank that raised alone would run its fit again (load,
        collectives of the first pass) while its peers, several collectives ahead on its garbage sums, wait for other
        ones.  Behind a pass the ranks' counts arrive summed in the exchange buffer (``shared``); elsewhere (end of a
        queue, end of the fit) they are summed here.  Every rank then raises, or none does."""
        if not self._cluster_guarded():
            return
        bad = self.ctx.cluster_status()                       # (also clears this rank's flag)
        if self.comm.is_distributed:
            if shared is None:
                shared = self.comm.allreduce_host(np.array([float(bad)]))[0]
            bad = max(bad, int(shared)) if np.isfinite(shared) else max(bad, 1)
        if bad:
            raise _capi.ClusterTimeout(f"{bad} hand-off(s) between the workgroups of a column cluster timed out "
                                       "(csrc/colpass_cluster.hip)")

    def scaled_products(self):
        """after an implicit first pass (raw matrix, statistics known): the many-right-hand-side products X^T V / X W
        standardise their operand in registers as well"""
        self._kscale = bool(self._istd)

    def stats_only_ok(self):
        """The products of this fit (Gram, X^T V, X W) can standardise their operand in registers -- and should: the
        Gram pays ~6 % for it (a fifth load per k-step plus the operand arithmetic: 2.00 s instead of 1.88 s at
        BASELINE config[3]), more than the read-only statistics pass saves (27 ms instead of 57 ms), so the stored path
        is kept whenever a standardised working copy comfortably fits beside the caller's matrix (or copy=False lets
        the matrix itself be overwritten) and the never-writes-X path takes over when it does not (config[3]: 131 GB
        of X on a 288 GB device).  MBPLS_KERNEL_IMPLICIT=1/0 forces either."""
        if (self.sparse or getattr(self, "rowmask", None) is not None or not hasattr(self.X, "standardize_stats_sweep")
                or not self._raw_space_ok() or not self.opts.implicit_std):
            return False
        if self.opts.kernel_implicit is not None:
            return self.opts.kernel_implicit == "1"
        if self._src is None:
            return False                              # the matrix is ours to overwrite: standardise in place
        need = self.X.nbytes + 8 * self.n * self.n * 3 + (17 << 30)     # working copy + Gram + its split-K slices
        free = self.ctx.info()["free_bytes"] + self.ctx._pool_bytes      # pooled buffers are reusable
        return free < 1.1 * need

    def prepare_stats(self, standardize):
        """a1 for the methods whose X-sized work is matrix products (KERNEL: X X^T :704, X^T [U Ts] :731/:734, X_b W_b
        :752): StandardScaler statistics and sums of squares in ONE read, nothing written -- the products apply
        z = (x - mean_j) / scale_j to their operand in registers.  The caller's matrix is used as it is: copy=True
        costs no second 8np-byte buffer (BASELINE config[3]: 131 GB instead of 262) and the read+write standardise
        pass (55 ms there) becomes a read-only one.  Returns ssq_x[B]."""
        n, B = self.n, self.nblocks
        src = self._src if self._src is not None else self.X
        self.var = None
        mode = _capi.SCALE_NONE
        if standardize:
            self.mean, self.scale = self.ctx.empty(self.p), self.ctx.empty(self.p)
            self.rs, self.r2, self.var = self.ctx.empty(self.p), self.ctx.empty(self.p), self.ctx.empty(self.p)
            mode = _capi.SCALE_FIT
        dv = self._put_vec(0, np.zeros(n))         # the pass has a fused sweep; with v = 0 it only costs its flops
        _capi.check(self.ctx.lib.mbpls_hip_set_variance_output(self.ctx.handle, _capi._ptr(self.var)))
        try:
            src.standardize_stats_sweep(mode, self.mean, self.scale, self.rs, self.r2, self._xptr(n * B + 2 * B),
                                        self._xptr(n * B + 3 * B), dv, 1.0, None, self._xptr(0), self._xptr(n * B))
        finally:
            self.ctx.lib.mbpls_hip_set_variance_output(self.ctx.handle, None)
        self.run_deferred(2)
        _, _, _, x = self._finish_pass(False, False, False, True, nd=1)
        if not np.all(np.isfinite(x)):
            raise ValueError("Input X contains NaN, infinity or a value too large for dtype('float64').")
        self.X, self._src = src, None              # the raw matrix, never written
        self.plan["standardisation"] = "statistics only (products standardise in registers)" if standardize else "none"
        self._kscale = bool(standardize)
        self._cen = self._kscale and float(np.sqrt(np.max(self._last_d))) > self.CENTRE_FROM_OFFSET
        if standardize:
            self.stream_stats()
        return x

    # Raw-space dot products carry rounding noise of max_j |mean_j| / scale_j ulps.  Above this offset ratio the
    # raw-space passes subtract the column mean from every element in registers first (template flag CEN of
    # csrc/colpass.hip; 8-13 % slower, the passes are partly issue-bound) -- which is what keeps the NIPALS stop rule
    # ||ts_old - ts|| <= 1e-14 reachable for offset data.  The statistics pass itself always shifts (by the first sample).
    CENTRE_FROM_OFFSET = 8.0

    def _raw_space_ok(self):
        """The never-writes-X passes exist for register-resident columns up to 8192 rows (csrc/colpass.hip) and for the
        two-phase long-column passes (csrc/colpass_long.hip, n > long_columns_from)."""
        return self.n <= 8192 or self.n > getattr(self.ctx, "long_columns_from", 8192)

    def _implicit_ok(self, v, mean):
        return (self._lazy_pslot is not None and v is not None and mean is None and not self.sparse
                and getattr(self, "rowmask", None) is None and hasattr(self.X, "standardize_stats_sweep")
                and self._raw_space_ok() and self.opts.implicit_std)

    def _prepare_implicit(self, src, standardize, mode, v, alpha, g_slot, defer=False):
        """First pass of a fit that never writes X (lazy deflation + implicit standardisation): statistics and the
        first sweep only.  Its offset measure max_j |mean_j|/scale_j decides whether the later raw-space passes centre
        the columns in registers (CENTRE_FROM_OFFSET); with defer=True the pass is only queued and the step kernel
        looks at the measure (nipals_resident)."""
        n, B = self.n, self.nblocks
        if standardize:
            self.rs, self.r2 = self.ctx.empty(self.p), self.ctx.empty(self.p)
        dv = self._put_vec(0, v)
        src.standardize_stats_sweep(mode, self.mean, self.scale, self.rs, self.r2, self._xptr(n * B + 2 * B),
                                    self._xptr(n * B + 3 * B), dv, alpha, self._slot_col(g_slot), self._xptr(0),
                                    self._xptr(n * B))
        if defer:
            self.X, self._src = src, None
            self._istd = bool(standardize)
            if self._istd:
                self.slot("P~raw", self._lazy_K)
                self._true_space()
            return "queued"
        self.run_deferred(2)
        t, g, _, x = self._finish_pass(True, True, False, True, nd=1)
        if not np.all(np.isfinite(x)):
            raise ValueError("Input X contains NaN, infinity or a value too large for dtype('float64').")
        self.X, self._src = src, None        # the raw matrix, never written
        self._istd = bool(standardize)
        self._cen = self._istd and float(np.sqrt(np.max(self._last_d))) > self.CENTRE_FROM_OFFSET
        if self._istd:
            self.slot("P~raw", self._lazy_K)
            self._true_space()
            t = t - t.mean(axis=0)           # raw-space accumulation: remove the column means' share
        return x, t, g

    def _centre(self, v):
        """zero-sum copy of an n-vector (the vectors of a fit on standardised data are centred up to rounding)"""
        v = np.asarray(v, dtype=np.float64)
        return v - v.mean(axis=0)

    def _true_space(self, p_true=None):
        """Raw-space passes store TRUE-space weights (g~_j / scale_j) and, into ``p_true`` (a device column), true-space
        loadings beside the raw-space ones -- inside the pass, no p-sized rescaling kernel behind it
        (mbpls_hip_set_true_space_outputs).  In force until the next call; close() clears it."""
        _capi.check(self.ctx.lib.mbpls_hip_set_true_space_outputs(self.ctx.handle, _capi._ptr(self.rs),
                                                                  _capi._ptr(p_true)))

    def sweep(self, v, alpha, g_slot, overlap=None):
        """a2+a3: g = alpha X^T v (into the slot column), t~_b = X_b g_b, ||g_b||^2.  One read of X.
        ``overlap``: host work to do while the kernel runs (e.g. fetching finished columns)."""
        n, B = self.n, self.nblocks
        if overlap is not None:
            self.ctx.fence()    # (its copies wait for what was queued before this pass, not for the pass)
        if self._istd:
            v = self._centre(v)
        if self._lazy_pslot:
            v = self._lazy_project(v)
        dv = self._put_vec(0, v)
        k = len(self._lazy_ts) if self._lazy_pslot else 0
        if self.sparse:      # alpha is implied: every column divides by its own masked <u, u>
            self.X.nan_sweep(dv, self._slot_col(g_slot), self._xptr(0), self._xptr(self._d_off), self._xptr(n * B))
        elif self._istd:
            self._true_space()
            self.X.sweep_lazy(dv, alpha, self._slot_col(g_slot), self._xptr(0), self._xptr(n * B),
                              self.slots["P~raw"], self.p, k, self._lazy_coef(0, v), self._xptr(n * B + 3 * B),
                              r2=self.r2, mean=self.mean if self._cen else None)
        elif k:
            P = self.slots[self._lazy_pslot]
            self.X.sweep_lazy(dv, alpha, self._slot_col(g_slot), self._xptr(0), self._xptr(n * B), P, self.p, k,
                              self._lazy_coef(0, v), self._xptr(n * B + 3 * B))
        else:
            self.X.sweep(dv, alpha, self._slot_col(g_slot), self._xptr(0), self._xptr(n * B))
        if overlap is not None:
            overlap()
        else:
            self.run_deferred()
        t, g, _, _ = self._finish_pass(True, True, False, False, nd=k)
        if k:
            t = self._lazy_fix(t, self._last_d)
        if self._istd:
            t = t - t.mean(axis=0)           # the column means' share of the raw accumulation (scores are centred)
        return t, g

    def deflate_sweep(self, ts, inv_tt, deflate, p_slot, v=None, alpha=1.0, g_slot=None, overlap=None):
        """a5+a7 (+ next component's first sweep): p = inv_tt X^T ts, X -= ts p^T.  One read + one write."""
        if overlap is not None:
            self.ctx.fence()    # (its copies wait for what was queued before this pass, not for the pass)
        n, B = self.n, self.nblocks
        if self._istd:       # raw-space pass: zero-sum vectors
            ts = self._centre(ts)
            v = self._centre(v) if v is not None else None
        if self._lazy_pslot and v is not None:
            v = self._lazy_project(v, extra_ts=ts if deflate else None)
        dts = self._put_vec(1, ts)
        dv = self._put_vec(0, v) if v is not None else None
        if self.sparse:      # inv_tt is implied: every column divides by its own masked <ts, ts>
            self.X.nan_deflate_sweep(dts, deflate, self._slot_col(p_slot), self._xptr(n * B + B), dv,
                                     self._slot_col(g_slot), self._xptr(0) if v is not None else None,
                                     self._xptr(self._d_off) if v is not None else None,
                                     self._xptr(n * B) if v is not None else None)
            if overlap is not None:
                overlap()
            else:
                self.run_deferred()
            t, g, p, _ = self._finish_pass(v is not None, v is not None, True, False)
            return p, t, g
        if self._kscale and not self._lazy_pslot:
            # statistics-only preparation (prepare_stats): X is the caller's RAW matrix and is never deflated either
            # (several responses, solvers._fit_nipals_subspace) -- loadings and sweep in raw-column space, one read
            assert not deflate and v is not None and p_slot is not None and g_slot is not None
            dts, dv = self._put_vec(1, self._centre(ts)), self._put_vec(0, self._centre(v))
            self._true_space(self._slot_col(p_slot))     # p~ -> p in place (the raw-space loading is not needed), g~ -> g
            self.X.loadings_sweep_raw(dts, inv_tt, self._slot_col(p_slot), self._xptr(n * B + B), dv, alpha,
                                      self._slot_col(g_slot), self._xptr(0), self._xptr(n * B), self.r2,
                                      mean=self.mean if self._cen else None)
            if overlap is not None:
                overlap()
            else:
                self.run_deferred()
            t, g, p, _ = self._finish_pass(True, True, True, False)
            return p, t - t.mean(axis=0), g
        if self._lazy_pslot:
            assert p_slot is not None and p_slot[0] == self._lazy_pslot and p_slot[1] == len(self._lazy_ts)
            k = len(self._lazy_ts)
            P = self.slots[self._lazy_pslot]
            p_out = self._slot_col(p_slot)
            if self._istd:
                # raw-space pass: raw-space loadings in their own slot, the true column beside it
                P = self.slots["P~raw"]
                p_out = P.col(k)
                self._true_space(self._slot_col(p_slot))
            self.X.loadings_sweep_lazy(dts, inv_tt, p_out, self._xptr(n * B + B), dv, alpha,
                                       self._slot_col(g_slot), self._xptr(0) if v is not None else None,
                                       self._xptr(n * B) if v is not None else None, P, self.p, k,
                                       self._lazy_coef(0, v) if v is not None else None, self._lazy_coef(1, ts),
                                       self._xptr(n * B + 3 * B) if v is not None else None,
                                       r2=self.r2 if self._istd else None,
                                       mean=self.mean if self._cen else None)
            if overlap is not None:
                overlap()
            else:
                self.run_deferred()
            t, g, p, _ = self._finish_pass(v is not None, v is not None, True, False, nd=(k + 1) if v is not None else 0)
            if v is not None:
                t = self._lazy_fix(t, self._last_d, extra_ts=np.asarray(ts))
                if self._istd:
                    t = t - t.mean(axis=0)
            if deflate:
                self._lazy_ts.append(np.array(ts, dtype=np.float64).ravel())
            return p, t, g
        self.X.deflate_sweep(dts, inv_tt, deflate, self._slot_col(p_slot), self._xptr(n * B + B), dv, alpha,
                             self._slot_col(g_slot), self._xptr(0) if v is not None else None,
                             self._xptr(n * B) if v is not None else None)
        if overlap is not None:
            overlap()
        else:
            self.run_deferred()
        t, g, p, _ = self._finish_pass(v is not None, v is not None, True, False)
        return p, t, g

    # ------------------------------------------------------------------ device-resident NIPALS (one response)
    def deferred_first_pass(self, standardize, v, alpha, g_slot):
        """The first pass of a device-resident fit (statistics + first sweep, only queued) as a callable for
        nipals_resident / unipals_resident: they launch it FIRST and build the queue's device state while it runs
        (queued again when the first step reports offset data)."""
        def first_pass():
            self.prepare(standardize, v=v, alpha=alpha, g_slot=g_slot, defer=True)
        return first_pass

    def resident_supported(self, q, K, norm_kind):
        """The whole fit can be queued without a host round trip: lazy deflation (X is never written, so a wrong
        speculation costs nothing but the skipped launches), one response, shapes the step kernel covers."""
        return (self._lazy_pslot is not None and not self.sparse and norm_kind is not None and q == 1
                and self.nblocks <= _capi.STEP_MAX_BLOCKS and K < self.KD - 1 and self.opts.resident)

    def _res_sweep(self, k, g_slot):
        """sweep with the device-resident vector / coefficients (written by the step kernel); alpha comes from par"""
        n, B = self.n, self.nblocks
        dv, g = self._vec_n[0], self._slot_col(g_slot)
        if self._istd:
            self.X.sweep_lazy(dv, 1.0, g, self._xptr(0), self._xptr(n * B), self.slots["P~raw"], self.p, k,
                              self._coef[0], self._xptr(n * B + 3 * B), r2=self.r2,
                              mean=self.mean if self._cen else None)
        elif k:
            self.X.sweep_lazy(dv, 1.0, g, self._xptr(0), self._xptr(n * B), self.slots[self._lazy_pslot], self.p, k,
                              self._coef[0], self._xptr(n * B + 3 * B))
        else:
            self.X.sweep(dv, 1.0, g, self._xptr(0), self._xptr(n * B))

    def _res_loadings(self, k, last, g_slot):
        """loadings of component k (+ lazy deflation + first sweep of component k+1 unless last), device-resident
        operands"""
        n, B = self.n, self.nblocks
        pslot = self._lazy_pslot
        P = self.slots["P~raw"] if self._istd else self.slots[pslot]
        dv = None if last else self._vec_n[0]
        if self._istd:
            self._true_space(self.slots[pslot].col(k))
        self.X.loadings_sweep_lazy(self._vec_n[1], 1.0, P.col(k), self._xptr(n * B + B), dv, 1.0,
                                   None if last else self._slot_col(g_slot), None if last else self._xptr(0),
                                   None if last else self._xptr(n * B), P, self.p, k, self._coef[0], self._coef[1],
                                   self._xptr(n * B + 3 * B), r2=self.r2 if self._istd else None,
                                   mean=self.mean if self._cen else None)

    def nipals_resident(self, Y, K, norm_kind, tol, wslot, Wn_h, W_h, P_h, first_pass):
        """Queue a complete one-response NIPALS fit (mbpls.py:836-969 for every component) on the device: the first
        pass must already be queued (prepare(..., defer=True)).  Per component: step, sweep, step, loadings (+ lazy
        deflation + next first sweep); the step kernel (csrc/resident.hip) does the n x B algebra and the stop test
        between the passes, RCCL all-reduces the exchange buffer on the same stream when the columns are sharded.
        Finished columns of W~, W and P are copied to the host by a helper thread behind events.  The queue is
        speculative (two iterations per component, which is what one response takes); returns the results of the
        components that converged that way (``k_done`` of them -- the caller resumes from there when k_done < K)."""
        if first_pass is not None:
            first_pass()      # statistics + first sweep: running while the queue's device state is built and uploaded
        with _ResidentQueue(self, Y, K, norm_kind, tol, mode=0) as rq:
            Wn, Pm = self.slots[wslot], self.slots[self._lazy_pslot]
            p8, B = 8 * self.p, self.nblocks
            with self._timed("queue_passes"):
                rq.incremental(wslot)
                rq.step(0, 1, 0)
                for k in range(K):
                    last = k == K - 1
                    self._res_sweep(k, (wslot, k))                   # second iteration (:856, :875)
                    rq.step(k, 2, k)
                    # w~ column k is final; w_b = w~_b / ||w~_b|| (:860) is formed on the host from it and the norms the
                    # step just recorded.  The copies start when the step BEHIND the loadings pass has begun (it says
                    # so in host-mapped memory: no event in the stream -- an event record between two launches idles
                    # the queue for 6 us) -- except for the last component, whose columns leave behind an event
                    # recorded before its loadings pass (they would otherwise wait for the fit's last kernel)
                    nu_copy, unit = rq.unit_blocks_on_host(Wn_h, W_h, k)
                    cols = ([(rq.col(Wn_h, k), Wn.ptr + p8 * k)] + nu_copy
                            + ([(rq.col(P_h, k - 1), Pm.ptr + p8 * (k - 1))] if k else []))
                    if last:
                        rq.stream_out(cols, post=unit, k=k)
                    rq.r_update(k)        # R_[:, k], beta_: extra workgroups of the chunk reduction behind the next pass
                    self._res_loadings(k, last, None if last else (wslot, k + 1))   # (:928, :969) + next first sweep
                    rq.r_flush()
                    if not last:
                        rq.stream_out(cols + rq.r_cols(k), post=unit, k=k, after_next_step=True)
                        rq.step(k + 1, 1, k + 1)
                    else:
                        rq.allreduce()
                rq.last_loadings_norms()
                rq.stream_out([(rq.col(P_h, K - 1), Pm.ptr + p8 * (K - 1))] + rq.r_cols(K - 1, beta=True))
                if not rq.incr:
                    rq.cross(self._lazy_pslot, wslot)
        return rq.result

    def unipals_resident(self, Y, K, wslot, gslot, W_h, P_h, first_pass, Wraw_h=None):
        """The p > n branch of UNIPALS with one response (mbpls.py:481-572) as one device queue: a component costs ONE
        pass (loadings + lazy deflation + the products X^T y, X_b (X^T y)_b of the next component, see
        solvers.fit_unipals); the step kernel forms ts = X (X^T y) / ||.||, v, u, the scalar c with w = c X^T y, the block
        norms and scores between the passes.  No iteration, so nothing is speculative; the offset check of the first
        step is the same as for NIPALS.  ``first_pass``: None when the caller has already queued it."""
        if first_pass is not None:
            first_pass()
        with _ResidentQueue(self, Y, K, 0, 0.0, mode=1) as rq:
            S, Wc, Pm = self.slots[gslot], self.slots[wslot], self.slots[self._lazy_pslot]
            p8 = 8 * self.p
            with self._timed("queue_passes"):
                rq.incremental(wslot)
                rq.step(0, 1, 0)
                for k in range(K):
                    last = k == K - 1
                    # w_raw = X^T u = c X^T y (:503): the column the pass left in S times the scalar the step recorded
                    rq.scale_col(S.ptr, Wc.ptr + p8 * k, k)
                    # (Wraw_h: the caller also wants the raw weight columns -- they are what arrives; W_h is formed from them)
                    src_h = W_h if Wraw_h is None else Wraw_h
                    nu_copy, unit = rq.unit_blocks_on_host(src_h, W_h, k)            # w_b / ||w_b|| (:510-512)
                    rq.stream_out([(rq.col(src_h, k), Wc.ptr + p8 * k)] + nu_copy
                                  + ([(rq.col(P_h, k - 1), Pm.ptr + p8 * (k - 1))] if k else [])
                                  + (rq.r_cols(k - 1) if k else []), post=unit, k=k, after_next_step=not last)
                    rq.r_update(k)
                    self._res_loadings(k, last, None if last else (gslot, 0))        # p (:520), deflation (:535)
                    rq.r_flush()
                    if not last:
                        rq.step(k + 1, 1, k + 1)
                    else:
                        rq.allreduce()
                rq.last_loadings_norms()
                rq.stream_out([(rq.col(P_h, K - 1), Pm.ptr + p8 * (K - 1))] + rq.r_cols(K - 1, beta=True))
                if not rq.incr:
                    rq.cross(self._lazy_pslot, wslot)
        return rq.result

    def xt_v(self, V, alpha, slot, k0=0, want_ssq=False):
        """slot[:, k0:k0+c] = alpha X^T V  (V: n x c host).  Optional per-block column norms^2 (B x c)."""
        V = np.asarray(V, dtype=np.float64)
        if V.ndim == 1:
            V = V.reshape(-1, 1)
        c = V.shape[1]
        dV = self.ctx.from_host(V)
        S = self.slots[slot]
        ssq = self.ctx.empty((self.nblocks, c)) if want_ssq else None
        if self._kscale:
            self.X.xt_v(dV, c, alpha, S.col(k0, c), ssq, mean=self.mean, rscale=self.rs)
        else:
            self.X.xt_v(dV, c, alpha, S.col(k0, c), ssq)
        out = None
        if want_ssq:
            out = self.comm.allreduce_host(ssq.get())
            ssq.free()
        dV.free()
        return out

    def xt_self(self, slot):
        """slot (p x p) = X^T X: the right-hand sides are the columns of X where they lie (leading dimension ld)"""
        S = self.slots[slot]
        self.X.xt_v(self.X.data_ptr, self.p, 1.0, S, None, ldv=self.X.ld, ldg=self.p)

    def xtx_operator(self):
        """w -> (X^T X) w with the p x p matrix left where the matrix cores wrote it: it is wrapped as a p x p
        'matrix' whose x_w product IS the symmetric matrix-vector product (KERNEL n >= p, mbpls.py:586-633).  Column
        shards would need the cross products of different shards: the host form then."""
        if self.comm.is_distributed:
            return super().xtx_operator()
        p = self.p
        ldg = p + (p & 1)
        store = self.ctx.empty(ldg * p, zero=True)
        free = self.ctx.info()["free_bytes"] + self.ctx._pool_bytes
        if p >= 2048 and 8 * self.n * (p + 1) + 3 * 8 * ldg * p < free:
            # X^T X is the association matrix of X^T: the symmetric matrix-core kernel (half the flops, operands reused
            # in 128 x 128 tiles) on a transposed copy, instead of p / 48 many-right-hand-side products that each read
            # all of X (n = p = 20 000: 0.39 s at 41 TFLOP/s of the full product -> 0.12 s)
            Xt = self.X.transposed()
            try:
                Xt.gram(store, ldg)
            finally:
                Xt.close()
        else:
            self.X.xt_v(self.X.data_ptr, p, 1.0, store, None, ldv=self.X.ld, ldg=ldg)
        Vm = _capi.DeviceMatrix(self.ctx, p, p, [0, p], storage=store.ptr)
        self._xtx = (Vm, store)

        def apply(w):
            w = np.ascontiguousarray(np.asarray(w, dtype=np.float64).reshape(p, -1).T)     # c x p: columns contiguous
            c = w.shape[0]
            dW = self.ctx.from_host(w.ravel())
            buf = self.ctx.empty(p * c)
            Vm.x_w(dW, c, buf, ldw=p, ldo=p)
            out = buf.get().reshape(c, p).T.copy()
            buf.free()
            dW.free()
            return out
        return apply

    def _xw(self, Wdev, c, per_block, mean=None, scale=None, mat=None, nan=None, overlap=None):
        """nan: None dense; "skip" missing entries contribute nothing; "indicator" sums of W over the present
        columns of every row (csrc/gemv.hip k_xw NANM).  ``overlap``: host work to do while the kernel runs."""
        if overlap is not None:
            self.ctx.fence()    # (its copies wait for what was queued before this product, not for the product)
        if mat is None and self._kscale and mean is None:
            mean, scale = self.mean, self.scale          # raw working matrix: standardise on the fly
        mat = mat if mat is not None else self.X
        n = mat.n
        nb = mat.nblocks if per_block else 1
        buf = self.comm.exchange_buffer(self.ctx, n * c * nb)
        if nan is None:
            mat.x_w(Wdev, c, buf, per_block=per_block, mean=mean, scale=scale, ldw=mat.p, ldo=n)
        else:
            mat.x_w_nan(Wdev, c, buf, per_block=per_block, mean=mean, scale=scale, indicator=(nan == "indicator"),
                        ldw=mat.p, ldo=n)
        self.comm.allreduce_device(buf)
        if overlap is not None:
            overlap()
        h = buf.get()
        if hasattr(buf, "free"):
            buf.free()
        if per_block:
            return [h[b * n * c:(b + 1) * n * c].reshape(c, n).T.copy() for b in range(nb)]
        return h.reshape(c, n).T.copy()

    def present_per_row(self):
        """number of present (non-NaN) entries of every sample in every block, n x B (one indicator pass)"""
        ones = self.ctx.from_host(np.ones((self.p, 1)))
        cnt = self._xw(ones, 1, True, nan="indicator")
        ones.free()
        return np.column_stack([c[:, 0] for c in cnt])

    def x_w_slot(self, slot, k0, c, per_block=False, overlap=None):
        """X @ slot[:, k0:k0+c]  -> n x c (or one n x c per block)."""
        return self._xw(self.slots[slot].col(k0, c), c, per_block, overlap=overlap)

    def x_w_host(self, W, per_block=False):
        W = np.asarray(W, dtype=np.float64)
        if W.ndim == 1:
            W = W.reshape(-1, 1)
        dW = self.ctx.from_host(W)
        out = self._xw(dW, W.shape[1], per_block)
        dW.free()
        return out

    # ------------------------------------------------------------------ KERNEL method
    def gram(self):
        """A = X X^T on the matrix cores; kept in HBM as an n x n 'matrix' so A @ Z reuses x_w."""
        n = self.n
        ldg = n + (n & 1)
        store = self._gram_store(ldg * n)
        # (the wrapper first: creating a matrix object synchronises the stream -- behind the launch it made the host wait
        # for the whole Gram, 1.9 s at BASELINE config[3], with nothing to do but wait)
        Gm = _capi.DeviceMatrix(self.ctx, n, n, [0, n], storage=store.ptr)
        if self._kscale:
            # (the one-fma form, centre_first=False, measured no faster: the cost is the fifth load per k-step)
            self.X.gram(store.ptr, ldg, mean=self.mean, rscale=self.rs, centre_first=True)
        else:
            self.X.gram(store.ptr, ldg)
        self.comm.allreduce_device(store)        # partial Grams of the column shards (in place, RCCL when sharded)
        self.G = (Gm, store)
        return self

    def _gram_store(self, count):
        """zeroed device buffer for association matrices: owned by the communicator when sharded (RCCL reduces it in
        place), by the context otherwise"""
        if self.comm.is_distributed:
            return self.comm.exchange_buffer(self.ctx, count)
        return self.ctx.empty(count, zero=True)

    def gram_blocks(self):
        """Per-block association matrices A_b = X_b X_b^T, side by side in one n x (B n) 'matrix' whose blocks are the
        B Grams: A_b z for every block is then ONE per-block x_w over B n^2 doubles (multi-response NIPALS iterates
        on these instead of on X)."""
        n, B = self.n, self.nblocks
        ldg = n + (n & 1)
        store = self._gram_store(ldg * n * B)
        for b in range(B):
            view = _capi.DeviceMatrix(self.ctx, n, self.block_sizes[b], None,
                                      storage=self.X.data_ptr + 8 * self.X.ld * self.block_offsets[b])
            view.gram(store.ptr + 8 * ldg * n * b, ldg)
            view.close()
        self.comm.allreduce_device(store)
        Gm = _capi.DeviceMatrix(self.ctx, n, n * B, [b * n for b in range(B + 1)], storage=store.ptr)
        self.Gb = (Gm, store, self.ctx.empty(n * B), self.ctx.empty(n * B))
        # the inner loop only needs the product with the SUM of the Grams (ts ~ P (sum_b A_b) P u): one n x n matrix
        # (134 MB at n = 4096: resident in the 256 MB Infinity Cache) instead of B of them per iteration
        self.Gsum = None
        if B > 1:
            ssum = self.ctx.empty(ldg * n)
            ones = np.ones((B, 1))
            _capi.check(self.ctx.lib.mbpls_hip_right_multiply(self.ctx.handle, ldg * n, store.ptr, ldg * n, B,
                                                              ones.ctypes.data, 1, ssum.ptr, ldg * n))
            self.ctx.sync()                 # `ones` is a host temporary of the asynchronous call
            self.Gsum = (_capi.DeviceMatrix(self.ctx, n, n, [0, n], storage=ssum.ptr), ssum)
        return self

    def gram_apply_blocks(self, z):
        """[A_1 z | ... | A_B z]  (n x B) for a host n-vector z"""
        Gm, _, dW, buf = self.Gb
        n, B = self.n, self.nblocks
        dW.set(np.tile(np.asarray(z, dtype=np.float64).ravel(), B))
        Gm.x_w(dW, 1, buf, per_block=True, ldw=n * B, ldo=n)
        return buf.get().reshape(B, n).T.copy()

    def gram_power_loop(self, Y, Tp, u0, norm_kind, tol, max_iter, batch=16):
        """The multi-response NIPALS inner loop (mbpls.py:841-913) on the device: per iteration one product with the
        Grams and one mbpls_hip_nipals_gram_step, queued ``batch`` at a time; the host only reads the 3-double state
        between batches.  Returns (iterations, capped, ts, u, v, z_used) with z_used = P u of the LAST iteration's
        sweep vector (what the weights of the component are taken with)."""
        Gm = self.Gb[0]
        n, B, k, q = self.n, self.nblocks, Tp.shape[1], Y.shape[1]
        ctx = self.ctx
        if getattr(self, "_xtx", None) is not None:
            self._xtx[0].close()
            self._xtx[1].free()
            self._xtx = None
        if getattr(self, "Gsum", None) is not None:
            Gm, B = self.Gsum[0], 1         # iterate on the summed Gram: one n x n product per iteration
        if getattr(self, "_gp", None) is None or self._gp["q"] != q:
            self._gp = dict(q=q, Y=ctx.from_host(Y), W=[ctx.empty(n * B), ctx.empty(n * B)], a=ctx.empty(n),
                            ts=ctx.empty(n), u=ctx.empty(n), v=ctx.empty(q), state=ctx.empty(3))
        g = self._gp
        dTp = ctx.from_host(Tp) if k else None
        z0 = u0 - Tp.dot(Tp.T.dot(u0))
        g["W"][0].set(np.tile(z0.ravel(), B))
        g["ts"].set(np.zeros(n))
        g["state"].set(np.zeros(3))
        it = 0
        while True:
            for _ in range(batch):
                Gm.x_w(g["W"][it % 2], 1, g["a"], ldw=n * B, ldo=n)
                _capi.check(ctx.lib.mbpls_hip_nipals_gram_step(
                    ctx.handle, n, g["a"].ptr, dTp.ptr if k else None, k, g["Y"].ptr, q, int(norm_kind), float(tol),
                    int(max_iter), g["ts"].ptr, g["u"].ptr, g["v"].ptr, g["W"][(it + 1) % 2].ptr, B, g["state"].ptr))
                it += 1
            st = g["state"].get()
            if st[0] != 0.0:
                break
        iters = int(st[1])
        z_used = g["W"][(iters - 1) % 2].get()[:n].reshape(-1, 1)
        out = (iters, st[0] == 2.0, g["ts"].get().reshape(-1, 1), g["u"].get().reshape(-1, 1),
               g["v"].get().reshape(-1, 1), z_used)
        if dTp is not None:
            dTp.free()
        return out

    def nipals_subspace_loop(self, G, Y, c0, norm_kind, tol, max_iter):
        """The multi-response NIPALS inner loop of one component inside ONE kernel launch (csrc/nipals_step.hip,
        k_nipals_subspace_loop): G = P X (X^T (P Y)) (n x q, host), every sweep vector is u = Y c.  Returns
        (iterations, capped, ts, u, v, c_used) -- c_used: coefficients of the sweep vector of the last iteration.
        With the 2-norm stop rule the loop runs in q x q algebra (k_nipals_subspace_qloop: ~1 us per iteration, any n).
        None when the shape is outside the kernels' range (the caller iterates on the host then)."""
        n, q = G.shape
        qspace = norm_kind == 0 and self.opts.nipals_qspace        # 2-norm: the loop in q x q algebra, any n
        if q > 16 or norm_kind is None or (n > 8192 and not qspace):
            return None
        ctx = self.ctx
        st = getattr(self, "_ss", None)
        if st is None or st["q"] != q:
            # inputs [G | c0] and outputs [state(3) | v(q) | c_used(q) | ts(n) | u(n)] as ONE device array each: one
            # upload and one download per component instead of two and five blocking copies
            st = self._ss = dict(q=q, Y=ctx.from_host(np.asfortranarray(Y)), inp=ctx.empty(n * q + q),
                                 out=ctx.empty(3 + 2 * q + 2 * n), moments=ctx.empty(3 * q * q),
                                 host_in=np.empty(n * q + q))
        hin = st["host_in"]
        hin[:n * q] = np.asarray(G, dtype=np.float64).T.ravel()          # column-major n x q
        hin[n * q:] = np.asarray(c0, dtype=np.float64).ravel()
        st["inp"].set(hin)
        dG, dc0, o = st["inp"].ptr, st["inp"].ptr + 8 * n * q, st["out"].ptr
        dstate, dv, dc, dts, du = o, o + 8 * 3, o + 8 * (3 + q), o + 8 * (3 + 2 * q), o + 8 * (3 + 2 * q + n)
        if qspace:
            _capi.check(ctx.lib.mbpls_hip_nipals_subspace_qloop(ctx.handle, n, q, dG, st["Y"].ptr, dc0, float(tol),
                                                                int(max_iter), dts, du, dv, dc, dstate, st["moments"].ptr))
        else:
            _capi.check(ctx.lib.mbpls_hip_nipals_subspace_loop(ctx.handle, n, q, dG, st["Y"].ptr, dc0, int(norm_kind),
                                                               float(tol), int(max_iter), dts, du, dv, dc, dstate))
        h = st["out"].get()
        state = h[:3]
        return (int(state[1]), state[0] == 2.0, h[3 + 2 * q:3 + 2 * q + n].reshape(-1, 1).copy(),
                h[3 + 2 * q + n:].reshape(-1, 1).copy(), h[3:3 + q].reshape(-1, 1).copy(),
                h[3 + q:3 + 2 * q].reshape(-1, 1).copy())

    def gram_apply(self, Z):
        """A @ Z for a host n x c matrix (A symmetric, so the feature-major product X W is A Z)."""
        Z = np.asarray(Z, dtype=np.float64)
        if Z.ndim == 1:
            Z = Z.reshape(-1, 1)
        Gm = self.G[0]
        dZ = self.ctx.from_host(Z)
        n, c = Z.shape
        buf = self.ctx.empty(n * c)
        Gm.x_w(dZ, c, buf, ldw=n, ldo=n)
        h = buf.get().reshape(c, n).T.copy()
        buf.free()
        dZ.free()
        return h

    def gram_host(self):
        n = self.n
        ldg = n + (n & 1)
        return np.asarray(self.G[1].get()).reshape(n, ldg).T[:n, :n]


# The p x K attributes of a fit at BASELINE size are four or five arrays of 80 MB.  A caller that fits in a loop lets the
# previous model go with every new one: 20-26 ms of munmap for 55 ms of fit (and as many page faults again for the next
# fit's arrays, which the helper thread of host_array has to hide).  Large host buffers are therefore handed out as views
# of owners the context keeps: an owner nobody else refers to any more -- every attribute, slice and view of it gone,
# numpy collapses the base chain onto the owner -- is handed out again, pages mapped.  Bounded (HOST_POOL_BYTES), released
# by Context.trim_pool().
HOST_POOL_MIN = 4 << 20
HOST_POOL_BYTES = 2 << 30


def _pooled_host_array(ctx, shape):
    """(F-ordered float64 array of ``shape``, True when its pages are mapped already)"""
    nbytes = 8 * math.prod(shape)                       # (np.prod of a tuple costs 4 us, a small fit asks ten times)
    if nbytes < HOST_POOL_MIN:
        return np.empty(shape, order="F"), False
    pool = ctx.__dict__.setdefault("_host_pool", [])     # (a Context has the attribute; the tests pass a bare object)
    raw, mapped = None, False
    for cand in pool:
        # references to an owner at rest: the list, `cand`, the argument of getrefcount
        if nbytes <= cand.nbytes <= nbytes + (nbytes >> 2) and sys.getrefcount(cand) == 3:
            raw, mapped = cand, True
            break
    if raw is None:
        raw = np.empty(nbytes, dtype=np.uint8)
        held = sum(r.nbytes for r in pool)
        i = 0
        while held + nbytes > HOST_POOL_BYTES and i < len(pool):
            # a full pool of other sizes (the caller has moved on to another problem): owners at rest go, oldest first
            cand = pool[i]
            if sys.getrefcount(cand) == 3:
                held -= cand.nbytes
                del pool[i]
            else:
                i += 1
        cand = None
        if held + nbytes <= HOST_POOL_BYTES:
            pool.append(raw)
    return raw[:nbytes].view(np.float64).reshape(shape, order="F"), mapped


class _Keep:
    """a flat float64 view over several adjacent host arrays, for one merged copy: keeps them alive, looks like an
    array to ``Context.copier_submit``"""

    def __init__(self, view, keeps):
        self.view, self.keeps = view, keeps
        self.ctypes, self.nbytes = view.ctypes, view.nbytes


_PREFAULT = {}


def _prefault(arr, min_bytes=1 << 20, tail_bytes=0, only_tail=False):
    """Map the pages of a freshly allocated host array in a helper thread (see HipEngine.host_array).  Linux
    madvise(MADV_POPULATE_WRITE) populates writable pages WITHOUT touching their contents, so it may run beside the
    copy that fills the array; the call releases the GIL.  Where it is not available nothing happens (the copy then
    faults the pages in itself)."""
    if arr.nbytes < min_bytes or _PREFAULT.get("off"):
        return
    if "env" not in _PREFAULT:
        import os
        _PREFAULT["env"] = True
        if os.environ.get("MBPLS_NO_PREFAULT") == "1":
            _PREFAULT["off"] = True
            return
    import ctypes
    import queue
    import threading
    q = _PREFAULT.get("queue")
    if q is None or not _PREFAULT["thread"].is_alive():
        try:
            libc = ctypes.CDLL(None, use_errno=True)
            madvise = libc.madvise
            madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
            madvise.restype = ctypes.c_int
        except Exception:
            _PREFAULT["off"] = True
            return
        q = queue.Queue()

        def work():
            import mmap
            page = mmap.PAGESIZE            # (64 KB pages elsewhere: a hard-coded 4096 made every call fail with EINVAL)
            while True:
                # the ARRAY travels with its address: it stays alive until its pages are mapped (an array collected
                # before the thread got to it left madvise on an unmapped or reused range -- round-3 advisor)
                keep, tail, only_tail = q.get()
                addr, nbytes = keep.ctypes.data, keep.nbytes
                lo = (addr + page - 1) // page * page
                hi = (addr + nbytes) // page * page
                # in small pieces: the call holds the process's mmap lock while it populates, and every allocation /
                # release of a large array on another thread (np.empty -> mmap) waits for it -- 8 MB pieces made a
                # 3 MB np.empty on the fitting thread take 0.35 ms
                step = 512 << 10
                # the tail of the array first (its last column arrives behind the last pass of a fit: a copy into
                # unmapped pages there costs 0.25 us per page with nothing to hide it -- 3 MB took 300 us instead of 90)
                mid = max(lo, (hi - min(tail, hi - lo)) // page * page) if tail else hi
                for a, b in ((mid, hi),) if only_tail else ((mid, hi), (lo, mid)):
                    while a < b and not _PREFAULT.get("off"):
                        n = min(step, b - a)
                        if madvise(a, n, 23) != 0:     # MADV_POPULATE_WRITE (Linux >= 5.14)
                            if ctypes.get_errno() == 22:    # EINVAL: not supported here
                                _PREFAULT["off"] = True
                            break
                        a += n
                keep = None

        th = threading.Thread(target=work, daemon=True, name="mbpls-prefault")
        th.start()
        _PREFAULT.update(queue=q, thread=th)
    q.put((arr, int(tail_bytes), bool(only_tail)))


class _ResidentQueue:
    """State of one device-resident fit (csrc/resident.hip): the device buffers the step kernel works on, the helper
    thread that copies finished columns to the host behind events, and the read-back of the results.  Used as a context
    manager around the queueing of the passes; ``result`` is available afterwards."""

    def __init__(self, eng, Y, K, norm_kind, tol, mode):
        self.eng, self.K, self.mode, self.istd = eng, K, mode, bool(eng._istd)
        n, B, q = eng.n, eng.nblocks, Y.shape[1]
        ctx = self.ctx = eng.ctx
        self.lib = ctx.lib
        # ONE device allocation, ONE upload, ONE download: [u_cur | Y | ts_old | par | scal | iters | ssq_x | C | V | Ts |
        # U | Tb | Tsc].  Everything up to V is uploaded (zeros where the step expects them); par .. Tb come back in one
        # copy.
        sizes = [("u_cur", n), ("Y", n * q), ("ts_old", n), ("par", 8), ("scal", K * 4 * B), ("iters", K), ("ssqx", B),
                 ("C", K * K), ("V", q * K), ("Ts", n * K), ("U", n * K), ("Tb", n * B * K)] \
            + ([("Tsc", n * K)] if self.istd else []) + [("M", K * K), ("MV", K * q)]
        self.crossed = False
        self.finalised = None
        self.incr = False
        self.off, o = {}, 0
        for key, sz in sizes:
            self.off[key] = (o, sz)
            o += sz
        self.state = ctx.empty(o)
        up = np.zeros(self.off["V"][0])
        up[:n] = Y[:, 0]
        up[n:n + n * q] = np.asfortranarray(Y).ravel(order="F")
        _capi.check(self.lib.mbpls_hip_memcpy_h2d(ctx.handle, self.state.ptr, up.ctypes.data, up.nbytes))
        self._up = up                               # (the copy is stream-ordered: keep the source alive)
        ptr = {key: self.state.ptr + 8 * o for key, (o, _) in self.off.items()}
        if not self.istd:
            ptr["Tsc"] = ptr["Ts"]
        self.ptr = ptr
        # raw-space passes without centring are only good for modestly offset data: the first step checks the measure
        # the statistics pass left behind and stops the queue when the centring flavour is needed
        if self.istd and not eng._cen:
            hint = eng._offset_hint()
            if hint is None and not eng.comm.is_distributed and 8 * n * eng.p >= eng.OFFSET_ASK_BYTES:
                # nothing known about this matrix and a pass takes longer than this queue takes to build: the first
                # pass (already under way) is waited for and asked, instead of a queue that may have to be thrown away
                hint = eng._measured_offset() > eng.CENTRE_FROM_OFFSET ** 2
                eng._set_offset_hint(hint)
            if hint:
                eng._cen = True                     # (the last fit of this matrix / shape needed centring passes)
        amp2 = eng.CENTRE_FROM_OFFSET ** 2 if (self.istd and not eng._cen) else 0.0
        self.st = _capi.NipalsStep(n=n, nblocks=B, q=q, centre=int(self.istd), norm_kind=int(norm_kind), tol=float(tol),
                                   amp2_max=amp2, mode=int(mode), xchg=eng._xchg.ptr, Y=ptr["Y"], Ts=ptr["Ts"],
                                   Tsc=ptr["Tsc"], U=ptr["U"], V=ptr["V"], Tb=ptr["Tb"], scal=ptr["scal"],
                                   ssq_x=ptr["ssqx"], iters=ptr["iters"], u_cur=ptr["u_cur"],
                                   ts_old=ptr["ts_old"], vpass=eng._vec_n[0].ptr, tspass=eng._vec_n[1].ptr,
                                   cu=eng._coef[0].ptr, ce=eng._coef[1].ptr, par=ptr["par"])
        # block norms of every recorded component, written by the step straight into pinned host memory: the helper
        # thread that forms w_b = w~_b / ||w~_b|| reads them there (a copy of B doubles into pageable memory is a blit
        # kernel that waits behind -- and slows -- the running pass: 180 us each)
        blk_h, blk_d = ctx.mapped_scalars(K * B + 8)
        self.nu2_h, nu2_d = blk_h[8:8 + K * B], blk_d + 64
        self.st.nu2_host = nu2_d
        # ... and the number of the step launch that has started, stored there by every step first of all: the copier
        # thread starts a job when the step queued BEHIND the job's data has begun, instead of waiting for an event
        # recorded on the stream (an event record between two launches idles the queue for ~6 us: 60 us of a fit)
        # The word has a FIXED place in the context's block and its values never decrease over the life of the context
        # (``ctx._seq_base`` + step number): nothing a fit leaves behind can release a job of the next one early.
        self.seq_h = blk_h[0:1]
        self.seq0 = float(getattr(ctx, "_seq_base", 0.0))
        self.seq_h[0] = self.seq0                   # (what the last step of the previous queue stored; new block: unset)
        self.st.host_seq = blk_d
        self.nsteps = 0
        self.ticket = 0                             # of the last job handed to the copier thread
        self.scores_ticket = 0                      # of the job that carries the last component's n-vectors
        # host image of [par .. Tb]: the n-vectors arrive per component through the copier, the scalars at the end
        self.h = np.empty(self.off["Tb"][0] + self.off["Tb"][1] - self.off["par"][0])
        self.result = None
        # SMALL problems (the reference's own data sets: tens to hundreds of samples, thousands of features): a copy
        # costs the copier ~30 us whatever its size, and a fit hands out six columns per component -- 36 copies made the
        # fitting thread wait 1.1 ms for the copier behind a queue the GPU had finished in 0.45 ms (n = 150, p = 3000,
        # K = 5: tools/host_timeline.py).  Below BULK_BYTES per attribute the columns are not streamed: ONE job behind
        # the queue copies every array whole (adjacent columns merged into one copy) and the block normalisation of
        # the weights is a few numpy divisions afterwards.
        self.bulk = 8 * max(eng.p, n * max(B, 1)) * K <= self.BULK_BYTES
        self._bulk_copies, self._bulk_posts = [], []

    BULK_BYTES = 1 << 20

    @staticmethod
    def col(arr, k):
        c = arr[:, k]
        assert c.flags.c_contiguous
        return c

    @staticmethod
    def _merged(copies):
        """adjacent (host view, device address) pairs -- columns k, k + 1 of the same arrays -- as one copy each"""
        items = sorted(((int(src), dst.ctypes.data, dst.nbytes, dst) for dst, src in copies if dst.nbytes), key=lambda c: c[0])
        out = []
        for src, haddr, nb, keep in items:
            if out and out[-1][0] + out[-1][2] == src and out[-1][1] + out[-1][2] == haddr:
                out[-1][2] += nb
                out[-1][3].append(keep)
            else:
                out.append([src, haddr, nb, [keep]])
        merged = []
        for src, haddr, nb, keeps in out:
            view = np.ctypeslib.as_array((_capi.C.c_double * (nb // 8)).from_address(haddr))
            view = view.view()
            merged.append((_Keep(view, keeps), src))
        return merged

    def _submit_bulk(self):
        """small problems: everything stream_out was asked for, as one job behind the whole queue"""
        if self._bulk_copies:
            self.ticket = self.scores_ticket = self.ctx.copier_submit(self._merged(self._bulk_copies))
            self._bulk_copies = []

    def stream_out(self, copies, post=None, k=None, after_next_step=False):
        """copy finished device columns to the host (the context's copier thread, csrc/copier.hip) behind everything
        queued on the compute stream so far; ``post``: the block normalisation the thread does on what arrived
        (``unit_blocks_on_host``); ``k``: the n-vectors the step recorded for component k (super score, u, block
        scores) leave with this job as well -- the read-back behind the last pass is then a few hundred bytes instead
        of n (2 + B) K doubles (1 MB at n = 4096, K = 10: 90 us of every fit); ``after_next_step``: the caller launches
        a step right after this call -- the job waits for THAT step to start (``host_seq``) instead of an event"""
        if k is not None:
            n, B, o0 = self.eng.n, self.eng.nblocks, self.off["par"][0]
            for key, width in (("Ts", 1), ("U", 1), ("Tb", B)):
                a = self.off[key][0] - o0 + n * width * k
                copies = copies + [(self.h[a:a + n * width], self.ptr[key] + 8 * n * width * k)]
        if self.bulk:
            self._bulk_copies += copies
            if post is not None:
                self._bulk_posts.append(post)
            return
        flag = (self.seq_h.ctypes.data, self.seq0 + self.nsteps + 1) if after_next_step else None
        self.ticket = self.ctx.copier_submit(copies, scale=post, flag=flag)
        if k is not None and k == self.K - 1:
            self.scores_ticket = self.ticket

    def unit_blocks_on_host(self, src_h, dst_h, k):
        """(copies, post) for stream_out: w_b = w~_b / ||w~_b|| (mbpls.py:860 / :510-512) for column k, formed by the
        copier thread from the column it has just copied (``src_h[:, k]``, may be ``dst_h`` itself) and the block norms
        the step of component k left in host-mapped memory -- a p-sized kernel between two passes costs the queue more
        than the division costs a thread that is waiting anyway (on a side stream it ran 60x slower and slowed the
        pass: the fat pass workgroups leave no registers for a second kernel's waves)."""
        B = self.eng.nblocks
        offs = getattr(self, "_offs64", None)
        if offs is None:
            offs = self._offs64 = np.ascontiguousarray(self.eng.block_offsets, dtype=np.int64)
        return [], (self.col(src_h, k), self.col(dst_h, k), offs, self.nu2_h.ctypes.data + 8 * k * B)

    # ---- R_ / beta_ column by column while the fit runs
    def incremental(self, wslot):
        """R_ = W (P^T W)^+ and beta_ = R_ V^T (mbpls.py:988-989 / :569-570) without a p x K product and a p x K copy
        BEHIND the last pass (at p = 10^6, K = 10: 88 MB over PCIe, 2.8 ms of a 106 ms fit; 0.35 of the 1.5 ms a
        shard-sized fit spends outside its passes).  C = P^T W~ is upper triangular -- X_i w~_k = 0 for i > k: every
        deflation removes the score direction X_k w~_k -- and the column scaling of W cancels, so R C = W~ yields
        column k from the columns before it as soon as column k of C is known.  That column costs nothing: above the
        diagonal it is the lazy-deflation sums p_i^T w~_k the sweep pass delivers anyway, on the diagonal
        ts^T (X_k w~_k), a dot product of vectors the step has in registers (``mbpls_nipals_step.C``).  The p-sized
        update rides as extra workgroups on the chunk reduction of the next pass (``mbpls_hip_queue_r_update``: no
        launch of its own), and the finished column leaves with W~'s and P's.  The conditioning of C is checked when
        the queue returns; an exhausted matrix (where the reference's pinv truncates) takes the product route."""
        eng, K = self.eng, self.K
        if not eng.opts.r_incremental or self.st.q != 1:
            return
        self.st.C, self.st.ldc = self.ptr["C"], K
        self.Rd, self.Bd = eng.slot("R", K, zero=False), eng.slot("beta", 1, zero=False)
        self.R_h = eng._host_for_slot("R", (eng.p, K))
        self.beta_h = eng._host_for_slot("beta", (eng.p, 1))
        self.wslot = wslot
        self.incr = True

    def r_update(self, k):
        if self.incr:
            eng, K, p = self.eng, self.K, self.eng.p
            W = eng.slots[self.wslot]
            _capi.check(self.lib.mbpls_hip_queue_r_update(self.ctx.handle, p, W.ptr + 8 * p * k, self.Rd.ptr, p, k,
                                                          self.ptr["C"] + 8 * K * k, self.ptr["V"] + 8 * k, self.Bd.ptr))

    def r_flush(self):
        """the queued update goes by itself if the pass had no chunk reduction to carry it"""
        if self.incr:
            _capi.check(self.lib.mbpls_hip_flush_r_update(self.ctx.handle))

    def r_cols(self, k, beta=False):
        if not self.incr:
            return []
        return ([(self.col(self.R_h, k), self.Rd.ptr + 8 * self.eng.p * k)]
                + ([(self.col(self.beta_h, 0), self.Bd.ptr)] if beta else []))

    # ---- launches
    def allreduce(self):
        if self.eng.comm.is_distributed:
            self.eng.comm.allreduce_device(self.eng._xchg)

    def step(self, kc, run, nd):
        self.allreduce()
        self.nsteps += 1
        self.st.kc, self.st.run, self.st.nd, self.st.seq_value = kc, run, nd, self.seq0 + self.nsteps
        _capi.check(self.lib.mbpls_hip_nipals_step(self.ctx.handle, self.st))

    def last_loadings_norms(self):
        """||p_b||^2 of the last (loadings-only) pass: B values of the exchange buffer, moved into the results block on
        the device (the step of the next component records them otherwise)"""
        eng, B = self.eng, self.eng.nblocks
        _capi.check(self.lib.mbpls_hip_memcpy_d2d(self.ctx.handle, self.ptr["scal"] + 8 * (4 * B * (self.K - 1) + 2 * B),
                                                  eng._xptr(eng.n * B + B), 8 * B))

    def cross(self, pslot, wslot):
        """C = P^T W~ (K x K, mbpls.py:988) queued behind the last pass, summed over the column shards on the device:
        it comes back with the other results instead of costing a host round trip of its own after the queue"""
        eng, K = self.eng, self.K
        if K > 32 or not getattr(eng.comm, "device_views_ok", False):
            return
        A, Bm = eng.slots[pslot], eng.slots[wslot]
        dC = _capi.DeviceArray(self.ctx, (K * K,), ptr=self.ptr["C"])
        _capi.check(self.lib.mbpls_hip_cross_product_dev(self.ctx.handle, eng.p, A.ptr, eng.p, K, Bm.ptr, eng.p, K, dC.ptr))
        if eng.comm.is_distributed:
            eng.comm.allreduce_device(dC)
        self.crossed = (pslot, wslot)
        # ... and the rest of the end-of-fit algebra right behind it, all on the device: M = D (C D)^+ (one-wave Jacobi
        # SVD, numpy.linalg.pinv's cutoff), R_ = W~ M, beta_ = W~ M V^T (:988-989 / :569-570); their copies start in the
        # helper thread as soon as the products have run -- no host round trip between the last pass and the attributes
        q = self.st.q
        _capi.check(self.lib.mbpls_hip_finalise_small(self.ctx.handle, K, eng.nblocks, q, self.ptr["C"], self.ptr["scal"],
                                                      self.ptr["V"], self.ptr["M"], self.ptr["MV"]))
        R, Bt = eng.slot("R", K, zero=False), eng.slot("beta", q, zero=False)
        _capi.check(self.lib.mbpls_hip_right_multiply_dev(self.ctx.handle, eng.p, Bm.ptr, eng.p, K, self.ptr["M"], K,
                                                          R.ptr, eng.p))
        _capi.check(self.lib.mbpls_hip_right_multiply_dev(self.ctx.handle, eng.p, Bm.ptr, eng.p, K, self.ptr["MV"], q,
                                                          Bt.ptr, eng.p))
        self.finalised = (eng.fetch_async("R")[1], eng.fetch_async("beta")[1])

    def scale_col(self, src, dst, k):
        """dst = src * c_k with the scalar the step of component k left at scal[k][3][0]"""
        B = self.eng.nblocks
        _capi.check(self.lib.mbpls_hip_scale_col_by(self.ctx.handle, self.eng.p, src, dst,
                                                    self.ptr["scal"] + 8 * (4 * B * k + 3 * B)))

    # ---- context manager
    def __enter__(self):
        _capi.check(self.lib.mbpls_hip_set_pass_params(self.ctx.handle, self.ptr["par"]))
        return self

    def _view(self, h, key, shape=None):
        o, sz = self.off[key]
        o -= self.off["par"][0]
        a = h[o:o + sz]
        return a if shape is None else a.reshape(shape, order="F")

    def __exit__(self, et, ev, tb):
        eng, K = self.eng, self.K
        n, B, q = eng.n, eng.nblocks, self.st.q
        h = None
        try:
            if et is None:
                self._submit_bulk()
                eng.run_deferred(all_of_it=True)          # host chores while the GPU works through the queue
                with eng._timed("wait_kernel+d2h_small"):
                    # par .. V in ONE copy (waits for the queue); Ts, U, Tb have left column by column (stream_out)
                    o0, o1 = self.off["par"][0], self.off["V"][0] + self.off["V"][1]
                    h = self.h
                    _capi.check(self.lib.mbpls_hip_memcpy_d2h(self.ctx.handle, h.ctypes.data, self.state.ptr + 8 * o0,
                                                              8 * (o1 - o0)))
        finally:
            self.lib.mbpls_hip_set_pass_params(self.ctx.handle, None)
            ctx, ticket = self.ctx, self.ticket
            ctx._seq_base = self.seq0 + self.nsteps

            posts, nu2_h, nB = self._bulk_posts, self.nu2_h, eng.nblocks

            def wait_copies(meaningful=True):   # every column handed to the copier thread by this queue has landed
                ctx.copier_wait(ticket)
                if not meaningful:          # a queue that gave up early copied zeros: nothing to normalise (0 / 0)
                    posts.clear()
                while posts and ticket:     # (small problems: w_b = w~_b / ||w~_b|| here, not in the copier thread)
                    src, dst, offs, addr = posts.pop()
                    k = (int(addr) - nu2_h.ctypes.data) // (8 * nB)
                    nu = np.sqrt(nu2_h[k * nB:(k + 1) * nB])
                    for b in range(nB):
                        np.divide(src[offs[b]:offs[b + 1]], nu[b], out=dst[offs[b]:offs[b + 1]])

            scores_ticket = self.scores_ticket or ticket

            def wait_scores():              # ... the n-vectors of every component (on their way before the last pass)
                ctx.copier_wait(scores_ticket)

        if et is not None or h is None:
            # what was queued before the error may still be running: let it finish before the buffers it works on go
            # back to the pool and before the flag is touched (a step of THIS queue storing its number later on would
            # otherwise race with the release below)
            self.lib.mbpls_hip_sync(self.ctx.handle)
            self.seq_h[0] = np.inf          # (a job may be waiting for a step that was never launched)
            try:
                wait_copies(meaningful=False)
            except _capi.HipError:
                pass                        # (the error that brought us here is the one to report)
            self.seq_h[0] = ctx._seq_base
            self.state.free()
            return False
        par = self._view(h, "par")
        if eng.comm.is_distributed and eng._cluster_guarded():
            # column shards: a hand-off that gave up on ANY rank sends every rank up to MBPLS._fit (which runs the fit
            # again on the two-phase passes) -- one small all-reduce per queue, long columns only
            try:
                eng.check_cluster_passes()
            except _capi.ClusterTimeout:
                wait_copies(meaningful=False)
                self.state.free()
                raise
        if par[7] != 0.0:
            # a grid barrier of the multi-workgroup step gave up (cooperative launch: should never happen) -- nothing
            # was written to X, so the caller simply runs the fit with its host-driven loop.  (A hand-off of a cluster
            # pass that gave up sets the same flag: that one goes up to MBPLS._fit, which runs the fit again on the
            # two-phase passes.)
            wait_copies(meaningful=False)
            self.state.free()
            eng.check_cluster_passes()
            self.result = dict(k_done=0, reason="barrier")
            return False
        if par[2] != 0.0 and par[5] == 1.0:
            # offset data (max |mean|/scale = sqrt(par[6])): the caller queues again, now with centring passes
            eng._cen = True
            eng._set_offset_hint(True)
            wait_copies(meaningful=False)
            self.state.free()
            self.result = dict(k_done=0, reason="offset", offset=float(np.sqrt(par[6])))
            return False
        k_done = K if par[2] == 0.0 else int(par[3])
        if self.istd:                                # (what the statistics pass measured: the next fit's flavour)
            eng._set_offset_hint(float(par[6]) > eng.CENTRE_FROM_OFFSET ** 2)
        scal = self._view(h, "scal").reshape(K, 4, B)
        if self.incr and k_done == K:
            # R_ / beta_ came column by column from R C = W~ with the triangular C the steps recorded: good as long as
            # C diag(1 / ||w~_k||) is comfortably invertible (numpy's pinv of :988 only departs from the inverse when
            # singular values fall below 1e-15 of the largest: an exhausted matrix); otherwise the product route
            Cm = self._view(h, "C").reshape((K, K), order="F")
            try:
                sv = np.linalg.svd(Cm / np.sqrt(scal[:, 0, :].sum(axis=1)), compute_uv=False)
            except np.linalg.LinAlgError:                 # (not finite)
                sv = np.zeros(2)
            if sv[-1] > 1e-10 * sv[0]:                    # (False for NaN as well)
                R_h, beta_h = self.R_h, self.beta_h
                self.finalised = (lambda: (wait_copies(), R_h)[1], lambda: beta_h)
        # the attribute columns the helper thread is still copying / scaling must have landed before they are used:
        # the caller waits right before it hands them out (wait_copies), not here
        # (R_ / beta_ were computed and sent on their way by the queue itself: cross())
        finalised = self.finalised if (self.finalised and k_done == K) else None
        if self.finalised and finalised is None:
            for w in self.finalised:                  # a failed speculation: let the (meaningless) copies land, drop them
                w()
        self.result = dict(k_done=k_done, wait_copies=wait_copies, wait_scores=wait_scores, finalised=finalised,
                           r_route=("column by column during the fit" if (self.incr and finalised) else
                                    "one product behind the last pass"), Ts=self._view(h, "Ts", (n, K)), U=self._view(h, "U", (n, K)),
                           V=self._view(h, "V", (q, K)), Tb=self._view(h, "Tb").reshape(K, B, n), scal=scal,
                           ssq_x=self._view(h, "ssqx").copy(), iters=self._view(h, "iters"), last_diff=float(par[4]),
                           PtW=self._view(h, "C").reshape(K, K).copy() if (self.crossed and k_done == K) else None)
        if k_done < K:
            # hand the lazy-deflation state to the host-driven loop
            Tc = _capi.DeviceArray(self.ctx, (n, K), ptr=self.ptr["Tsc"]).get()
            eng._lazy_ts = [np.ascontiguousarray(Tc[:, i]) for i in range(k_done)]
        # the copier reads Ts / U / Tb columns out of the state block: the last of those jobs must have run before the
        # block goes back to the pool (the next allocation of that size may overwrite it).  The caller needs the scores
        # at once anyway, so the wait is not an extra one.
        wait_scores()
        self.state.free()
        return False
