"""TEST-ONLY stand-in for ``mbpls_amd.engine.HipEngine`` with numpy kernels.

Lets the host drivers (``mbpls_amd/solvers.py``) and the column-sharding logic be exercised on a
machine without a GPU (``-m "not gpu"`` tests, world_size-2 gloo tests).  It implements the
kernel-level semantics of include/mbpls_hip.h, nothing more.  The product never imports this:
``MBPLS`` always builds a ``HipEngine`` and fails loudly without the HIP library.
"""
import numpy as np

from mbpls_amd.comm import LocalComm
from mbpls_amd.engine_base import EngineBase

_EPS = np.finfo(np.float64).eps


class NumpyEngine(EngineBase):
    def __init__(self, comm=None):
        self._init_base()
        self.comm = comm if comm is not None else LocalComm()
        self.slots = {}
        self.mean = self.scale = None
        self.G = None

    def load_host_blocks(self, blocks):
        self.Xm = np.hstack([np.asarray(b, dtype=np.float64) for b in blocks]).copy()
        self.n, self.p = self.Xm.shape
        self.nblocks = len(blocks)
        self.block_offsets = [0] + list(np.cumsum([b.shape[1] for b in blocks]))
        self.block_sizes = [b.shape[1] for b in blocks]
        return self

    def close(self):
        pass

    # slots
    def slot(self, name, K):
        if name not in self.slots or self.slots[name].shape != (self.p, K):
            self.slots[name] = np.zeros((self.p, K))
        return self.slots[name]

    def fetch_col(self, name, k, out):
        out[:, k] = self.slots[name][:, k]

    def fetch(self, name):
        return self.slots[name].copy()

    def put(self, name, arr):
        arr = np.asarray(arr, dtype=np.float64)
        if arr.ndim == 1:
            arr = arr.reshape(-1, 1)
        self.slots[name] = arr.copy()

    def cross(self, a, b, rows=None):
        (na, ka, ca), (nb, kb, cb) = a, b
        r0, r1 = rows if rows is not None else (0, self.p)
        return self.comm.allreduce_host(self.slots[na][r0:r1, ka:ka + ca].T.dot(self.slots[nb][r0:r1, kb:kb + cb]))

    def right_mul(self, a, M, c, rows=None):
        (na, ka, c1), (nc, kc) = a, c
        M = np.asarray(M, dtype=np.float64)
        r0, r1 = rows if rows is not None else (0, self.p)
        self.slots[nc][r0:r1, kc:kc + M.shape[1]] = self.slots[na][r0:r1, ka:ka + c1].dot(M)

    def put_col(self, name, k, vec):
        self.slots[name][:, k] = np.asarray(vec).ravel()

    def _ranges(self):
        return list(zip(self.block_offsets[:-1], self.block_offsets[1:]))

    # ---- sparse_data=True: the masked sums of csrc/colpass_nan.hip (mbpls.py:846-854, :865-873, :919-926, :969)
    def _nan_sweep_core(self, v, g_slot):
        X = self.Xm
        pres = ~np.isnan(X)
        X0 = np.where(pres, X, 0.0)
        v = np.asarray(v, dtype=np.float64).reshape(-1, 1)
        num, den = X0.T.dot(v).ravel(), pres.T.astype(np.float64).dot(v ** 2).ravel()
        g = np.where(den != 0.0, num / np.where(den == 0.0, 1.0, den), 0.0)
        if g_slot is not None:
            self.slots[g_slot[0]][:, g_slot[1]] = g
        A = np.column_stack([X0[:, s:e].dot(g[s:e]) for s, e in self._ranges()])
        D = np.column_stack([pres[:, s:e].astype(np.float64).dot(g[s:e] ** 2) for s, e in self._ranges()])
        ssq = np.array([(g[s:e] ** 2).sum() for s, e in self._ranges()])
        A, D, ssq = self.comm.allreduce_host(A), self.comm.allreduce_host(D), self.comm.allreduce_host(ssq)
        with np.errstate(divide="ignore", invalid="ignore"):
            t = np.where(D > 0.0, A * (ssq[None, :] / D), 0.0)       # nu^2 A_i / D_i (engine._finish_pass)
        return t, ssq

    def _nan_prepare(self, standardize):
        X = self.Xm
        rm = getattr(self, "rowmask", None)
        inc = np.ones(X.shape[0], dtype=bool) if rm is None else (np.asarray(rm).ravel() != 0)
        X = np.where(inc[:, None], X, 0.0)                           # an excluded sample: present, and zero
        use = ~np.isnan(X) & inc[:, None]
        if standardize:
            cnt = use.sum(axis=0).astype(np.float64)
            mu = np.where(use, X, 0.0).sum(axis=0) / cnt
            d = np.where(use, X - mu, 0.0)
            var = ((d ** 2).sum(axis=0) - d.sum(axis=0) ** 2 / cnt) / cnt
            sc = np.sqrt(var)
            sc[var <= cnt * _EPS * var + (cnt * mu * _EPS) ** 2] = 1.0
            X = np.where(inc[:, None], X - mu, 0.0) / sc             # NaN stays NaN

            class _A:
                def __init__(s, a): s.a = a
                def get(s): return s.a
            self.mean, self.scale, self.count = _A(mu), _A(sc), _A(cnt)
        self.Xm = X
        return self.comm.allreduce_host(np.array([np.nansum(X[:, s:e] ** 2) for s, e in self._ranges()]))

    def _sweep_core(self, v, alpha, g_slot):
        if self.sparse:
            return self._nan_sweep_core(v, g_slot)
        g = alpha * self.Xm.T.dot(v).ravel()
        if g_slot is not None:
            self.slots[g_slot[0]][:, g_slot[1]] = g
        t = np.column_stack([self.Xm[:, s:e].dot(g[s:e]) for s, e in self._ranges()])
        ssq = np.array([(g[s:e] ** 2).sum() for s, e in self._ranges()])
        return self.comm.allreduce_host(t), self.comm.allreduce_host(ssq)

    def prepare(self, standardize, v=None, alpha=1.0, g_slot=None, mean=None, scale=None):
        if self.sparse:
            ssq_x = self._nan_prepare(standardize)
            if v is None:
                return ssq_x, None, None
            t, g = self._nan_sweep_core(v, g_slot)
            return ssq_x, t, g
        X = self.Xm
        rm = getattr(self, "rowmask", None)
        inc = None if rm is None else (np.asarray(rm).ravel() != 0)
        if inc is not None:
            X = np.where(inc[:, None], X, 0.0)
            self.Xm = X
        if standardize:
            n = X.shape[0] if inc is None else int(inc.sum())
            mu = X.sum(axis=0) / n
            d = X - mu if inc is None else np.where(inc[:, None], X - mu, 0.0)
            var = ((d ** 2).sum(axis=0) - d.sum(axis=0) ** 2 / n) / n
            sc = np.sqrt(var)
            sc[var <= n * _EPS * var + (n * mu * _EPS) ** 2] = 1.0
            self.Xm = d / sc

            class _A:  # quacks like DeviceArray.get()
                def __init__(s, a): s.a = a
                def get(s): return s.a
            self.mean, self.scale = _A(mu), _A(sc)
        ssq_x = self.comm.allreduce_host(np.array([(self.Xm[:, s:e] ** 2).sum() for s, e in self._ranges()]))
        if v is None:
            return ssq_x, None, None
        t, g = self._sweep_core(v, alpha, g_slot)
        return ssq_x, t, g

    def sweep(self, v, alpha, g_slot, overlap=None):
        r = self._sweep_core(v, alpha, g_slot)
        if overlap is not None:
            overlap()
        return r

    def deflate_sweep(self, ts, inv_tt, deflate, p_slot, v=None, alpha=1.0, g_slot=None, overlap=None):
        if overlap is not None:
            overlap()
        if self.sparse:
            pres = ~np.isnan(self.Xm)
            tsc = np.asarray(ts, dtype=np.float64).reshape(-1, 1)
            num, den = np.where(pres, self.Xm, 0.0).T.dot(tsc).ravel(), pres.T.astype(np.float64).dot(tsc ** 2).ravel()
            p = np.where(den != 0.0, num / np.where(den == 0.0, 1.0, den), 0.0)
        else:
            p = inv_tt * self.Xm.T.dot(ts).ravel()
        if p_slot is not None:
            self.slots[p_slot[0]][:, p_slot[1]] = p
        ssq_p = self.comm.allreduce_host(np.array([(p[s:e] ** 2).sum() for s, e in self._ranges()]))
        if deflate:
            self.Xm = self.Xm - ts.reshape(-1, 1) * p
        if v is None:
            return ssq_p, None, None
        t, g = self._sweep_core(v, alpha, g_slot)
        return ssq_p, t, g

    def xt_v(self, V, alpha, slot, k0=0, want_ssq=False):
        V = np.asarray(V, dtype=np.float64)
        if V.ndim == 1:
            V = V.reshape(-1, 1)
        G = alpha * self.Xm.T.dot(V)
        self.slots[slot][:, k0:k0 + V.shape[1]] = G
        if want_ssq:
            return self.comm.allreduce_host(np.array([(G[s:e] ** 2).sum(axis=0) for s, e in self._ranges()]))

    def x_w_slot(self, slot, k0, c, per_block=False, overlap=None):
        if overlap is not None:
            overlap()
        return self.x_w_host(self.slots[slot][:, k0:k0 + c], per_block)

    def x_w_host(self, W, per_block=False):
        W = np.asarray(W, dtype=np.float64)
        if W.ndim == 1:
            W = W.reshape(-1, 1)
        if per_block:
            return [self.comm.allreduce_host(self.Xm[:, s:e].dot(W[s:e])) for s, e in self._ranges()]
        return self.comm.allreduce_host(self.Xm.dot(W))

    def gram(self):
        self.G = self.comm.allreduce_host(self.Xm.dot(self.Xm.T))
        return self

    def gram_apply(self, Z):
        return self.G.dot(np.asarray(Z).reshape(self.n, -1))

    class _X:
        def __init__(s, eng): s.eng = eng
        def download(s, order="F"): return np.asarray(s.eng.Xm, order=order)

    @property
    def X(self):
        return NumpyEngine._X(self)
