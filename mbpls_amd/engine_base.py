"""The engine interface the solvers are written against.

``solvers.py`` calls these methods unconditionally; an engine overrides what it accelerates.  ``HipEngine`` (the
product: HIP kernels through the C ABI) overrides all of it; the test-only numpy engine of ``tests/numpy_engine.py``
(kernel semantics of include/mbpls_hip.h in numpy, for the CPU tests of the host drivers and of the column sharding)
overrides the products and passes and inherits the plain defaults below -- no optional fast path is taken, every fit
goes through the host-driven loops.
"""
from __future__ import annotations

import numpy as np

from .plan import FitOptions


class EngineBase:
    sparse = False
    rowmask = None
    # capabilities that change the solver route (plain data, decided by the engine class)
    has_association_matrices = False     # gram_blocks / gram_power_loop (several-response NIPALS on per-block Grams)
    has_device_helpers = False           # asynchronous slot algebra with a device context (helper-thread fetches)

    def _init_base(self):
        self.opts = FitOptions.from_env()     # route switches, read once per engine (= per fit)
        self.plan = {}                        # the route this fit took (MBPLS.fit_plan_)
        self.timings = {}

    # ---- host buffers
    def host_array(self, shape, reserve=None, prefault=True, prefault_tail_only=False):
        return np.empty(shape, order="F")

    def run_deferred(self, count=1, all_of_it=False):
        """host chores an estimator queued for the time the device is busy (HipEngine); nothing to do here"""

    def fetch_async(self, name):
        out = self.fetch(name)
        return out, (lambda: out)

    # ---- finished columns on their way to the host while the fit goes on (HipEngine: the copier thread)
    def send_cols(self, cols, unit=None):
        """cols = [(slot name, k, F-ordered host array)]: column k of each slot is final and goes to column k of its
        host array; ``unit`` = (src host array, dst host array, k, block norms): afterwards dst[:, k] = src[:, k]
        block by block over its norm (w_b = w~_b / ||w~_b||, mbpls.py:860 / :510-512; src may be dst).  Returns a ticket
        for ``wait_sent``"""
        for name, k, out in cols:
            self.fetch_col(name, k, out)
        if unit is not None:
            src, dst, k, nu = unit
            offs = self.block_offsets
            for b in range(self.nblocks):
                np.divide(src[offs[b]:offs[b + 1], k], nu[b], out=dst[offs[b]:offs[b + 1], k])
        return 0

    def send_slot(self, name, out):
        out[...] = self.fetch(name)
        return 0

    def wait_sent(self, ticket):
        pass

    # ---- routes an engine may offer
    def use_lazy_deflation(self, K, pslot, prefer=None):
        return False

    def resident_supported(self, q, K, norm_kind):
        return False

    def scaled_products(self):
        pass

    def sweep_reads_once(self):
        return True

    def stats_only_ok(self):
        return False

    def check_cluster_passes(self):
        pass

    def nipals_subspace_loop(self, G, Y, c0, norm_kind, tol, max_iter):
        return None                           # "outside the kernel's range": the caller iterates on the host

    def sync(self):
        pass

    def xtx_operator(self):
        """callable w (p x c host) -> (X^T X) w: here through a host copy of the p x p matrix.  Column shards: the
        rows of X^T X that belong to this rank's features need every rank's columns, so (X^T X) w is taken as
        X_r^T (sum over ranks of X_s w_s) -- two products per call, the first one summed over the shards."""
        if getattr(self.comm, "is_distributed", False):
            def apply(w):
                w = np.asarray(w, dtype=np.float64).reshape(self.p, -1)
                t = self.x_w_host(w)                        # n x c, summed over the shards
                self.slot("XtXw", w.shape[1])
                self.xt_v(t, 1.0, "XtXw", 0)
                return self.fetch("XtXw")[:, :w.shape[1]].copy()
            return apply
        self.slot("XtX", self.p)
        self.xt_self("XtX")
        VAR = self.fetch("XtX")
        return lambda w: VAR.dot(w)

    def xt_self(self, slot):
        """slot (p x p) = X^T X"""
        self.xt_v(self.X.download(order="F"), 1.0, slot, 0)
