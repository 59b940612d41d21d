"""ctypes binding of ``libmbpls_hip.so`` (C ABI declared in ``include/mbpls_hip.h``).

This is the only module that touches the shared library.  There is NO fallback: if the
library is missing or a call fails, an exception is raised -- the product never computes
on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmbpls_hip.so")

ROW_MAJOR, COL_MAJOR = 0, 1
COMM_ID_BYTES = 128
SCALE_FIT, SCALE_APPLY, SCALE_NONE = 0, 1, 2
ABI_VERSION = 9

_vp, _i64, _dbl, _int, _sz = C.c_void_p, C.c_int64, C.c_double, C.c_int, C.c_size_t
_pp = C.POINTER(C.c_void_p)



class NipalsStep(C.Structure):
    """``mbpls_nipals_step`` of include/mbpls_hip.h (device-resident NIPALS step)."""
    _fields_ = [("n", _i64), ("nblocks", _int), ("q", _int), ("kc", _int), ("run", _int), ("nd", _int),
                ("centre", _int), ("norm_kind", _int), ("mode", _int), ("tol", _dbl), ("amp2_max", _dbl),
                ("xchg", _vp), ("Y", _vp), ("Ts", _vp), ("Tsc", _vp), ("U", _vp), ("V", _vp), ("Tb", _vp),
                ("scal", _vp), ("ssq_x", _vp), ("iters", _vp), ("u_cur", _vp), ("ts_old", _vp), ("vpass", _vp),
                ("tspass", _vp), ("cu", _vp), ("ce", _vp), ("par", _vp), ("nu2_host", _vp), ("C", _vp), ("ldc", _i64), ("host_seq", _vp), ("seq_value", _dbl)]


class Copy(C.Structure):
    """``mbpls_copy``: one device-to-host copy of a copier job"""
    _fields_ = [("h_dst", _vp), ("d_src", _vp), ("bytes", _sz)]


class UnitScale(C.Structure):
    """``mbpls_unit_scale``: h_dst[j] = h_src[j] / sqrt(h_nu2[block of j]) once the job's copies have landed"""
    _fields_ = [("h_src", _vp), ("h_dst", _vp), ("nblocks", _int), ("offsets", _vp), ("h_nu2", _vp)]


STEP_MAX_BLOCKS = 16
STEP_MAX_RESPONSES = 16

# name -> (restype, argtypes); must list every symbol of include/mbpls_hip.h
SIGNATURES = {
    "mbpls_hip_abi_version": (_int, []),
    "mbpls_hip_last_error": (C.c_char_p, []),
    "mbpls_hip_device_count": (_int, [C.POINTER(_int)]),
    "mbpls_hip_create": (_int, [_int, _vp, _pp]),
    "mbpls_hip_destroy": (_int, [_vp]),
    "mbpls_hip_warmup": (_int, [_vp]),
    "mbpls_hip_sync": (_int, [_vp]),
    "mbpls_hip_device_info": (_int, [_vp, C.c_char_p, _sz, C.POINTER(_int), C.POINTER(_sz), C.POINTER(_sz)]),
    "mbpls_hip_device_pci_bus_id": (_int, [_vp, C.c_char_p, _sz]),
    "mbpls_hip_set_tunable": (_int, [_vp, C.c_char_p, _int]),
    "mbpls_hip_fence": (_int, [_vp]),
    "mbpls_hip_cluster_status": (_int, [_vp, C.POINTER(_int)]),
    "mbpls_hip_cluster_status_into": (_int, [_vp, _vp]),
    "mbpls_hip_cluster_columns": (_int, [_vp, _i64, C.POINTER(_int)]),
    "mbpls_hip_malloc": (_int, [_vp, _sz, _pp]),
    "mbpls_hip_free": (_int, [_vp, _vp]),
    "mbpls_hip_memcpy_h2d": (_int, [_vp, _vp, _vp, _sz]),
    "mbpls_hip_memcpy_d2h": (_int, [_vp, _vp, _vp, _sz]),
    "mbpls_hip_memcpy_d2h_overlap": (_int, [_vp, _vp, _vp, _sz]),
    "mbpls_hip_memcpy_d2d": (_int, [_vp, _vp, _vp, _sz]),
    "mbpls_hip_memset": (_int, [_vp, _vp, _int, _sz]),
    "mbpls_hip_timer_start": (_int, [_vp]),
    "mbpls_hip_timer_stop": (_int, [_vp, C.POINTER(_dbl)]),
    "mbpls_hip_profile_enable": (_int, [_vp, _int]),
    "mbpls_hip_profile_reset": (_int, [_vp]),
    "mbpls_hip_profile_get": (_int, [_vp, _int, C.POINTER(_i64), C.POINTER(_dbl), C.POINTER(_dbl)]),
    "mbpls_hip_mat_create": (_int, [_vp, _i64, _i64, _int, C.POINTER(_i64), _vp, _pp]),
    "mbpls_hip_mat_destroy": (_int, [_vp]),
    "mbpls_hip_mat_info": (_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), _pp,
                                  C.POINTER(_int), C.POINTER(_int)]),
    "mbpls_hip_mat_upload": (_int, [_vp, _vp, _i64, _int, _i64, _i64]),
    "mbpls_hip_mat_download": (_int, [_vp, _vp, _i64, _int, _i64, _i64]),
    "mbpls_hip_mat_copy": (_int, [_vp, _vp]),
    "mbpls_hip_mat_fill_latent": (_int, [_vp, _vp, _int, _dbl, C.c_uint64, _i64]),
    "mbpls_hip_probe_read_bandwidth": (_int, [_vp, _int, C.POINTER(_dbl)]),
    "mbpls_hip_standardize_sweep": (_int, [_vp, _vp, _int, _vp, _vp, _vp, _vp, _dbl, _vp, _vp, _vp]),
    "mbpls_hip_standardize_sweep_masked": (_int, [_vp, _vp, _int, _vp, _dbl, _vp, _vp, _vp, _vp, _dbl, _vp, _vp, _vp]),
    "mbpls_hip_sweep": (_int, [_vp, _vp, _dbl, _vp, _vp, _vp]),
    "mbpls_hip_deflate_sweep": (_int, [_vp, _vp, _dbl, _int, _vp, _vp, _vp, _dbl, _vp, _vp, _vp]),
    "mbpls_hip_sweep_lazy": (_int, [_vp, _vp, _dbl, _vp, _vp, _vp, _vp, _i64, _int, _vp, _vp]),
    "mbpls_hip_standardize_stats_sweep": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _vp, _vp, _vp]),
    "mbpls_hip_nipals_gram_step": (_int, [_vp, _i64, _vp, _vp, _int, _vp, _int, _int, _dbl, _int, _vp, _vp, _vp, _vp, _int,
                                          _vp]),
    "mbpls_hip_nipals_subspace_loop": (_int, [_vp, _i64, _int, _vp, _vp, _vp, _int, _dbl, _int, _vp, _vp, _vp, _vp, _vp]),
    "mbpls_hip_nipals_subspace_qloop": (_int, [_vp, _i64, _int, _vp, _vp, _vp, _dbl, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mbpls_hip_scale_rows": (_int, [_vp, _i64, _int, _vp, _i64, _vp, _vp, _i64]),
    "mbpls_hip_sweep_lazy_scaled": (_int, [_vp, _vp, _dbl, _vp, _vp, _vp, _vp, _i64, _int, _vp, _vp, _vp, _vp]),
    "mbpls_hip_loadings_sweep_lazy_scaled": (_int, [_vp, _vp, _dbl, _vp, _vp, _vp, _dbl, _vp, _vp, _vp, _vp, _i64, _int,
                                                    _vp, _vp, _vp, _vp, _vp]),
    "mbpls_hip_loadings_sweep_raw": (_int, [_vp, _vp, _dbl, _vp, _vp, _vp, _dbl, _vp, _vp, _vp, _vp, _vp]),
    "mbpls_hip_loadings_sweep_lazy": (_int, [_vp, _vp, _dbl, _vp, _vp, _vp, _dbl, _vp, _vp, _vp, _vp, _i64, _int,
                                             _vp, _vp, _vp]),
    "mbpls_hip_xt_v": (_int, [_vp, _vp, _i64, _int, _dbl, _vp, _i64, _vp]),
    "mbpls_hip_x_w": (_int, [_vp, _vp, _i64, _int, _int, _vp, _vp, _vp, _i64]),
    "mbpls_hip_nan_standardize_sweep": (_int, [_vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mbpls_hip_nan_standardize_sweep_masked": (_int, [_vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mbpls_hip_nan_sweep": (_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "mbpls_hip_nan_deflate_sweep": (_int, [_vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mbpls_hip_x_w_nan": (_int, [_vp, _vp, _i64, _int, _int, _vp, _vp, _int, _vp, _i64]),
    "mbpls_hip_gram": (_int, [_vp, _vp, _i64]),
    "mbpls_hip_gram_scaled": (_int, [_vp, _vp, _vp, _int, _vp, _i64]),
    "mbpls_hip_xt_v_scaled": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _dbl, _vp, _i64, _vp]),
    "mbpls_hip_comm_unique_id": (_int, [_vp]),
    "mbpls_hip_comm_init_rank": (_int, [_vp, _int, _int, _vp]),
    "mbpls_hip_comm_info": (_int, [_vp, C.POINTER(_int), C.POINTER(_int), C.POINTER(_int)]),
    "mbpls_hip_comm_query": (_int, [_vp, C.POINTER(_int), C.POINTER(_int), C.POINTER(_int)]),
    "mbpls_hip_comm_destroy": (_int, [_vp]),
    "mbpls_hip_comm_set_timeout": (_int, [_vp, _int]),
    "mbpls_hip_comm_inject_fault": (_int, [_vp, _int, _int]),
    "mbpls_hip_allreduce": (_int, [_vp, _vp, _sz]),
    "mbpls_hip_nipals_step": (_int, [_vp, C.POINTER(NipalsStep)]),
    "mbpls_hip_set_pass_params": (_int, [_vp, _vp]),
    "mbpls_hip_set_variance_output": (_int, [_vp, _vp]),
    "mbpls_hip_block_unit_col": (_int, [_vp, _vp, _vp, _vp]),
    "mbpls_hip_set_true_space_outputs": (_int, [_vp, _vp, _vp]),
    "mbpls_hip_scale_col_by": (_int, [_vp, _i64, _vp, _vp, _vp]),
    "mbpls_hip_copier_submit": (_int, [_vp, _int, C.POINTER(Copy), C.POINTER(UnitScale), _vp, _dbl, C.POINTER(C.c_uint64)]),
    "mbpls_hip_copier_wait": (_int, [_vp, C.c_uint64]),
    "mbpls_hip_mat_transpose": (_int, [_vp, _vp]),
    "mbpls_hip_queue_r_update": (_int, [_vp, _i64, _vp, _vp, _i64, _int, _vp, _vp, _vp]),
    "mbpls_hip_flush_r_update": (_int, [_vp]),
    "mbpls_hip_host_alloc": (_int, [_vp, _sz, _pp, _pp]),
    "mbpls_hip_host_free": (_int, [_vp, _vp]),
    "mbpls_hip_cross_product": (_int, [_vp, _i64, _vp, _i64, _int, _vp, _i64, _int, _vp]),
    "mbpls_hip_cv_weights": (_int, [_vp, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mbpls_hip_xt_v_centred": (_int, [_vp, _vp, _int, _vp, _i64, _int, _vp, _i64]),
    "mbpls_hip_cv_fold_stats": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mbpls_hip_cross_product_dev": (_int, [_vp, _i64, _vp, _i64, _int, _vp, _i64, _int, _vp]),
    "mbpls_hip_finalise_small": (_int, [_vp, _int, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    "mbpls_hip_right_multiply_dev": (_int, [_vp, _i64, _vp, _i64, _int, _vp, _int, _vp, _i64]),
    "mbpls_hip_cv_qloops": (_int, [_vp, _int, _int, _vp, _dbl, _int, _vp]),
    "mbpls_hip_right_multiply_folds": (_int, [_vp, _i64, _int, _vp, _i64, _i64, _int, _vp, _int, _vp, _i64, _i64]),
    "mbpls_hip_right_multiply": (_int, [_vp, _i64, _vp, _i64, _int, _vp, _int, _vp, _i64]),
}

_lib = None


class HipError(RuntimeError):
    pass


class ClusterTimeout(HipError):
    """A hand-off between the workgroups of a column cluster gave up (csrc/colpass_cluster.hip): the results of that
    pass are garbage, nothing was written to X.  ``MBPLS._fit`` switches the context to the two-phase passes and runs
    the fit again."""


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipError(
            f"{LIB_PATH} not found: build it with `python -m mbpls_amd.build` (hipcc, gfx950). "
            "mbpls_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    v = lib.mbpls_hip_abi_version()
    if v != ABI_VERSION:
        raise HipError(f"ABI version mismatch: library {v}, binding {ABI_VERSION}")
    _lib = lib
    return lib


class CommLost(HipError):
    """A peer of the RCCL communicator has gone (asynchronous error or missed deadline, csrc/rccl_comm.hip): the
    communicator was aborted.  Nothing can be retried in this process group -- the caller lets the exception end the
    process (non-zero exit) and the launcher tears the job down."""


ERR_COMM = -5


def check(rc):
    if rc != 0:
        msg = load().mbpls_hip_last_error()
        text = f"mbpls_hip error {rc}: {msg.decode() if msg else '?'}"
        raise (CommLost if rc == ERR_COMM else HipError)(text)


def device_count():
    n = _int(0)
    check(load().mbpls_hip_device_count(C.byref(n)))
    return n.value


def _ptr(x):
    """Device pointer of a DeviceArray / int / None."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    return x.ptr


class Context:
    """One device + stream + workspace (``mbpls_ctx``)."""

    def __init__(self, device=0, stream=None):
        self.lib = load()
        h = _vp()
        check(self.lib.mbpls_hip_create(int(device), stream, C.byref(h)))
        self.handle = h
        self.device = int(device)
        # size-keyed cache of freed device buffers: repeated fits (bench steps, cross-validation)
        # reuse the 8np-byte working matrix instead of paying hipMalloc/hipFree of tens of GB
        # one fit / predict / transform at a time per context: the passes of a fit share the context's parameter block,
        # workspace, pinned rings and copier -- estimators driven from several threads (joblib's threading backend)
        # take turns (mbpls.MBPLS, model_selection)
        import threading
        self.lock = threading.RLock()
        self._pool = {}
        self._pool_bytes = 0
        self._mat_cache = {}        # (n, block offsets) -> upload matrix of a small problem, kept for its next fit (engine.py)
        self._offset_hints = {}     # (n, block offsets) of host data whose last fit needed centring passes (engine.py)
        self._host_pool = []        # owners of large host buffers handed out as attribute arrays (engine._pooled_host_array)
        self._pool_cap = None

    # ---- pooled raw allocations
    @staticmethod
    def _bucket(nbytes):
        """allocation size class: small buffers (vectors, K x K, parameter blocks) are pooled too -- a hipMalloc /
        hipFree pair per buffer and fit makes the runtime release memory in batches, and that release waits for the
        device (measured: one 20 ms stall of the running queue every sixth fit)"""
        nbytes = max(int(nbytes), 8)
        return (nbytes + 255) // 256 * 256 if nbytes < (1 << 20) else nbytes

    def _raw_alloc(self, nbytes):
        nbytes = self._bucket(nbytes)
        lst = self._pool.get(nbytes)
        if lst:
            self._pool_bytes -= nbytes
            return lst.pop()
        h = _vp()
        rc = self.lib.mbpls_hip_malloc(self.handle, nbytes, C.byref(h))
        if rc == -3 and self._pool_bytes:      # out of memory: drop the cache and retry once
            self.trim_pool()
            rc = self.lib.mbpls_hip_malloc(self.handle, nbytes, C.byref(h))
        check(rc)
        return h.value

    def _raw_free(self, ptr, nbytes):
        nbytes = self._bucket(nbytes)
        if self._pool_cap is None:
            # freed buffers are kept for reuse up to this share of HBM (MBPLS_POOL_CAP_FRACTION; set_pool_cap):
            # lower it when other allocators (torch, RCCL) share the device
            frac = float(os.environ.get("MBPLS_POOL_CAP_FRACTION", "0.5"))
            self._pool_cap = int(max(0.0, min(frac, 0.9)) * self.info()["total_bytes"])
        if self._pool_bytes + nbytes <= self._pool_cap:
            self._pool.setdefault(nbytes, []).append(ptr)
            self._pool_bytes += nbytes
        else:
            self.lib.mbpls_hip_free(self.handle, ptr)

    def set_pool_cap(self, fraction):
        """share of HBM the pool of freed buffers may hold (default 0.5); anything above is released now"""
        self._pool_cap = int(max(0.0, min(float(fraction), 0.9)) * self.info()["total_bytes"])
        if self._pool_bytes > self._pool_cap:
            self.trim_pool()

    def trim_pool(self):
        with self.lock:                                  # (not under a fit of another thread)
            self._host_pool = []                         # (large host buffers kept for the next fit's attributes, engine.py)
            kept, self._mat_cache = self._mat_cache, {}
            for m in kept.values():
                m.close()
            for lst in self._pool.values():
                for ptr in lst:
                    self.lib.mbpls_hip_free(self.handle, ptr)
            self._pool = {}
            self._pool_bytes = 0

    def close(self):
        if getattr(self, "handle", None):
            self._mapped = None     # (released by mbpls_hip_destroy, once the copier thread -- which may poll a flag in it -- has stopped)
            self.trim_pool()
            self.lib.mbpls_hip_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self.lib.mbpls_hip_sync(self.handle))

    def mapped_scalars(self, count):
        """(numpy view, device pointer) of a small block of pinned host memory the device writes directly; one per
        context, kept for its life (grown on demand)"""
        cur = getattr(self, "_mapped", None)
        if cur is None or cur[0].size < count:
            if cur is not None:
                self.lib.mbpls_hip_host_free(self.handle, cur[2])
            h, d = _vp(), _vp()
            size = max(int(count), 4096)
            check(self.lib.mbpls_hip_host_alloc(self.handle, 8 * size, C.byref(h), C.byref(d)))
            arr = np.ctypeslib.as_array((C.c_double * size).from_address(h.value))
            cur = self._mapped = (arr, d.value, h.value)
        return cur[0], cur[1]

    def copier_submit(self, copies, scale=None, flag=None):
        """Hand a job to the context's copier thread (``mbpls_hip_copier_submit``): ``copies`` = [(contiguous host
        array, device address)], carried out once everything queued on the compute stream so far has run;
        ``scale`` = (src column, dst column, int64 block offsets, address of the squared block norms in host-mapped
        memory): dst = src / sqrt(norm of its block) afterwards.  ``flag`` = (host address, value): no event is
        recorded on the stream; the job starts once the double at that address is >= value (written by a kernel queued
        behind the producers, ``mbpls_nipals_step.host_seq``).  Returns the ticket for ``copier_wait``.  The host
        arrays are kept alive here until a wait has covered them."""
        n = len(copies)
        arr = (Copy * max(n, 1))()
        for i, (dst, src) in enumerate(copies):
            arr[i].h_dst, arr[i].d_src, arr[i].bytes = dst.ctypes.data, int(src), dst.nbytes
        sc = None
        if scale is not None:
            src_h, dst_h, offs, nu2_addr = scale
            sc = C.byref(UnitScale(src_h.ctypes.data, dst_h.ctypes.data, len(offs) - 1, offs.ctypes.data, int(nu2_addr)))
        t = C.c_uint64()
        faddr, fval = (None, 0.0) if flag is None else (int(flag[0]), float(flag[1]))
        check(self.lib.mbpls_hip_copier_submit(self.handle, n, arr, sc, faddr, fval, C.byref(t)))
        keep = self.__dict__.setdefault("_copier_keep", [])
        keep.append((t.value, [c[0] for c in copies], scale))
        return t.value

    def copier_wait(self, ticket):
        """every job up to ``ticket`` has landed (raises the first failure of any of them)"""
        if not ticket:
            return
        try:
            check(self.lib.mbpls_hip_copier_wait(self.handle, C.c_uint64(int(ticket))))
        finally:
            keep = self.__dict__.get("_copier_keep")
            if keep:
                keep[:] = [e for e in keep if e[0] > ticket]

    def info(self):
        name = C.create_string_buffer(256)
        cus, tot, free = _int(0), _sz(0), _sz(0)
        check(self.lib.mbpls_hip_device_info(self.handle, name, 256, C.byref(cus), C.byref(tot), C.byref(free)))
        return dict(name=name.value.decode(), compute_units=cus.value, total_bytes=tot.value, free_bytes=free.value)

    def pci_bus_id(self):
        buf = C.create_string_buffer(64)
        check(self.lib.mbpls_hip_device_pci_bus_id(self.handle, buf, 64))
        return buf.value.decode()

    long_columns_from = 8192     # mirrors the C default; n above it -> long-column passes
    cluster_passes = 0 if os.environ.get("MBPLS_HIP_CLUSTER_PASSES", "1") == "0" else 1   # ... on clusters of workgroups

    def set_tunable(self, key, value):
        check(self.lib.mbpls_hip_set_tunable(self.handle, key.encode(), int(value)))
        if key == "long_columns_from":
            self.long_columns_from = max(0, min(int(value), 65536))
        if key == "cluster_passes":
            self.cluster_passes = 1 if value else 0

    def fence(self):
        """overlapped copies issued from now on wait for everything queued on the compute stream so far"""
        check(self.lib.mbpls_hip_fence(self.handle))

    def cluster_columns(self, n):
        """the never-writes-X passes of a matrix with n samples run on clusters of workgroups (one read of X)"""
        v = _int(0)
        check(self.lib.mbpls_hip_cluster_columns(self.handle, int(n), C.byref(v)))
        return bool(v.value)

    def cluster_status(self):
        """hand-offs of the cluster passes (long columns) that gave up since the last call; drains the stream"""
        v = _int(0)
        check(self.lib.mbpls_hip_cluster_status(self.handle, C.byref(v)))
        return int(v.value)

    def timer_start(self):
        check(self.lib.mbpls_hip_timer_start(self.handle))

    def timer_stop(self):
        ms = _dbl(0)
        check(self.lib.mbpls_hip_timer_stop(self.handle, C.byref(ms)))
        return ms.value

    # ---- per-kernel profile (HIP events on the launch stream)
    KERNEL_KINDS = ("standardize_sweep", "sweep", "deflate_sweep", "deflate", "xt_v", "x_w", "gram")

    def profile(self, on=True):
        check(self.lib.mbpls_hip_profile_enable(self.handle, int(bool(on))))

    def profile_reset(self):
        check(self.lib.mbpls_hip_profile_reset(self.handle))

    def profile_get(self):
        out = {}
        for k, name in enumerate(self.KERNEL_KINDS):
            cnt, ms, by = _i64(0), _dbl(0), _dbl(0)
            check(self.lib.mbpls_hip_profile_get(self.handle, k, C.byref(cnt), C.byref(ms), C.byref(by)))
            if cnt.value:
                out[name] = dict(launches=cnt.value, total_ms=ms.value, avg_ms=ms.value / cnt.value,
                                 bytes_per_launch=by.value / cnt.value,
                                 GBps=by.value / 1e9 / (ms.value / 1e3) if ms.value > 0 else 0.0)
        return out

    # ---- device arrays
    def empty(self, shape, zero=False):
        return DeviceArray(self, shape, zero=zero)

    def from_host(self, arr):
        arr = np.asarray(arr, dtype=np.float64)
        d = DeviceArray(self, arr.shape)
        d.set(arr)
        return d


class DeviceArray:
    """float64 device buffer; 2-D arrays are COLUMN-major (each column contiguous), matching
    the p x K / n x c conventions of the C ABI."""

    def __init__(self, ctx, shape, zero=False, ptr=None):
        # (dozens of views per fit are built through here: no numpy calls)
        if isinstance(shape, int):
            shape = (shape,)
        self.ctx = ctx
        self.shape = shape = tuple(int(s) for s in shape)
        size = 1
        for v in shape:
            size *= v
        self.size = size
        self.nbytes = 8 * size
        self._own = ptr is None
        if ptr is None:
            self.ptr = ctx._raw_alloc(self.nbytes)
        else:
            self.ptr = int(ptr)
        if zero:
            check(ctx.lib.mbpls_hip_memset(ctx.handle, self.ptr, 0, self.nbytes))

    @property
    def ld(self):
        return self.shape[0]

    def col(self, k, ncols=1):
        """View of columns [k, k+ncols) (no copy)."""
        rows = self.shape[0]
        return DeviceArray(self.ctx, (rows, ncols) if ncols > 1 else (rows,), ptr=self.ptr + 8 * rows * k)

    def set(self, arr):
        a = np.asarray(arr, dtype=np.float64)
        if a.ndim == 2:
            a = np.asfortranarray(a)
        else:
            a = np.ascontiguousarray(a)
        assert a.size == self.size, (a.shape, self.shape)
        check(self.ctx.lib.mbpls_hip_memcpy_h2d(self.ctx.handle, self.ptr, a.ctypes.data, self.nbytes))
        self._keep = a  # the copy is stream-ordered: keep the source alive until the next set()/free()
        return self

    def get(self):
        if len(self.shape) == 2:
            out = np.empty(self.shape, dtype=np.float64, order="F")
        else:
            out = np.empty(self.shape, dtype=np.float64)
        if self.size:
            check(self.ctx.lib.mbpls_hip_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr, self.nbytes))
        return out

    def free(self):
        if self._own and self.ptr:
            self.ctx._raw_free(self.ptr, self.nbytes)
            self.ptr = None

    def __del__(self):
        try:
            if self.ctx.handle:
                self.free()
        except Exception:
            pass


class DeviceMatrix:
    """Feature-major n x p matrix in HBM with block offsets (``mbpls_mat``)."""

    def __init__(self, ctx, n, p, block_offsets=None, storage=None):
        self.ctx = ctx
        self.n, self.p = int(n), int(p)
        offs = [0, self.p] if block_offsets is None else [int(o) for o in block_offsets]
        self.block_offsets = offs
        self.nblocks = len(offs) - 1
        arr = (_i64 * len(offs))(*offs)
        h = _vp()
        self._storage = None
        if storage is None:
            # storage comes from the context's pool (ld = n rounded up to even, as the library lays it out)
            self._storage = storage = DeviceArray(ctx, (((self.n + 1) & ~1) * self.p,))
        check(ctx.lib.mbpls_hip_mat_create(ctx.handle, self.n, self.p, self.nblocks, arr, _ptr(storage), C.byref(h)))
        self.handle = h
        n_, p_, ld_, d_, nb_, nc_ = _i64(), _i64(), _i64(), _vp(), _int(), _int()
        check(ctx.lib.mbpls_hip_mat_info(h, C.byref(n_), C.byref(p_), C.byref(ld_), C.byref(d_), C.byref(nb_), C.byref(nc_)))
        self.ld, self.data_ptr, self.nchunks = ld_.value, d_.value, nc_.value

    @property
    def nbytes(self):
        return 8 * self.ld * self.p

    def close(self):
        if getattr(self, "handle", None) and self.ctx.handle:
            self.ctx.lib.mbpls_hip_mat_destroy(self.handle)
            if self._storage is not None:
                self._storage.free()
                self._storage = None
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def transposed(self):
        """A new p x n DeviceMatrix holding this matrix's transpose (one block): the Gram kernel applied to it yields
        X^T X (``mbpls_hip_mat_transpose``).  The caller closes it."""
        T = DeviceMatrix(self.ctx, self.p, self.n)
        check(self.ctx.lib.mbpls_hip_mat_transpose(self.handle, T.handle))
        return T

    def upload(self, host, col0=0):
        """Copy a host n x w array into columns [col0, col0+w)."""
        a = np.asarray(host, dtype=np.float64)
        assert a.ndim == 2 and a.shape[0] == self.n, (a.shape, self.n)
        if a.flags.f_contiguous and not a.flags.c_contiguous:
            layout, ld = COL_MAJOR, a.strides[1] // 8
        else:
            if not a.flags.c_contiguous:
                # a column slice of a C-ordered array keeps unit column stride: pass it through un-copied
                if a.strides[1] == 8 and a.strides[0] % 8 == 0 and a.strides[0] >= 8 * a.shape[1]:
                    pass
                else:
                    a = np.ascontiguousarray(a)
            layout, ld = ROW_MAJOR, a.strides[0] // 8
        check(self.ctx.lib.mbpls_hip_mat_upload(self.handle, a.ctypes.data, ld, layout, int(col0), a.shape[1]))
        return self

    def download(self, col0=0, ncols=None, order="C"):
        ncols = self.p - col0 if ncols is None else ncols
        out = np.empty((self.n, ncols), dtype=np.float64, order=order)
        if order == "F":
            layout, ld = COL_MAJOR, self.n
        else:
            layout, ld = ROW_MAJOR, ncols
        check(self.ctx.lib.mbpls_hip_mat_download(self.handle, out.ctypes.data, ld, layout, int(col0), int(ncols)))
        return out

    def copy_from(self, other):
        check(self.ctx.lib.mbpls_hip_mat_copy(self.handle, other.handle))

    def probe_read_bandwidth(self, reps=3):
        """best read-only GB/s over this matrix's storage (measured ceiling of the HBM-bound passes)"""
        g = _dbl(0)
        check(self.ctx.lib.mbpls_hip_probe_read_bandwidth(self.handle, int(reps), C.byref(g)))
        return g.value

    def fill_latent(self, Ts_dev, r, noise, seed, col_base=0):
        check(self.ctx.lib.mbpls_hip_mat_fill_latent(self.handle, Ts_dev.ptr, int(r), float(noise), int(seed), int(col_base)))

    # ---- hot path
    def standardize_sweep(self, dst, mode, mean, scale, ssq_x, v=None, alpha=1.0, g=None, t=None, ssq_g=None,
                          rowmask=None, n_eff=0.0):
        check(self.ctx.lib.mbpls_hip_standardize_sweep_masked(
            self.handle, dst.handle, int(mode), _ptr(rowmask), float(n_eff), _ptr(mean), _ptr(scale), _ptr(ssq_x),
            _ptr(v), float(alpha), _ptr(g), _ptr(t), _ptr(ssq_g)))

    def sweep(self, v, alpha, g, t, ssq_g):
        check(self.ctx.lib.mbpls_hip_sweep(self.handle, _ptr(v), float(alpha), _ptr(g), _ptr(t), _ptr(ssq_g)))

    def deflate_sweep(self, ts, inv_tt, deflate, p, ssq_p, v=None, alpha=1.0, g=None, t=None, ssq_g=None):
        check(self.ctx.lib.mbpls_hip_deflate_sweep(self.handle, _ptr(ts), float(inv_tt), int(bool(deflate)), _ptr(p),
                                                   _ptr(ssq_p), _ptr(v), float(alpha), _ptr(g), _ptr(t), _ptr(ssq_g)))

    def standardize_stats_sweep(self, mode, mean, scale, rscale, r2, ssq_x, amp, v, alpha, g, t, ssq_g):
        check(self.ctx.lib.mbpls_hip_standardize_stats_sweep(self.handle, int(mode), _ptr(mean), _ptr(scale),
                                                             _ptr(rscale), _ptr(r2), _ptr(ssq_x), _ptr(amp), _ptr(v),
                                                             float(alpha), _ptr(g), _ptr(t), _ptr(ssq_g)))

    def sweep_lazy(self, v, alpha, g, t, ssq_g, P, ldp, kprev, cu, d, r2=None, mean=None):
        if r2 is not None:
            check(self.ctx.lib.mbpls_hip_sweep_lazy_scaled(self.handle, _ptr(v), float(alpha), _ptr(g), _ptr(t),
                                                           _ptr(ssq_g), _ptr(P), int(ldp), int(kprev), _ptr(cu),
                                                           _ptr(d), _ptr(r2), _ptr(mean)))
            return
        check(self.ctx.lib.mbpls_hip_sweep_lazy(self.handle, _ptr(v), float(alpha), _ptr(g), _ptr(t), _ptr(ssq_g),
                                                _ptr(P), int(ldp), int(kprev), _ptr(cu), _ptr(d)))

    def loadings_sweep_lazy(self, ts, inv_tt, p, ssq_p, v, alpha, g, t, ssq_g, P, ldp, kprev, cu, ce, d,
                            r2=None, mean=None):
        if r2 is not None:
            check(self.ctx.lib.mbpls_hip_loadings_sweep_lazy_scaled(
                self.handle, _ptr(ts), float(inv_tt), _ptr(p), _ptr(ssq_p), _ptr(v), float(alpha), _ptr(g), _ptr(t),
                _ptr(ssq_g), _ptr(P), int(ldp), int(kprev), _ptr(cu), _ptr(ce), _ptr(d), _ptr(r2), _ptr(mean)))
            return
        check(self.ctx.lib.mbpls_hip_loadings_sweep_lazy(self.handle, _ptr(ts), float(inv_tt), _ptr(p), _ptr(ssq_p),
                                                         _ptr(v), float(alpha), _ptr(g), _ptr(t), _ptr(ssq_g),
                                                         _ptr(P), int(ldp), int(kprev), _ptr(cu), _ptr(ce), _ptr(d)))

    def loadings_sweep_raw(self, ts, inv_tt, p, ssq_p, v, alpha, g, t, ssq_g, r2, mean=None):
        check(self.ctx.lib.mbpls_hip_loadings_sweep_raw(self.handle, _ptr(ts), float(inv_tt), _ptr(p), _ptr(ssq_p),
                                                        _ptr(v), float(alpha), _ptr(g), _ptr(t), _ptr(ssq_g),
                                                        _ptr(r2), _ptr(mean)))

    def xt_v(self, V, c, alpha, G, ssq=None, ldv=None, ldg=None, mean=None, rscale=None):
        if mean is not None:      # raw matrix, standardised on the fly
            check(self.ctx.lib.mbpls_hip_xt_v_scaled(self.handle, _ptr(mean), _ptr(rscale), _ptr(V), int(ldv or self.n),
                                                     int(c), float(alpha), _ptr(G), int(ldg or self.p), _ptr(ssq)))
            return
        check(self.ctx.lib.mbpls_hip_xt_v(self.handle, _ptr(V), int(ldv or self.n), int(c), float(alpha), _ptr(G),
                                          int(ldg or self.p), _ptr(ssq)))

    def x_w(self, W, c, out, per_block=False, mean=None, scale=None, ldw=None, ldo=None):
        check(self.ctx.lib.mbpls_hip_x_w(self.handle, _ptr(W), int(ldw or self.p), int(c), int(bool(per_block)),
                                         _ptr(mean), _ptr(scale), _ptr(out), int(ldo or self.n)))

    # ---- missing values (sparse_data=True)
    def nan_standardize_sweep(self, dst, mode, mean, scale, count, ssq_x, v=None, g=None, t=None, d=None, ssq_g=None,
                              rowmask=None):
        check(self.ctx.lib.mbpls_hip_nan_standardize_sweep_masked(
            self.handle, dst.handle, int(mode), _ptr(rowmask), _ptr(mean), _ptr(scale), _ptr(count), _ptr(ssq_x),
            _ptr(v), _ptr(g), _ptr(t), _ptr(d), _ptr(ssq_g)))

    def nan_sweep(self, v, g, t, d, ssq_g):
        check(self.ctx.lib.mbpls_hip_nan_sweep(self.handle, _ptr(v), _ptr(g), _ptr(t), _ptr(d), _ptr(ssq_g)))

    def nan_deflate_sweep(self, ts, deflate, p, ssq_p, v=None, g=None, t=None, d=None, ssq_g=None):
        check(self.ctx.lib.mbpls_hip_nan_deflate_sweep(self.handle, _ptr(ts), int(bool(deflate)), _ptr(p), _ptr(ssq_p),
                                                       _ptr(v), _ptr(g), _ptr(t), _ptr(d), _ptr(ssq_g)))

    def x_w_nan(self, W, c, out, per_block=False, mean=None, scale=None, indicator=False, ldw=None, ldo=None):
        check(self.ctx.lib.mbpls_hip_x_w_nan(self.handle, _ptr(W), int(ldw or self.p), int(c), int(bool(per_block)),
                                             _ptr(mean), _ptr(scale), int(bool(indicator)), _ptr(out),
                                             int(ldo or self.n)))

    def gram(self, G, ldg=None, mean=None, rscale=None, centre_first=True):
        check(self.ctx.lib.mbpls_hip_gram_scaled(self.handle, _ptr(mean), _ptr(rscale), int(bool(centre_first)), _ptr(G),
                                                 int(ldg or self.n)))
