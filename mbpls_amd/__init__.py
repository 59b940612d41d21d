"""mbpls_amd -- MI355X-native hot path of (multiblock) PLS fitting.

    from mbpls_amd import MBPLS          # same API as mbpls.mbpls.MBPLS
    MBPLS(n_components=10, method='NIPALS').fit(X, y).predict(X)

The data matrix lives feature-major in HBM and is touched only by hand-written gfx950 HIP
kernels (``csrc/``) reached through the C ABI of ``include/mbpls_hip.h``.
"""
from ._capi import CommLost, Context, DeviceArray, DeviceMatrix, HipError, device_count
from .comm import LocalComm, RcclComm, TorchDistComm, shard_columns
from .mbpls import MBPLS
from .model_selection import cross_val_predict
from .runtime import column_sharded, get_comm, get_context, set_comm, set_device

__all__ = ["MBPLS", "cross_val_predict", "Context", "DeviceArray", "DeviceMatrix", "HipError", "CommLost", "device_count", "LocalComm",
           "RcclComm", "TorchDistComm", "shard_columns", "column_sharded", "get_comm", "get_context", "set_comm",
           "set_device"]
__version__ = "0.1.0"
