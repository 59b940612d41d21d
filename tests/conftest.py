import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_BLAS_CAP = []


def pytest_sessionstart(session):
    """The BLAS pools of the test process (the oracle's numpy) are capped at the CPU quota of the box for the whole
    session.  The GPU boxes show 256 CPUs under a 16-CPU cgroup quota: OpenBLAS sizes its pool from the first number
    (128 threads), the pool spins for ~0.1 s after every call, uses up the quota of the 100 ms scheduler period within
    12 ms and the whole process -- the thread that drives the GPU and the copier included -- stands still for the rest
    of it.  A fit of 1.5 ms then takes 75 (profiles/r06_stall_throttled.txt; tools/stall_trace.py).  Inside a fit the
    package limits the pools itself (runtime.host_threads); this is for the oracle fits and data generators around it."""
    try:
        import threadpoolctl
        from mbpls_amd import runtime
        _BLAS_CAP.append(threadpoolctl.threadpool_limits(limits=max(1, min(runtime.cpu_quota(), 32))))
    except Exception:           # no threadpoolctl: the tests still run, wall-clock guards discount throttled periods
        pass


def _gpu_present():
    try:
        from mbpls_amd import _capi
        return _capi.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_required():
    """GPU tests must run the HIP path: a missing library or device is a failure
    on the GPU box, never a silent skip to some fallback."""
    from mbpls_amd import _capi
    lib = _capi.load()  # raises loudly if the .so is missing
    assert _capi.device_count() > 0, "no HIP device visible"
    return lib


@pytest.fixture
def two_pass_nipals(monkeypatch):
    """One-response NIPALS with the reference's iteration structure (two passes per component, mbpls.py:841-913) instead
    of the default one-pass route (the second pass only re-derives the first): for the tests of that queue, its stop rule
    and its iteration counts."""
    monkeypatch.setenv("MBPLS_NIPALS_Q1_FIXED_POINT", "0")


@pytest.fixture(params=["one_pass", "two_passes"])
def q1_route(request, monkeypatch):
    """both routes of one-response NIPALS: the default (one pass per component) and the reference's two passes"""
    if request.param == "two_passes":
        monkeypatch.setenv("MBPLS_NIPALS_Q1_FIXED_POINT", "0")
    else:
        monkeypatch.delenv("MBPLS_NIPALS_Q1_FIXED_POINT", raising=False)
    return request.param


KERNEL_ROUTES = {"gram": "1", "products": "0"}


@pytest.fixture(params=list(KERNEL_ROUTES))
def kernel_route(request, monkeypatch):
    """both routes of KERNEL p > n: the association matrix X X^T on the matrix cores, and the maintained thin products
    (one pass over X per component); without the variable a cost model picks one (solvers._kernel_on_products)"""
    monkeypatch.setenv("MBPLS_KERNEL_GRAM", KERNEL_ROUTES[request.param])
    return request.param
