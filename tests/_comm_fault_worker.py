"""Worker of test_gpu_rccl.py::test_a_dead_peer_is_an_error_not_a_hang (one process, no launcher).

A real RCCL communicator of world size 1 on the process's context, then a fault injected through the C ABI's test hook
(mbpls_hip_comm_inject_fault): the next all-reduce is a kernel that waits as a collective waits for a peer that has
died.  argv[1] = "async": RCCL's asynchronous error turns up after a few polls; "deadline": it never does, the deadline
of the wait has to catch it.  Either way the sharded fit must RAISE CommLost within seconds -- never hang inside the
device queue --, later collectives must fail at once, and the context must remain usable without the communicator."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import mbpls_amd  # noqa: E402
from mbpls_amd import MBPLS, CommLost, RcclComm  # noqa: E402
from oracle import mbpls_oracle as orc  # noqa: E402
from tests import util  # noqa: E402


def main():
    mode, route = sys.argv[1], sys.argv[2]
    ctx = mbpls_amd.get_context()
    comm = RcclComm(ctx, 0, 1, lambda payload: payload)
    info = comm.info()
    assert info["rccl_ranks"] == 1 and info["rccl_version"] > 0, info
    q = 1 if route == "queue" else 2                 # one response: the whole fit is ONE device queue; two: host-driven passes
    X, Y = orc.make_latent(300, 2000, q, r=6, seed=4)
    Xs = [np.ascontiguousarray(X[:, :900]), np.ascontiguousarray(X[:, 900:])]
    with mbpls_amd.column_sharded(comm):
        good = MBPLS(n_components=3).fit(Xs, Y)      # the communicator works
    if mode == "async":
        comm.inject_fault(1, 3)
    else:
        comm.set_timeout(0.4)
        comm.inject_fault(2, 0)
    t0 = time.perf_counter()
    try:
        with mbpls_amd.column_sharded(comm):
            MBPLS(n_components=3).fit(Xs, Y)
    except CommLost as e:
        dt = time.perf_counter() - t0
        msg = str(e)
    else:
        print("the fit returned although its collective could not complete", flush=True)
        return 1
    assert dt < 6.0, dt                               # (the stand-in kernel itself would give up after ~8 s)
    assert ("asynchronous error" in msg) if mode == "async" else ("deadline" in msg), msg
    assert "rank 0 of 1" in msg and "aborted" in msg, msg
    t0 = time.perf_counter()
    try:
        comm.allreduce_host(np.ones(3))               # a dead communicator fails at once
    except CommLost as e:
        assert "aborted earlier" in str(e), e
    else:
        return 1
    assert time.perf_counter() - t0 < 0.5
    again = MBPLS(n_components=3).fit(Xs, Y)          # the context itself is intact
    assert util.compare(util.model_to_flat(again), util.model_to_flat(good), 1e-12) < 1e-12
    comm.close()
    print(f"dead peer ({mode}, {route}) detected after {dt:.2f} s: {msg[:160]}", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
