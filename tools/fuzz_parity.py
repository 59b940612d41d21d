"""Randomised parity sweep on the GPU (see tests/fuzz.py):  python tools/fuzz_parity.py [cases] [seed] [big|long|wide|rccl1]
(big: 1024 < n <= 8192 instead of the few-samples shapes; long: 8192 < n <= 131072, narrow matrices; wide: n <= 300, p up to 200000)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import fuzz

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
mode = sys.argv[3] if len(sys.argv) > 3 else ""
if mode == "rccl1":
    # the sharded code path through the C ABI's RCCL communicator with ONE rank (what every rank of the 8-GPU run executes:
    # ncclAllReduce on the kernels' stream, device views, the column-by-column R_ under a communicator)
    import mbpls_amd
    comm = mbpls_amd.RcclComm(mbpls_amd.get_context(), 0, 1, lambda payload: payload)
    bad, worst = fuzz.run(cases, int(sys.argv[2]), shard=(comm, 0, 1))
    comm.close()
    print(f"{cases} cases through RcclComm (world size 1), {bad} failures, worst relerr {worst:.2e}")
    sys.exit(0)
bad, worst = fuzz.run(cases, int(sys.argv[2]) if len(sys.argv) > 2 else 0, big=mode == "big", long=mode == "long",
                      wide=mode == "wide")
print(f"{cases} cases, {bad} failures, worst relerr {worst:.2e}")
