"""world_size ranks sharing ONE GPU (gloo, host-staged exchange): the randomised sweep of tests/fuzz.py with every problem
column-sharded over the ranks -- every method, block splits, missing values, both routes of one-response NIPALS -- each rank
against the oracle's attributes as that rank holds them.  torch.distributed.run starts it (tests/test_gpu_rccl.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import mbpls_amd  # noqa: E402
from tests import fuzz  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    mbpls_amd.set_device(0)
    comm = mbpls_amd.TorchDistComm(device="cuda:0")
    cases, seed = int(os.environ.get("MBPLS_FUZZ_CASES", "40")), int(os.environ.get("MBPLS_FUZZ_SEED", "3"))
    bad, worst = fuzz.run(cases, seed, verbose=(rank == 0 or bool(os.environ.get("MBPLS_FUZZ_TRACE"))), shard=(comm, rank, world))
    everyone = [None] * world
    dist.all_gather_object(everyone, (bad, worst))
    assert all(b == 0 for b, _ in everyone), everyone
    assert max(w for _, w in everyone) < 1e-7, everyone
    print(f"rank {rank}/{world}: {cases} random sharded problems == oracle (worst {worst:.1e})", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
