"""The sharded product path over RCCL (torch.distributed backend "nccl") on the GPU box."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("which,q", [("torch", 2), ("torch", 1), ("rccl", 2), ("rccl", 1), ("rccl", 3)])
def test_rccl_sharded_fit(gpu_required, which, q):
    """All four methods column-sharded over the visible GPUs, the exchange through torch.distributed ("nccl" = RCCL)
    or through the C ABI's own RCCL communicator (ncclAllReduce on the context's stream; with one response the whole
    NIPALS fit is then a single device queue).  One GPU here: world_size 1, every piece of plumbing is the one the
    8-GPU run uses."""
    from mbpls_amd import _capi
    n = max(1, min(_capi.device_count(), 2))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", MBPLS_TEST_COMM=which, MBPLS_TEST_Q=str(q))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_nccl_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("RCCL-sharded fit == oracle") == n, r.stdout


def test_two_ranks_share_one_gpu(gpu_required):
    """world_size 2 with the real engine on ONE GPU (gloo, host-staged exchange): see _shared_gpu_worker.py."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_shared_gpu_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("sharded fit on a shared GPU == oracle") == 2, r.stdout


def test_two_ranks_random_problems(gpu_required):
    """the randomised sweep (tests/fuzz.py) with every problem column-sharded over two ranks that share the GPU"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", MBPLS_FUZZ_CASES="40", MBPLS_FUZZ_SEED="3")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_shared_gpu_fuzz_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("random sharded problems == oracle") == 2, r.stdout


def _run_rank_equality(world, shared):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", MBPLS_TEST_SHARED_GPU="1" if shared else "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_rank_equality_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("bit-equal on all ranks, all == oracle") == world, r.stdout


def test_ranks_agree_bit_for_bit(gpu_required):
    """BASELINE config[4]'s premise on real hardware: one rank per GPU over RCCL, every replicated attribute and the
    iteration counts bit-equal across ranks and equal to the oracle.  Needs two GPUs (skips on the single-GPU pool; the
    same worker runs there with both ranks on one GPU over gloo, below)."""
    from mbpls_amd import _capi
    if _capi.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device)")
    _run_rank_equality(min(_capi.device_count(), 8), shared=False)


def test_ranks_agree_bit_for_bit_on_a_shared_gpu(gpu_required):
    _run_rank_equality(2, shared=True)


@pytest.mark.parametrize("mode", ["async", "deadline"])
@pytest.mark.parametrize("route", ["queue", "host"])
def test_a_dead_peer_is_an_error_not_a_hang(gpu_required, mode, route):
    """mbpls_hip_allreduce queues a collective into a device queue; a peer that dies would leave every later wait for
    the stream hanging for ever.  With a communicator on the context every such wait polls the stream, asks
    ncclCommGetAsyncError and keeps a deadline; on either the communicator is aborted and the call returns
    MBPLS_ERR_COMM (``CommLost``).  Fault-injected at world size 1 (tests/_comm_fault_worker.py): asynchronous error /
    silent peer, device-resident queue / host-driven passes."""
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_comm_fault_worker.py"), mode, route], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "dead peer" in r.stdout and "detected after" in r.stdout, r.stdout
