"""bench.py itself on the GPU box, the way the driver runs it for N > 1 -- the path a single-GPU pool never takes by
accident: (a) as ONE rank under torch.distributed.run over the C ABI's RCCL communicator (world size = the GPUs of the
box, 1 here): sharded self-check of both NIPALS routes and UNIPALS against the unsharded fit and the CPU oracle, timed
sharded fits, the strong-scaling leg, the collective's latency; (b) self-launched, two ranks sharing GPU 0 over gloo
(`--gpus 2`, MBPLS_BENCH_SHARED_GPU=1).  Small matrices: the numbers mean nothing, the line and the exit status do."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.lstrip().startswith("{")]
    assert lines, stdout[-2000:]
    return json.loads(lines[-1])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_as_a_rank_over_rccl(gpu_required):
    from mbpls_amd import _capi
    world = max(1, min(_capi.device_count(), 2))
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world), "--p", "40000", "--steps", "2", "--warmup", "1", "--cpu-cols", "4096", "--e2e-cols", "0"]
    if world == 1:
        cmd.append("--dist")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = _line(r.stdout)
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == world and d["value"] > 0
    assert d["comm"]["backend"].startswith("rccl") and d["comm"]["rccl_ranks"] == world
    chk = d["sharded_self_check"]
    assert chk["worst_rel_err"] < 1e-8 and chk["worst_rel_err_vs_oracle_rank0"] < 1e-6 and len(chk["methods"]) == 3
    assert d["allreduce"]["mean_us"] > 0 and d["strong_scaling_c2"]["fit_wall_s"] > 0
    assert d["config"]["nipals_iterations"] == [1] * 10 and "fixed point" in d["config"]["route"]


def test_bench_self_launched_two_ranks_on_one_gpu(gpu_required):
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", MBPLS_BENCH_SHARED_GPU="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--p", "40000", "--steps", "2", "--warmup", "1",
           "--cpu-cols", "4096"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = _line(r.stdout)
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 2 and "sharded over 2 GPUs" in d["config"]["workload"]
    assert d["sharded_self_check"]["worst_rel_err"] < 1e-8
    assert d["comm"]["world_size"] == 2 and d["cpu_baseline"]["value"] > 0
