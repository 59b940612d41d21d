"""Column-sharded fit with world_size 2 on CPU (gloo): the N>1 path's host logic and collectives."""
import os
import socket
import subprocess
import sys

import pytest

import numpy as np

from mbpls_amd.comm import shard_columns

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_columns_partition():
    for sizes in ([10], [7, 13, 1], [300000, 300000, 400000]):
        for world in (1, 2, 3, 8):
            cover = [np.zeros(s, dtype=int) for s in sizes]
            widths = []
            for r in range(world):
                rs = shard_columns(sizes, r, world)
                widths.append([hi - lo for lo, hi in rs])
                for c, (lo, hi) in zip(cover, rs):
                    c[lo:hi] += 1
            assert all((c == 1).all() for c in cover)              # exact partition of every block
            w = np.array(widths)
            assert (w.max(axis=0) - w.min(axis=0) <= 1).all()      # balanced


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_fit_gloo(world):
    """p > n and n >= p fixtures, four methods, with / without standardisation and calc_all: every rank's shard of the
    attributes equals the unsharded reference (three ranks: uneven shards)"""
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_sharded_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("sharded == unsharded reference") == world, r.stdout
