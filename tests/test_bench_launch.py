"""`python bench.py --gpus N` without a launcher starts the N ranks itself (the driver's scaling run uses exactly that
form): rank environment, relay of rank 0's JSON line, failure propagation.  CPU only -- the children here are tiny
scripts, not the bench."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

CHILD = r"""
import json, os, sys, time
r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == str(r) and os.environ["LOCAL_WORLD_SIZE"] == str(w)
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
print("banner from rank", r)
mode = sys.argv[1]
if mode == "fail" and r == 1:
    sys.exit(7)
if mode == "fail":
    time.sleep(120)
print(json.dumps({"rank": r, "world": w, "port": os.environ["MASTER_PORT"]}))
"""


def test_ranks_started_and_rank0_line_relayed(capfd, monkeypatch):
    monkeypatch.setenv("MBPLS_BENCH_SKIP_GPU_COUNT", "1")
    monkeypatch.delenv("MASTER_PORT", raising=False)
    rc = bench.launch_ranks(3, ["-c", CHILD, "ok"])
    out, err = capfd.readouterr()
    assert rc == 0
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, out                   # ONE JSON line on stdout, everything else on stderr
    d = json.loads(lines[-1])
    assert d["rank"] == 0 and d["world"] == 3
    for r in range(3):
        assert f"banner from rank {r}" in err


def test_failing_rank_takes_the_job_down(capfd, monkeypatch):
    monkeypatch.setenv("MBPLS_BENCH_SKIP_GPU_COUNT", "1")
    t0 = time.time()
    rc = bench.launch_ranks(2, ["-c", CHILD, "fail"])
    out, err = capfd.readouterr()
    assert rc == 7
    assert time.time() - t0 < 30                  # rank 0 (sleeping) was ended, not waited for
    assert out.strip() == ""
    assert "rank 1 exited with 7" in err


def test_bench_refuses_more_ranks_than_gpus():
    """no launcher on the command line, fewer GPUs than ranks (none in this container): a message and a non-zero exit
    before any rank is started"""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("MBPLS_BENCH_SHARED_GPU", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "GPU(s) visible" in r.stderr and r.stdout.strip() == ""
