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


SLEEPER = r"""
import os, sys, time
open(os.path.join(sys.argv[1], "pid%s" % os.environ["RANK"]), "w").write(str(os.getpid()))
time.sleep(300)
"""

LAUNCHER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
os.environ["MBPLS_BENCH_SKIP_GPU_COUNT"] = "1"
import bench
sys.exit(bench.launch_ranks(2, ["-c", sys.argv[2], sys.argv[3]], timeout=float(sys.argv[4])))
"""


def _alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    # a zombie of an already reaped parent does not count
    try:
        return open(f"/proc/{pid}/stat").read().split(")")[-1].split()[0] != "Z"
    except OSError:
        return False


def _wait_for_pids(tmp, count, limit=60.0):
    t0 = time.time()
    while time.time() - t0 < limit:
        files = [os.path.join(tmp, f"pid{r}") for r in range(count)]
        if all(os.path.exists(f) and open(f).read().strip() for f in files):
            return [int(open(f).read()) for f in files]
        time.sleep(0.05)
    raise AssertionError("ranks did not start")


def test_signal_to_the_launcher_takes_the_ranks_down(tmp_path):
    """the advisor's case: `timeout 600 python bench.py --gpus 2` (or Ctrl-C) must not leave ranks behind in their own
    sessions, waiting in a collective with the GPUs leased"""
    import signal
    pr = subprocess.Popen([sys.executable, "-c", LAUNCHER, ROOT, SLEEPER, str(tmp_path), "600"])
    pids = _wait_for_pids(str(tmp_path), 2)
    assert all(_alive(p) for p in pids)
    pr.send_signal(signal.SIGTERM)
    rc = pr.wait(timeout=60)
    assert rc == 128 + signal.SIGTERM
    t0 = time.time()
    while any(_alive(p) for p in pids) and time.time() - t0 < 20:
        time.sleep(0.1)
    assert not any(_alive(p) for p in pids), pids


def test_killed_launcher_leaves_no_rank_behind(tmp_path):
    """SIGKILL cannot be handled: the ranks ask the kernel for a SIGTERM when their parent dies (PR_SET_PDEATHSIG)"""
    import signal
    pr = subprocess.Popen([sys.executable, "-c", LAUNCHER, ROOT, SLEEPER, str(tmp_path), "600"])
    pids = _wait_for_pids(str(tmp_path), 2)
    pr.send_signal(signal.SIGKILL)
    pr.wait(timeout=60)
    t0 = time.time()
    while any(_alive(p) for p in pids) and time.time() - t0 < 20:
        time.sleep(0.1)
    assert not any(_alive(p) for p in pids), pids


def test_timeout_takes_hanging_ranks_down(tmp_path):
    t0 = time.time()
    pr = subprocess.run([sys.executable, "-c", LAUNCHER, ROOT, SLEEPER, str(tmp_path), "2"], capture_output=True,
                        text=True, timeout=120)
    assert pr.returncode == 1, (pr.returncode, pr.stderr[-300:])
    assert "timeout after" in pr.stderr
    assert time.time() - t0 < 60
    pids = [int(open(os.path.join(str(tmp_path), f"pid{r}")).read()) for r in range(2)]
    assert not any(_alive(p) for p in pids)


def test_gpu_count_does_not_need_hip():
    """the launcher counts GPUs from the KFD topology (or in a throw-away child), never through HIP in-process"""
    n = bench.count_gpus_without_hip()
    assert n is None or (isinstance(n, int) and n >= 0)
    assert bench.DEFAULT_LAUNCH_TIMEOUT_S < float("inf")
