"""Where does a fit that now and then takes 10-30 ms instead of 1.5 spend the difference?  `sys.setprofile` on the
calling thread: for every fit slower than the threshold, the Python functions whose OWN time (children excluded)
exceeded 1 ms, with file:line.   python tools/stall_trace.py n p K method fits [two_phase] [gc_off]"""
import gc
import os
import sys
import time

T_START = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix

n, p, K = (int(v) for v in sys.argv[1:4])
method, fits = sys.argv[4], int(sys.argv[5])
flags = sys.argv[6:]
ctx = mbpls_amd.get_context()
if "two_phase" in flags:
    ctx.set_tunable("cluster_passes", 0)
if "gc_off" in flags:
    gc.disable()
Ts, Y = planted(n, 1, 32, 0)
X = DeviceMatrix(ctx, n, p)
X.fill_latent(ctx.from_host(Ts), 32, 0.5, 1234, 0)
ctx.sync()
print(f"before the fits: process cpu {time.process_time():.2f} s, wall since start {time.perf_counter() - T_START:.2f} s, "
      f"threads {len(os.listdir('/proc/self/task'))}", flush=True)

stack, own = [], []


def throttled():
    """(periods throttled, microseconds throttled) of this cgroup's CPU quota so far"""
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            d = dict(line.split() for line in open(path))
            return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", d.get("throttled_time", 0)))
        except Exception:
            pass
    return (-1, -1)


def switches():
    d = dict(line.split(":") for line in open("/proc/self/status") if "ctxt_switches" in line)
    return tuple(int(v) for v in d.values())


def prof(frame, event, arg):
    now = time.perf_counter()
    if event in ("call", "c_call"):
        name = (f"{os.path.basename(frame.f_code.co_filename)}:{frame.f_lineno}:{frame.f_code.co_name}" if event == "call"
                else f"<{getattr(arg, '__name__', arg)}> from {os.path.basename(frame.f_code.co_filename)}:{frame.f_lineno}")
        stack.append([name, now, 0.0])
    elif event in ("return", "c_return", "c_exception") and stack:
        name, t0, child = stack.pop()
        dur = now - t0
        if stack:
            stack[-1][2] += dur
        if dur - child > 1e-3:
            own.append((dur - child, name))


for i in range(fits):
    own.clear(); stack.clear()
    c0 = gc.get_count()
    th0, sw0, cpu0 = throttled(), switches(), time.process_time()
    sys.setprofile(prof)
    t0 = time.perf_counter()
    m = MBPLS(n_components=K, method=method).fit(X, Y)
    ctx.sync()
    dt = time.perf_counter() - t0
    sys.setprofile(None)
    th1, sw1, cpu1 = throttled(), switches(), time.process_time()
    line = (f"fit {i}: {1e3 * dt:7.2f} ms  gc {c0}  throttled +{th1[0] - th0[0]} periods +{th1[1] - th0[1]} us  "
            f"ctx switches vol +{sw1[0] - sw0[0]} invol +{sw1[1] - sw0[1]}  process cpu {1e3 * (cpu1 - cpu0):.1f} ms  "
            f"threads {len(os.listdir('/proc/self/task'))}")
    if dt > 0.006 and i >= 2:
        line += "  OWN > 1 ms: " + "; ".join(f"{nm} {1e3 * t:.2f}" for t, nm in sorted(own, reverse=True)[:6])
    print(line, flush=True)
X.close()
