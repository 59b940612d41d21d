#!/usr/bin/env python3
"""bench.py -- MBPLS.fit() on device-resident synthetic wide matrices (BASELINE.json metric:
"fit() wall-time + achieved HBM GB/s, n=4096 p=1e6 K=10; 1/2/4/8 GPU").

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one complete ``MBPLS(n_components=10, method='NIPALS').fit(X, Y)`` on a matrix that is already
resident in HBM (feature-major): statistics + all components + the fitted attributes fetched to the host.
With one response the reference's second NIPALS pass of a component only re-derives the first (u is y / v whatever
ts was), so the default route reads X once per component -- K + 1 reads per fit, every attribute as the reference's;
the leg ``reference_iteration_structure`` times the same fit with the reference's two passes per component (2K + 1
reads, MBPLS_NIPALS_Q1_FIXED_POINT=0) beside it.
X is generated on the device (planted latent structure), is NOT modified by fit (copy=True; with one
response the fit never writes X at all), so every step does identical work.

N = 1 : BASELINE config[1]  single-block NIPALS n=4096 p=1,000,000 K=10 q=1 float64.
N > 1 : BASELINE config[4]  multiblock (4 blocks) NIPALS, n=4096, p = N x 1,000,000: the feature columns of every
        block are split evenly over the ranks (weak scaling, 32.8 GB of X per GPU), one RCCL all-reduce of the
        exchange buffer [t~ (n x B), ||w~_b||^2, ...] per pass, enqueued on the kernels' stream by the C ABI's own
        communicator (mbpls_hip_allreduce).  Before timing every rank checks a small sharded fit against the same fit
        done unsharded on its own GPU and exits non-zero on a mismatch; the strong-scaling figure (config[1] split
        over the N GPUs) is reported beside the weak-scaling one.

value = ACHIEVED HBM traffic of the whole job / wall time [GB/s]: the X bytes the kernels of one fit really move
        (sum of their algorithmic bytes, 8np per read-only pass -- PMC-measured traffic is 1.003x that,
        profiles/) x ranks / time per fit.  fit_wall_s and fits_per_s are printed beside it; the reference
        algorithm's un-fused byte count ("effective" GB/s) is an extra key, never `value`.
Side legs of the N = 1 line (never `value`): shard_proxy (this problem / 8 through the RCCL communicator), explicit
        deflation, end to end from host arrays, other_solvers_same_matrix (ten responses with NIPALS / UNIPALS, SIMPLS
        and UNIPALS with one response on the same resident matrix), cpu_baseline.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E (MI355X_MICROARCH.md); the guide's measured copy ceiling is ~6290 GB/s


def note(msg):
    """progress on stderr (stdout carries the ONE JSON line): which leg a failure or a slow run belongs to"""
    if os.environ.get("RANK", "0") == "0":
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def planted(n, q, r, seed, decay=0.85, noise_y=0.5):
    rng = np.random.RandomState(seed)
    T = rng.standard_normal((n, r))
    C = rng.standard_normal((r, q))
    Y = T.dot(C) + noise_y * rng.standard_normal((n, q))
    return T * decay ** np.arange(r), Y


def fill_blocks(ctx, X, dT, r, offs, rank, world, pb_global):
    """rank `rank` owns columns [rank*pb_loc, (rank+1)*pb_loc) of each global block"""
    from mbpls_amd import DeviceMatrix
    for b in range(len(offs) - 1):
        w = offs[b + 1] - offs[b]
        view = DeviceMatrix(ctx, X.n, w, None, storage=X.data_ptr + 8 * X.ld * offs[b])
        view.fill_latent(dT, r, 0.5, 1234, b * pb_global + rank * w)
        view.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (the first fit of a process builds pools, pinned buffers and scratch; every later fit repeats within ~0.5 % --
    # fit_wall_s_each lists them all)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--p", type=int, default=1_000_000, help="feature columns PER GPU")
    ap.add_argument("--K", type=int, default=10)
    ap.add_argument("--q", type=int, default=1)
    ap.add_argument("--method", default="NIPALS")
    ap.add_argument("--dist", action="store_true",
                    help="use the sharded exchange path and the N>1 block layout even at world size 1")
    ap.add_argument("--comm", default=os.environ.get("MBPLS_BENCH_COMM", "rccl"), choices=["rccl", "torch"],
                    help="N>1 exchange: the C ABI's RCCL communicator (default) or torch.distributed all_reduce")
    ap.add_argument("--cpu-cols", type=int, default=32768,
                    help="largest column count of the CPU-baseline samples (three sizes: /4, /2, /1; 0 = skip)")
    ap.add_argument("--e2e-cols", type=int, default=-1,
                    help="columns of the end-to-end (host numpy in) leg; -1 = all when host RAM allows, 0 = skip")
    ap.add_argument("--no-extras", action="store_true", help="skip the side legs (explicit deflation, reference iteration structure, shard proxy ...)")
    ap.add_argument("--first-fit-child", action="store_true",
                    help="(internal) fresh process: generate X, time the FIRST fit of the process and the second, print JSON")
    ap.add_argument("--no-first-fit", action="store_true", help="skip the cold-start leg (a child process run first)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the two rocprofv3 --pmc child runs (FETCH_SIZE, WRITE_SIZE) that measure roofline.traffic "
                         "live; the committed figure under profiles/ is reported instead, labelled so")
    ap.add_argument("--extras-child", action="store_true",
                    help="(internal) the side legs of the N = 1 line in a process of their own; prints one JSON object")
    ap.add_argument("--full-fit-s", type=float, default=0.0, help="(internal) the parent's timed fit, for the shard proxy")
    ap.add_argument("--extras-in-process", action="store_true", help="run the side legs in the timing process itself")
    a = ap.parse_args()
    if a.first_fit_child:
        return first_fit_child(a)
    if a.extras_child:
        return extras_child(a)

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process only starts the N ranks (fresh processes,
        # before anything here has touched the GPU), relays rank 0's JSON line and exits with their status
        sys.exit(launch_ranks(a.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        sys.exit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    # cold start: what a user who calls fit() once per process pays.  A CHILD process, started before this one has
    # touched the GPU, generates the same matrix and times its first and second fit.
    cold = None
    if world == 1 and not a.dist and not a.no_first_fit and not a.no_extras:
        cold = run_first_fit_child(a)
    # roofline.traffic: HBM bytes per launch from the PMC counters of THIS machine -- one fit of the same workload in a
    # child process under `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE in separate runs, with --kernel-trace only), also
    # before this process touches the GPU
    live_pmc = None
    if world == 1 and not a.dist and not a.no_pmc and not a.no_extras:
        live_pmc = run_pmc_children(a)
    import mbpls_amd
    from mbpls_amd import MBPLS, DeviceMatrix

    dist = None
    comm_info = None
    shared_gpu = os.environ.get("MBPLS_BENCH_SHARED_GPU") == "1"
    if world > 1 or a.dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # N processes share one node's CPU quota: keep torch's host pools from oversubscribing it
        os.environ.setdefault("OMP_NUM_THREADS", "1")
        import torch
        import torch.distributed as dist
        torch.set_num_threads(1)
        if shared_gpu:
            # dry run of the N>1 code path on a single-GPU box: all ranks drive cuda:0, exchange host-staged
            # over gloo (RCCL refuses duplicate devices).  Not a measurement configuration.
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        mbpls_amd.set_device(local_rank)
        ctx = mbpls_amd.get_context()
        comm = None
        if not shared_gpu and a.comm == "rccl":
            # the C ABI's own communicator; every rank must agree on the outcome, so a failure anywhere (librccl not
            # loadable, init error) sends ALL ranks to the torch.distributed all_reduce -- also RCCL, through torch
            err = None
            try:
                comm = mbpls_amd.RcclComm.from_torch(ctx)
                comm_info = dict(comm.info(), backend="rccl (C ABI, mbpls_hip_allreduce on the kernels' stream)")
                # what RCCL itself reports (ncclCommCount / ncclCommUserRank), not what was passed in
                assert comm_info["rccl_ranks"] == world and comm_info["rccl_user_rank"] == rank, comm_info
                # one small collective under a short deadline before anything is timed: a communicator that initialises
                # but cannot move data (a transport the node does not offer, a peer that never arrives) is an error
                # here -- CommLost after 90 s -- and every rank falls back together, instead of a hang in the first fit
                comm.set_timeout(90.0)
                got = float(comm.allreduce_host(np.array([rank + 1.0]))[0])
                assert got == 0.5 * world * (world + 1), f"probe all-reduce returned {got}"
                comm.set_timeout(600.0)
            except Exception as e:                      # noqa: BLE001 -- reported, then the fallback below
                err = f"{type(e).__name__}: {e}"
            flags = [None] * world
            dist.all_gather_object(flags, err)
            if any(f is not None for f in flags):
                if comm is not None:
                    comm.close()
                comm = None
                print(f"[bench] rank {rank}: C-ABI RCCL communicator unavailable ({[f for f in flags if f][0]}); "
                      f"using torch.distributed all_reduce", file=sys.stderr, flush=True)
        if comm is None:
            comm = mbpls_amd.TorchDistComm(device=f"cuda:{local_rank}")
            comm_info = {"backend": "torch.distributed/" + str(dist.get_backend()), "world_size": comm.world_size}
        mbpls_amd.set_comm(comm)
        # which physical device every rank drives (hipDeviceGetPCIBusId) beside what RCCL reports for it: the record of
        # an N-GPU run shows N DISTINCT devices -- and the run refuses to time anything otherwise
        mine = {"rank": rank, "local_rank": local_rank, "pci_bus_id": ctx.pci_bus_id(),
                "rccl_device": (comm_info or {}).get("rccl_device"), "rccl_user_rank": (comm_info or {}).get("rccl_user_rank")}
        devices = [None] * world
        dist.all_gather_object(devices, mine)
        print(f"[bench] rank {rank}: device {mine['pci_bus_id']} (local rank {local_rank}, rccl_device {mine['rccl_device']})",
              file=sys.stderr, flush=True)
        comm_info = dict(comm_info, devices=devices)
        if not shared_gpu and len({d["pci_bus_id"] for d in devices}) != world:
            sys.exit(f"[bench] {world} ranks on {len({d['pci_bus_id'] for d in devices})} distinct device(s): {devices}")
    ctx = mbpls_amd.get_context()

    def barrier_sync():
        ctx.sync()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
        ctx.sync()

    def max_over_ranks(x):
        if dist is None:
            return x
        import torch
        dev = "cpu" if shared_gpu else f"cuda:{local_rank}"
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    n, p, K, q = a.n, a.p, a.K, a.q
    r = 32
    Ts, Y = planted(n, q, r, seed=0)
    dT = ctx.from_host(Ts)

    check = None
    if dist is not None:
        check = sharded_self_check(ctx, mbpls_amd, dT, Y, r, rank, world)      # exits non-zero on a mismatch

    # global matrix: world*p columns; N>1 = 4 equal blocks, every block split evenly over ranks
    B = 1 if dist is None else 4
    pb = p // B
    offs = [b * pb for b in range(B)] + [p]
    X = DeviceMatrix(ctx, n, p, offs)
    if dist is None:
        X.fill_latent(dT, r, 0.5, 1234, 0)
    else:
        fill_blocks(ctx, X, dT, r, offs, rank, world, world * pb)
    ctx.sync()

    def new_model():
        return MBPLS(n_components=K, method=a.method, standardize=True, calc_all=True, copy=True)

    # every step is the fit of a FRESH estimator in a warm process -- what cross-validation and every real caller do
    # (a refit of the same estimator is timed beside it, outside the timed region).  The fitted models stay alive until
    # the timed region ends: releasing the previous result (0.35 GB of attributes at p = 1e6) is not part of fit().
    note(f"timed region: {a.warmup} warm-up + {a.steps} timed fits")
    for _ in range(a.warmup):
        new_model().fit(X, Y)
    ctx.profile(True)
    ctx.profile_reset()
    per_fit, per_fit_phases, models = [], [], []
    barrier_sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        t1 = time.perf_counter()
        m = new_model()
        m.fit(X, Y)                        # returns with the fitted attributes on the host
        per_fit.append(time.perf_counter() - t1)
        per_fit_phases.append(m.fit_timings_)
        models.append(m)
    barrier_sync()
    dt = max_over_ranks(time.perf_counter() - t0)
    prof = ctx.profile_get()
    ctx.profile(False)
    mdl = models[-1]
    del models, m
    # the same estimator fitted again and again (round-2 protocol), for comparison
    refit = []
    for i in range(4):
        barrier_sync()
        t1 = time.perf_counter()
        mdl.fit(X, Y)
        barrier_sync()
        if i:
            refit.append(time.perf_counter() - t1)
    iters = list(getattr(mdl, "nipals_iterations_", [2] * K))
    phases = {k: round(1e3 * v, 2) for k, v in getattr(mdl, "fit_timings_", {}).items()}

    extras = {}
    e2e_from_child = None
    if dist is None and not a.no_extras and not a.extras_in_process:
        # The side legs (explicit deflation, the reference's iteration structure, shard proxy, the other solvers, end to end from host
        # memory) run in a CHILD process: nothing they do -- a fault in a secondary kernel included -- can cost the line
        # its headline numbers, which are complete at this point.
        note("side legs in a child process")
        extras = run_extras_child(a, dt / a.steps)
        e2e_from_child = extras.pop("end_to_end_host_numpy", None)
    elif dist is None and not a.no_extras:
        extras = side_legs(ctx, mbpls_amd, a, mdl, X, Y, dT, r, n, p, K, q, len(offs) == 2, barrier_sync, dt / a.steps)
    elif not a.no_extras and a.method == "NIPALS" and q == 1:
        extras = nipals_extras(ctx, mdl, X, Y, n, p, barrier_sync)
    if dist is not None and not a.no_extras:
        extras["allreduce"] = time_allreduce(ctx, mbpls_amd.get_comm(), n * B + 3 * B + B * 40)
        extras["strong_scaling_c2"] = strong_scaling(ctx, mbpls_amd, dT, Y, r, rank, world, n, a.p, K, a.method,
                                                     barrier_sync, max_over_ranks)

    # (the reference's own count: its loop makes two passes per component for one response whatever this build does)
    one_pass_route = "fixed point after one pass" in str(getattr(mdl, "fit_plan_", {}).get("solver", ""))
    ref_iters = [2] * len(iters) if one_pass_route else iters
    ref_units = sum(2 * i + 3 for i in ref_iters) + 1 + 3
    # X traffic the HIP path really generates per fit = sum of the kernels' algorithmic bytes (HIP-event
    # profile of the timed region); with lazy deflation every pass is read-only
    fused_units, unit_bytes, per_step, achieved = job_throughput(prof, n, p, world, a.steps, dt)

    if rank == 0:
        dom = max(prof.items(), key=lambda kv: kv[1]["total_ms"]) if prof else (None, None)
        roof = None
        if dom[0]:
            k = dom[1]
            traffic, src = pmc_traffic(dom[0], n, p, live_pmc)
            ceiling = X.probe_read_bandwidth(3)      # best pure-read kernel on this box, same buffer (after the timed region)
            roof = roofline_of(dom[0], k, n, p, world, traffic, src, ceiling)
        out = {
            "metric": "mbpls_fit_achieved_hbm_GBps", "value": round(achieved, 1), "unit": "GB/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(per_step * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (planted latent structure, generated on device)",
            "config": {"workload": (f"single-block PLS NIPALS n={n} p={p} K={K} q={q} f64 (BASELINE config[1])"
                                    if dist is None else
                                    f"multiblock NIPALS n={n} total p={p * world} ({B} blocks) K={K} q={q} f64, "
                                    f"feature columns sharded over {world} GPUs (BASELINE config[4])"),
                       "method": a.method, "standardize": True, "calc_all": True, "nipals_iterations": iters,
                       "route": getattr(mdl, "fit_plan_", {}).get("solver")},
            "fit_wall_s": round(per_step, 5), "fits_per_s": round(1.0 / per_step, 4),
            "fit_wall_s_median": round(float(np.median(per_fit)), 5),
            "fit_wall_s_max": round(float(np.max(per_fit)), 5),
            "first_fit_wall_s": None if not cold else cold.get("first_fit_wall_s"),
            "cold_start": cold,
            "estimator_per_step": "fresh (a new MBPLS object for every timed fit, warm process)",
            "fit_wall_s_same_estimator_refit": round(float(np.median(refit)), 5),
            "fit_wall_s_each": [round(t, 5) for t in per_fit],
            "units_moved_per_fit": fused_units, "unit_bytes": unit_bytes,
            "frac_of_hbm_peak_whole_fit": round(achieved / (HBM_PEAK_GBPS * world), 4),
            "effective_GBps_reference_algorithm_bytes": round(ref_units * unit_bytes / per_step / 1e9, 1),
            "ref_units_per_fit": ref_units,
            "roofline": roof,
            "pmc_traffic_this_run": live_pmc,
            "kernels": {k: {"launches": v["launches"], "avg_ms": round(v["avg_ms"], 4), "GBps": round(v["GBps"], 1)}
                        for k, v in prof.items()},
            "host_phases_ms_last_fit": phases,
            "host_phases_ms_slowest_fit": {k: round(1e3 * v, 2)
                                           for k, v in per_fit_phases[int(np.argmax(per_fit))].items()},
            "comm": comm_info, "sharded_self_check": check,
        }
        out.update(extras)
    if e2e_from_child is not None:
        if rank == 0:
            out["end_to_end_host_numpy"] = e2e_from_child
    elif dist is None and a.e2e_cols != 0 and (a.no_extras or a.extras_in_process):
        note("end-to-end leg (host numpy in)")
        e2e = end_to_end(ctx, X, Y, n, p if a.e2e_cols < 0 else min(a.e2e_cols, p), K, a.method)
        if rank == 0:
            out["end_to_end_host_numpy"] = e2e
    if rank == 0:
        if a.cpu_cols > 0:
            note("CPU baseline (oracle on the host cores)")
        if a.cpu_cols > 0 and dist is None:
            out["cpu_baseline"] = cpu_baseline(ctx, X, Y, n, p, min(a.cpu_cols, p), K, a.method, per_step)
        elif a.cpu_cols > 0:
            # N > 1: rank 0 times the oracle on a smaller sample of ITS columns (the other ranks wait at the barrier
            # below); the GPU-vs-oracle comparison of this configuration is sharded_self_check, run before the timing
            out["cpu_baseline"] = cpu_baseline(ctx, X, Y, n, p * world, min(a.cpu_cols // 2, p), K, a.method, per_step,
                                               gpu_check=False)
        else:
            out["cpu_baseline"] = None
    X.close()
    if dist is not None:
        dist.barrier()
        comm = mbpls_amd.get_comm()
        if hasattr(comm, "close"):
            comm.close()
        dist.destroy_process_group()
    # the ONE JSON line is the last thing on stdout: librccl writes a version banner through C stdio, which would
    # otherwise be flushed behind it at exit
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if rank == 0:
        print(json.dumps(out), flush=True)


# ------------------------------------------------------------------------------------------------ N > 1 helpers
def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def count_gpus_without_hip():
    """GPUs of this node, counted WITHOUT initialising the HIP / HSA runtime in this process (the launcher must stay a
    GPU-free parent of the ranks): the KFD topology in sysfs (a node with simd_count > 0 is a GPU), else a throw-away
    child process that asks the runtime; None when neither works (the caller then skips its check)."""
    import glob
    import subprocess
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if nodes:
        n = 0
        for f in nodes:
            try:
                props = dict(line.split()[:2] for line in open(f).read().splitlines() if len(line.split()) >= 2)
                if int(props.get("simd_count", "0")) > 0:
                    n += 1
            except (OSError, ValueError):
                pass
        hide = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES"))
        if hide is not None and hide.strip() != "":
            n = min(n, len([x for x in hide.split(",") if x.strip() != ""]))
        return n
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                             capture_output=True, text=True, timeout=300)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:
        return None


def _bind_prctl():
    """libc's prctl, resolved ONCE in the launcher before any thread or rank exists: the preexec of a rank runs between
    fork and exec, where an import or a dlopen can find a lock held by one of the launcher's pump threads"""
    try:
        import ctypes
        fn = ctypes.CDLL("libc.so.6", use_errno=True).prctl
        fn.argtypes = [ctypes.c_int, ctypes.c_ulong, ctypes.c_ulong, ctypes.c_ulong, ctypes.c_ulong]
        fn.restype = ctypes.c_int
        return fn
    except Exception:
        return None


def _die_with_parent(prctl, launcher_pid):
    """preexec of a rank: SIGTERM when the launcher dies (PR_SET_PDEATHSIG) -- a killed launcher leaves no rank behind
    in a collective with the GPUs leased.  Only calls what was bound before the fork; and if the launcher died between
    the fork and the prctl (the signal would never come), the rank gives up at once."""
    def preexec():
        if prctl is not None:
            prctl(1, 15, 0, 0, 0)                  # PR_SET_PDEATHSIG, SIGTERM
            if os.getppid() != launcher_pid:
                os._exit(1)
    return preexec


DEFAULT_LAUNCH_TIMEOUT_S = 3000.0


def launch_ranks(world, argv, timeout=DEFAULT_LAUNCH_TIMEOUT_S, python=None):
    """Start `world` child processes `python argv...`, one rank each (RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 / MASTER_PORT in the environment -- what torch.distributed.run would export), wait for them,
    print the LAST line of rank 0's stdout that parses as JSON as the last line of this process's stdout, and return
    the exit status: 0 only when every rank returned 0.  Everything else the ranks print goes to stderr, prefixed.
    The first rank that fails takes the others down (they would otherwise wait in a collective); so does a SIGTERM /
    SIGINT / SIGHUP sent to the launcher, its own death (PR_SET_PDEATHSIG in the ranks), a timeout, and any exception
    on the way out (try / finally): every rank started here is signalled, then killed after a grace period.  The
    caller itself never touches the GPU, so the children are ordinary fresh processes (no exec from a GPU-initialised
    process).  When the node shows fewer GPUs than ranks and MBPLS_BENCH_SHARED_GPU is not set, nothing is started."""
    import signal
    import subprocess
    import threading
    python = python or sys.executable
    if os.environ.get("MBPLS_BENCH_SHARED_GPU") != "1" and os.environ.get("MBPLS_BENCH_SKIP_GPU_COUNT") != "1":
        have = count_gpus_without_hip()
        if have is not None and have < world:
            print(f"[bench] --gpus {world} but only {have} GPU(s) visible (MBPLS_BENCH_SHARED_GPU=1 runs the ranks on "
                  f"one GPU over gloo as a dry run)", file=sys.stderr, flush=True)
            return 2
    env = dict(os.environ)
    env.update(WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(env.get("MASTER_PORT") or free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.setdefault("OMP_NUM_THREADS", "1")
    procs, lines0, pumps = [], [], []

    def pump(stream, rank, keep):
        for raw in iter(stream.readline, b""):
            text = raw.decode(errors="replace").rstrip("\n")
            if keep is not None:
                keep.append(text)
            else:
                print(f"[rank {rank}] {text}", file=sys.stderr, flush=True)

    rcs = [None] * world

    def take_down(grace=10.0):
        """SIGTERM to the session of every rank still running, SIGKILL after the grace period"""
        live = [(r, pr) for r, pr in enumerate(procs) if rcs[r] is None and pr.poll() is None]
        for _, pr in live:
            try:
                os.killpg(pr.pid, signal.SIGTERM)
            except (ProcessLookupError, PermissionError):
                pass
        deadline = time.time() + grace
        for r, pr in live:
            try:
                rcs[r] = pr.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except (ProcessLookupError, PermissionError):
                    pass
                rcs[r] = pr.wait()
        for r, pr in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = pr.poll()

    got_signal = []
    previous = {}

    def on_signal(signum, frame):
        got_signal.append(signum)

    in_main_thread = threading.current_thread() is threading.main_thread()
    if in_main_thread:
        for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
            previous[sg] = signal.signal(sg, on_signal)
    failed = None
    timed_out = False
    try:
        preexec = _die_with_parent(_bind_prctl(), os.getpid())
        for r in range(world):
            e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
            pr = subprocess.Popen([python] + list(argv), env=e, stdout=subprocess.PIPE, stderr=None,
                                  start_new_session=True, preexec_fn=preexec)
            th = threading.Thread(target=pump, args=(pr.stdout, r, lines0 if r == 0 else None), daemon=True)
            th.start()
            procs.append(pr)
            pumps.append(th)
        t0 = time.time()
        while any(rc is None for rc in rcs):
            for r, pr in enumerate(procs):
                if rcs[r] is None:
                    rcs[r] = pr.poll()
                    if rcs[r] not in (None, 0) and failed is None:
                        failed = r
            timed_out = timeout is not None and time.time() - t0 > timeout
            if failed is not None or timed_out or got_signal:
                take_down()
                break
            time.sleep(0.05)
    finally:
        take_down(grace=5.0)                          # nothing started here survives this call, whatever happened
        if in_main_thread:
            for sg, h in previous.items():
                signal.signal(sg, h)
    for th in pumps:
        th.join(timeout=5)
    if got_signal:
        print(f"[bench] launcher received signal {got_signal[0]}: ranks taken down, exit codes {rcs}", file=sys.stderr,
              flush=True)
        return 128 + int(got_signal[0])
    line = None
    for text in lines0:
        try:
            if text.lstrip().startswith("{") and isinstance(json.loads(text), dict):
                line = text
                continue
        except ValueError:
            pass
        print(f"[rank 0] {text}", file=sys.stderr, flush=True)
    ok = all(rc == 0 for rc in rcs)
    if not ok:
        why = f"rank {failed} exited with {rcs[failed]}" if failed is not None else f"timeout after {timeout} s"
        print(f"[bench] {why}; exit codes {rcs}", file=sys.stderr, flush=True)
    if line is not None and ok:
        print(line, flush=True)
    elif ok:
        print("[bench] rank 0 printed no JSON line", file=sys.stderr, flush=True)
        return 1
    if ok:
        return 0
    bad = rcs[failed] if failed is not None else next((rc for rc in rcs if rc), 1)
    return bad if 0 < bad < 256 else 1


def job_throughput(prof, n, p, world, steps, dt):
    """``value`` of the line: the X bytes the kernels of one fit really move -- the sum of their algorithmic bytes from
    the HIP-event profile of the timed region, in units of one traversal of THIS RANK's shard (8 n p bytes, p = columns
    per GPU) -- times the number of ranks (weak scaling: every rank moves the same share; the timed region is bracketed by
    barriers and ``dt`` is the maximum over ranks), divided by the time per fit.
    Returns (units per fit, bytes of one traversal of the WHOLE matrix = 8 n (world p), seconds per fit, GB/s)."""
    fused_units = round(sum(v["bytes_per_launch"] * v["launches"] for v in prof.values()) / steps / (8.0 * n * p), 2)
    unit_bytes = 8.0 * n * p * world
    per_step = dt / steps
    return fused_units, unit_bytes, per_step, fused_units * unit_bytes / per_step / 1e9


def roofline_of(name, k, n, p, world, traffic, src, ceiling):
    """``roofline`` of the line for the dominant kernel ``k`` (profile entry of rank 0).  achieved / peak / frac are PER
    GPU: one launch on one GPU moves ``bytes_per_launch`` = its algorithmic bytes over this rank's shard against that
    GPU's 8 TB/s.  ``whole_job`` states the same launch over all ranks (world x the bytes, world x the peak; the
    fraction is the same number), so that an N > 1 line can be read either way without guessing which p was meant."""
    roof = {"bound": "hbm", "kernel": name, "achieved": round(k["GBps"], 1), "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": round(k["GBps"] / HBM_PEAK_GBPS, 4), "traffic": traffic,
            "measured_read_ceiling_GBps": round(ceiling, 1),
            "frac_of_measured_read_ceiling": round(k["GBps"] / ceiling, 4) if ceiling > 0 else None,
            "traffic_source": src, "bytes_per_launch": k["bytes_per_launch"], "avg_ms": round(k["avg_ms"], 4),
            "launches": k["launches"],
            "timing": "HIP events on the launch stream around every launch of the timed region",
            "scope": "per GPU (rank 0's launches over its own shard of p columns)",
            "whole_job": {"n_gpus": world, "columns": p * world, "bytes_per_launch": k["bytes_per_launch"] * world,
                          "achieved": round(k["GBps"] * world, 1), "peak": HBM_PEAK_GBPS * world}}
    return roof


def sharded_self_check(ctx, mbpls_amd, dT, Y, r, rank, world):
    """A small 3-block problem fitted (a) column-sharded over all ranks through the communicator and (b) unsharded on
    this rank's own GPU: every rank compares its shard of (a) with (b).  Both are the product path -- this checks the
    exchange, not the kernels.  Exits the process non-zero on a mismatch, before anything is timed."""
    from mbpls_amd import MBPLS, DeviceMatrix, LocalComm
    from mbpls_amd.comm import shard_columns
    n = Y.shape[0]
    sizes = [1000 * world, 700 * world + 3, 1300 * world + 1]        # ragged: the shards differ in width
    goffs = np.concatenate(([0], np.cumsum(sizes)))
    full = DeviceMatrix(ctx, n, int(goffs[-1]), [int(o) for o in goffs])
    full.fill_latent(dT, r, 0.5, 99, 0)
    ranges = shard_columns(sizes, rank, world)
    loffs = np.concatenate(([0], np.cumsum([hi - lo for lo, hi in ranges])))
    local = DeviceMatrix(ctx, n, int(loffs[-1]), [int(o) for o in loffs])
    for b, (lo, hi) in enumerate(ranges):
        view = DeviceMatrix(ctx, n, hi - lo, None, storage=local.data_ptr + 8 * local.ld * int(loffs[b]))
        view.fill_latent(dT, r, 0.5, 99, int(goffs[b]) + lo)
        view.close()
    ctx.sync()
    worst, worst_oracle = 0.0, None
    comm = mbpls_amd.get_comm()
    Xh = full.download() if rank == 0 else None      # rank 0 also checks its shard against the CPU oracle
    # (NIPALS on both routes of one response: one pass per component -- the default -- and the reference's two passes,
    # whose stop decision every rank must take alike)
    for method, q1_route in (("NIPALS", "1"), ("NIPALS", "0"), ("UNIPALS", "1")):
        os.environ["MBPLS_NIPALS_Q1_FIXED_POINT"] = q1_route

        def iterations_differ(fitted, reference):
            if method != "NIPALS":
                return False
            got, want = fitted.nipals_iterations_, reference.nipals_iterations_
            if q1_route == "1":
                return list(got) != [1] * 4 or list(want) not in ([1] * 4, [2] * 4)
            return list(got) != list(want)

        sharded = MBPLS(n_components=4, method=method).fit(local, Y)
        if rank == 0:
            from oracle import mbpls_oracle as orc     # checker only, nothing timed here
            blocks = [Xh[:, int(goffs[b]):int(goffs[b + 1])] for b in range(len(sizes))]
            o = orc.fit(blocks, Y, n_components=4, method=method, faithful=True)
            so = np.sign(np.sum(o.Ts_ * sharded.Ts_, axis=0))
            eo = [np.max(np.abs(sharded.Ts_ * so - o.Ts_)) / np.max(np.abs(o.Ts_)),
                  np.max(np.abs(sharded.U_ * so - o.U_)) / np.max(np.abs(o.U_))]
            for b, (lo, hi) in enumerate(ranges):
                eo.append(np.max(np.abs(sharded.P_[b] * so - o.P_[b][lo:hi])) / np.max(np.abs(o.P_[b])))
                eo.append(np.max(np.abs(sharded.W_[b] * so - o.W_[b][lo:hi])) / np.max(np.abs(o.W_[b])))
            gl = np.concatenate([np.arange(int(goffs[b]) + lo, int(goffs[b]) + hi) for b, (lo, hi) in enumerate(ranges)])
            eo.append(np.max(np.abs(sharded.beta_ - o.beta_[gl])) / np.max(np.abs(o.beta_)))
            if iterations_differ(sharded, o):
                eo.append(1.0)
            worst_oracle = max(worst_oracle or 0.0, float(max(eo)))
        mbpls_amd.set_comm(LocalComm())
        try:
            ref = MBPLS(n_components=4, method=method).fit(full, Y)
        finally:
            mbpls_amd.set_comm(comm)
        s = np.sign(np.sum(ref.Ts_ * sharded.Ts_, axis=0))
        errs = [np.max(np.abs(sharded.Ts_ * s - ref.Ts_)) / np.max(np.abs(ref.Ts_))]
        for b, (lo, hi) in enumerate(ranges):
            errs.append(np.max(np.abs(sharded.P_[b] * s - ref.P_[b][lo:hi])) / np.max(np.abs(ref.P_[b])))
        glob = np.concatenate([np.arange(int(goffs[b]) + lo, int(goffs[b]) + hi) for b, (lo, hi) in enumerate(ranges)])
        errs.append(np.max(np.abs(sharded.beta_ - ref.beta_[glob])) / np.max(np.abs(ref.beta_)))
        if iterations_differ(sharded, ref):
            errs.append(1.0)
        worst = max(worst, float(max(errs)))
    os.environ.pop("MBPLS_NIPALS_Q1_FIXED_POINT", None)
    full.close(); local.close()
    ok = np.isfinite(worst) and worst < 1e-8
    if worst_oracle is not None:
        ok = ok and np.isfinite(worst_oracle) and worst_oracle < 1e-6     # north_star: 1e-6 relative, sign-aligned
    print(f"[bench] rank {rank}/{world}: sharded fit vs unsharded fit, worst rel. error {worst:.2e}"
          + (f"; vs the CPU oracle {worst_oracle:.2e}" if worst_oracle is not None else "")
          + f" ({'ok' if ok else 'MISMATCH'})", file=sys.stderr, flush=True)
    if not ok:
        os._exit(3)
    return {"worst_rel_err": worst, "worst_rel_err_vs_oracle_rank0": worst_oracle,
            "methods": ["NIPALS (one pass per component)", "NIPALS (two passes per component)", "UNIPALS"],
            "blocks": sizes, "checked": "Ts_, U_, P_, W_, beta_ shard slices, nipals_iterations_"}


def time_allreduce(ctx, comm, count, reps=200):
    """mean time of one all-reduce of the exchange buffer (the per-pass collective), back to back on the stream"""
    buf = comm.exchange_buffer(ctx, count)
    for _ in range(10):
        comm.allreduce_device(buf)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        comm.allreduce_device(buf)
    if hasattr(buf, "tensor"):
        import torch
        torch.cuda.synchronize()
    ctx.sync()
    us = (time.perf_counter() - t0) / reps * 1e6
    if hasattr(buf, "free"):
        buf.free()
    return {"doubles": int(count), "mean_us": round(us, 1), "per_fit": "one per pass over X (K + 1 per fit on the default route, 2K + 1 with two passes per component)"}


def strong_scaling(ctx, mbpls_amd, dT, Y, r, rank, world, n, p_total, K, method, barrier_sync, max_over_ranks):
    """BASELINE config[1] (ONE block of p_total columns) split over the ranks: the same problem as the 1-GPU line"""
    from mbpls_amd import MBPLS, DeviceMatrix
    from mbpls_amd.comm import shard_columns
    (lo, hi), = shard_columns([p_total], rank, world)
    Xs = DeviceMatrix(ctx, n, hi - lo, [0, hi - lo])
    Xs.fill_latent(dT, r, 0.5, 1234, lo)
    ctx.sync()
    mdl = MBPLS(n_components=K, method=method)
    for _ in range(2):
        mdl.fit(Xs, Y)
    barrier_sync()
    t0 = time.perf_counter()
    steps = 5
    for _ in range(steps):
        mdl.fit(Xs, Y)
    barrier_sync()
    dt = max_over_ranks(time.perf_counter() - t0) / steps
    Xs.close()
    return {"workload": f"single-block NIPALS n={n} p={p_total} K={K} split over {world} GPUs", "fit_wall_s": round(dt, 5),
            "fits_per_s": round(1.0 / dt, 3), "nipals_iterations": list(getattr(mdl, "nipals_iterations_", []))}


def other_solvers(ctx, X, n, K, r):
    """The secondary configurations of SURVEY 8d on the SAME resident matrix (never `value`): ten responses (the paper
    protocol's q) with NIPALS and UNIPALS, SIMPLS and UNIPALS with one response.  A fresh estimator per fit, one
    warm-up and three timed fits each; passes = launches of X-sized kernels per fit (HIP-event profile)."""
    from mbpls_amd import MBPLS
    out = {}
    # (KERNEL on both of its p > n routes: the default at this shape -- maintained thin products, one pass per
    # component -- and the association matrix on the f64 matrix cores, forced; the Gram's TFLOP/s beside it)
    for name, method, q, env in (("nipals_q10", "NIPALS", 10, {}), ("unipals_q10", "UNIPALS", 10, {}),
                                 ("simpls_q1", "SIMPLS", 1, {}), ("unipals_q1", "UNIPALS", 1, {}),
                                 ("kernel_q1", "KERNEL", 1, {}), ("kernel_q1_gram", "KERNEL", 1, {"MBPLS_KERNEL_GRAM": "1"})):
        note(f"  {name}")
        Yq = planted(n, q, r, 0)[1]                      # the same latent scores as X (seed 0), q response columns
        os.environ.update(env)
        try:
            MBPLS(n_components=K, method=method).fit(X, Yq)
            ctx.profile(True); ctx.profile_reset()
            walls, keep = [], []
            for _ in range(3):
                ctx.sync()
                t1 = time.perf_counter()
                keep.append(MBPLS(n_components=K, method=method).fit(X, Yq))
                ctx.sync()
                walls.append(time.perf_counter() - t1)
            pr = ctx.profile_get(); ctx.profile(False)
            m = keep[-1]
            out[name] = {"fit_wall_s": round(float(np.median(walls)), 5),
                         "passes_per_fit": sum(v["launches"] for v in pr.values()) // 3,
                         "sum_of_passes_ms": round(sum(v["total_ms"] for v in pr.values()) / 3, 2),
                         "solver": getattr(m, "fit_plan_", {}).get("solver"),
                         "standardisation": getattr(m, "fit_plan_", {}).get("standardisation")}
            if hasattr(m, "nipals_iterations_"):
                out[name]["nipals_iterations_total"] = int(sum(m.nipals_iterations_))
            if "gram" in pr:
                fl = float(n) * (n + 1) * X.p                # all Gram launches of one fit cover the p columns once
                out[name]["gram_TFLOPs"] = round(fl / (pr["gram"]["total_ms"] / 3 * 1e-3) / 1e12, 2)
                out[name]["gram_frac_of_f64_mfma_peak_78.6"] = round(out[name]["gram_TFLOPs"] / 78.6, 3)
            del keep, m
        except Exception as e:                           # noqa: BLE001 -- a side leg must not cost the line
            ctx.profile(False)
            out[name] = {"skipped": f"{type(e).__name__}: {e}"}
        finally:
            for key in env:
                os.environ.pop(key, None)
    return out


def shard_proxy(ctx, mbpls_amd, dT, Y, r, n, p_shard, K, full_fit_s):
    """What ONE of eight GPUs would do for this problem split over eight (strong scaling): the same fit on p / 8 columns
    through the sharded code path -- the C ABI's RCCL communicator with ONE rank, ncclAllReduce of the exchange buffer
    on the kernels' stream behind every pass.  Reports the fit, the sum of its passes and what is left (the fixed cost
    per fit: steps, chunk reductions, all-reduces, end-of-fit algebra and copies) -- the part that does not shrink with
    the shard and therefore caps the speed-up.  A one-rank all-reduce has no wire latency: a lower bound of the fixed
    cost, not a prediction for eight ranks."""
    from mbpls_amd import MBPLS, DeviceMatrix, LocalComm, RcclComm
    out = {"workload": f"single-block NIPALS n={n} p={p_shard} K={K} (this problem / 8), RCCL world size 1"}
    try:
        comm = RcclComm(ctx, 0, 1, lambda payload: payload)
    except Exception as e:                               # noqa: BLE001 -- librccl missing: say so, keep the line
        out["skipped"] = f"{type(e).__name__}: {e}"
        return out
    Xs = DeviceMatrix(ctx, n, p_shard, [0, p_shard])
    try:
        Xs.fill_latent(dT, r, 0.5, 1234, 0)
        ctx.sync()
        mbpls_amd.set_comm(comm)
        for _ in range(3):
            MBPLS(n_components=K).fit(Xs, Y)
        # the fit as it runs in production first (no events in the stream), then the same fits with HIP events around
        # every pass for their durations: the events themselves cost a shard-sized fit 0.1-0.2 ms (42 records, each a
        # barrier packet between two launches), which is the measurement's cost, not the fit's
        steps, keep = 10, []

        def timed_fits():
            walls = []
            for _ in range(steps):
                ctx.sync()
                t1 = time.perf_counter()
                keep.append(MBPLS(n_components=K).fit(Xs, Y))
                walls.append(time.perf_counter() - t1)
            return float(np.median(walls))

        wall = timed_fits()
        ctx.profile(True); ctx.profile_reset()
        wall_prof = timed_fits()
        pr = ctx.profile_get(); ctx.profile(False)
        passes = sum(v["total_ms"] for v in pr.values()) / steps
        out.update({"fit_wall_s": round(wall, 5), "fit_wall_with_events_s": round(wall_prof, 5),
                    "sum_of_passes_ms": round(passes, 3),
                    "fixed_ms_per_fit": round(1e3 * wall - passes, 3),
                    "fixed_ms_per_fit_with_events": round(1e3 * wall_prof - passes, 3),
                    "passes": {k: {"launches": v["launches"] // steps, "avg_ms": round(v["avg_ms"], 4)} for k, v in pr.items()},
                    "nipals_iterations": list(keep[-1].nipals_iterations_),
                    "ratio_full_fit_over_shard_fit": round(full_fit_s / wall, 2),
                    "note": "ratio = upper bound of the 8-GPU strong-scaling speed-up (all-reduce wire latency not included)"})
    finally:
        mbpls_amd.set_comm(LocalComm())
        comm.close()
        Xs.close()
    return out


# ------------------------------------------------------------------------------------------------ extras (N = 1)
def nipals_extras(ctx, mdl, X, Y, n, p, barrier_sync):
    """Outside the timed region, never part of `value`:
    * explicit deflation (MBPLS_DEFLATION=explicit): X is standardised into a working copy and rewritten in place by
      every deflation -- the read+write deflation pass north_star sets its >= 60 % target on;
    * the reference's iteration structure for one response: two passes per component, the second re-deriving the first
      until ||ts_old - ts|| <= max_tol (mbpls.py:841) -- what the default route (one pass per component) leaves out."""
    from mbpls_amd import MBPLS
    out = {}
    os.environ["MBPLS_DEFLATION"] = "explicit"
    try:
        m2 = MBPLS(n_components=mdl.n_components, method="NIPALS")
        m2.fit(X, Y)
        m2.fit(X, Y)
        ctx.profile(True); ctx.profile_reset()
        barrier_sync()
        t1 = time.perf_counter()
        for _ in range(3):
            m2.fit(X, Y)
        barrier_sync()
        wall = (time.perf_counter() - t1) / 3
        pr = ctx.profile_get(); ctx.profile(False)
        d = pr.get("deflate_sweep")
        out["explicit_deflation"] = {
            "fit_wall_s": round(wall, 5),
            "deflate_sweep_read_write": None if d is None else {
                "avg_ms": round(d["avg_ms"], 4), "bytes_per_launch": d["bytes_per_launch"], "GBps": round(d["GBps"], 1),
                "frac_of_hbm_peak": round(d["GBps"] / HBM_PEAK_GBPS, 4), "launches": d["launches"]},
            "kernels": {k: {"launches": v["launches"], "avg_ms": round(v["avg_ms"], 4), "GBps": round(v["GBps"], 1)}
                        for k, v in pr.items()},
            "note": "loadings + in-place rank-1 deflation + next sweep, 16np bytes per launch (mbpls.py:928+:969+:856)"}
    finally:
        del os.environ["MBPLS_DEFLATION"]
    os.environ["MBPLS_NIPALS_Q1_FIXED_POINT"] = "0"
    try:
        m3 = MBPLS(n_components=mdl.n_components, method="NIPALS")
        m3.fit(X, Y)
        ctx.profile(True); ctx.profile_reset()
        walls = []
        for _ in range(3):
            barrier_sync()
            t1 = time.perf_counter()
            m3.fit(X, Y)
            barrier_sync()
            walls.append(time.perf_counter() - t1)
        pr = ctx.profile_get(); ctx.profile(False)
        out["reference_iteration_structure"] = {
            "fit_wall_s": round(float(np.median(walls)), 5), "nipals_iterations": list(m3.nipals_iterations_),
            "passes_per_fit": sum(v["launches"] for v in pr.values()) // 3,
            "route": getattr(m3, "fit_plan_", {}).get("solver"),
            "kernels": {k: {"launches": v["launches"], "avg_ms": round(v["avg_ms"], 4), "GBps": round(v["GBps"], 1)}
                        for k, v in pr.items()},
            "note": "MBPLS_NIPALS_Q1_FIXED_POINT=0: two NIPALS passes per component as the reference makes them; same "
                    "attributes (tests), not used for value/roofline"}
    finally:
        del os.environ["MBPLS_NIPALS_Q1_FIXED_POINT"]
    return out


def end_to_end(ctx, X, Y, n, cols, K, method):
    """fit() from a HOST numpy array (C-order, what a user of the reference holds): chunked H2D with the
    transpose to feature-major on the GPU, fit, attributes back on the host.  PCIe-inclusive, never `value`."""
    from mbpls_amd import MBPLS
    try:
        avail = None
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable"):
                    avail = int(line.split()[1]) * 1024
        need = 8.0 * n * cols
        if avail is not None and need * 1.5 > avail:
            cols = int(avail / 1.5 / (8.0 * n)) // 1024 * 1024
            if cols < 1024:
                return {"skipped": "not enough host memory"}
        t0 = time.perf_counter()
        Xh = X.download(0, cols)                      # C-order n x cols host array (the generator's output)
        t_down = time.perf_counter() - t0
        m = MBPLS(n_components=K, method=method)
        times = []
        for _ in range(2):
            t0 = time.perf_counter()
            m.fit(Xh, Y)
            times.append(time.perf_counter() - t0)
        ph = {k: round(1e3 * v, 1) for k, v in m.fit_timings_.items()}
        gb = 8.0 * n * cols / 1e9
        return {"fit_wall_s": round(min(times), 4), "fit_wall_s_each": [round(t, 4) for t in times], "p": cols, "host_GB": round(gb, 2),
                "includes": "H2D (chunked, pageable host memory) + GPU transpose to feature-major + fit + D2H of the attributes",
                "upload_GBps": round(gb / (ph.get("load", 0.0) / 1e3), 1) if ph.get("load") else None,
                "host_phases_ms": ph, "setup_download_s": round(t_down, 2)}
    except MemoryError:
        return {"skipped": "not enough host memory"}


PASS_NAMES = {("true", "false", "true"): "standardize_sweep", ("false", "false", "true"): "sweep",
              ("false", "true", "true"): "deflate_sweep", ("false", "true", "false"): "deflate"}


def parse_pmc_csv(path, counter):
    """average `counter` value per launch of every register-resident column pass (k_colpass<R, TMAX, D, STD, DEFL,
    SWEEP, ...>) in a rocprofv3 counter_collection.csv, keyed by the pass names of the HIP-event profile"""
    import collections
    import csv
    agg = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            kn = r.get("Kernel_Name", "")
            if "k_colpass<" not in kn or r.get("Counter_Name") != counter:
                continue
            args = [x.strip() for x in kn.split("<", 1)[1].split(">")[0].split(",")]
            nm = PASS_NAMES.get(tuple(args[3:6]))
            if nm:
                agg[nm].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def run_pmc_children(a, timeout_s=120):
    """Two child runs of ONE fit of this workload under `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace` (separate
    passes; no other trace domain), from /tmp with TMPDIR=/tmp as the profiler wants.  Unit and correction exactly as
    MI355X_MICROARCH.md section HBM prescribes: the derived counters are in KiB; on gfx950 FETCH_SIZE tallies the
    128-byte requests of a wide coalesced (16 B/lane) stream as 64 bytes, so read bytes = 2 x FETCH_SIZE x 1024, and
    WRITE_SIZE x 1024 is exact for 16 B/lane streaming stores.  Returns {kernel: {...}} or {"error": ...}; never fatal."""
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    out = {"method": "rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in separate child runs of one fit of this workload "
                     "(--kernel-trace only); read bytes = 2 x FETCH_SIZE x 1024 (gfx950 correction for wide coalesced "
                     "streams, MI355X_MICROARCH.md section HBM), write bytes = WRITE_SIZE x 1024", "kernels": {}}
    tmp = tempfile.mkdtemp(prefix="mbpls_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    t0 = time.time()
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            note(f"PMC pass {counter} (child process under rocprofv3)")
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--cpu-cols", "0",
                   "--e2e-cols", "0", "--no-extras", "--no-first-fit", "--no-pmc", "--n", str(a.n), "--p", str(a.p),
                   "--K", str(a.K), "--q", str(a.q), "--method", a.method]
            pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                                  start_new_session=True)
            try:
                _, err = pr.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except OSError:
                    pass
                pr.wait()
                return {"error": f"{counter} pass timed out after {timeout_s} s"}
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if pr.returncode != 0 or not files:
                return {"error": f"{counter} pass: exit {pr.returncode}, {len(files)} counter files: {(err or '')[-300:]}"}
            for nm, (avg, cnt) in parse_pmc_csv(files[0], counter).items():
                k = out["kernels"].setdefault(nm, {})
                k[counter + "_KiB_avg"] = avg
                k["launches"] = cnt
            if "arch" not in out:
                out["arch"] = profiled_arch(d)
        # the x 2 on FETCH_SIZE is a property of gfx950's counter (MI355X_MICROARCH.md): anywhere else the raw counter is
        # reported and labelled as uncorrected
        fetch_factor = 2.0 if out.get("arch") == "gfx950" else 1.0
        if fetch_factor == 1.0:
            out["method"] += f" -- UNCORRECTED here: the profiled device is {out.get('arch') or 'unknown'}, not gfx950"
        unit = 8.0 * a.n * a.p
        for nm, k in list(out["kernels"].items()):
            if "FETCH_SIZE_KiB_avg" not in k or "WRITE_SIZE_KiB_avg" not in k:
                del out["kernels"][nm]
                continue
            k["hbm_read_bytes"] = fetch_factor * k["FETCH_SIZE_KiB_avg"] * 1024.0
            k["hbm_write_bytes"] = k["WRITE_SIZE_KiB_avg"] * 1024.0
            k["hbm_bytes_per_launch"] = k["hbm_read_bytes"] + k["hbm_write_bytes"]
            k["over_algorithmic_8np"] = round(k["hbm_bytes_per_launch"] / unit, 4)
        out["wall_s"] = round(time.time() - t0, 1)
        if not out["kernels"]:
            return {"error": "no column-pass kernels in the counter files"}
        return out
    except Exception as e:                               # noqa: BLE001 -- the committed figure is reported instead
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def profiled_arch(profile_dir):
    """gcnArchName of the device a rocprofv3 run profiled: from the agent table the profiler writes beside its counter
    files, else from rocm_agent_enumerator (a child process: this one has not touched the GPU yet)"""
    import glob
    import re
    import shutil
    import subprocess
    for f in glob.glob(os.path.join(profile_dir, "**", "*agent_info.csv"), recursive=True):
        try:
            m = re.search(r"gfx[0-9a-f]{3,4}", open(f).read())
        except OSError:
            m = None
        if m:
            return m.group(0)
    exe = shutil.which("rocm_agent_enumerator") or "/opt/rocm/bin/rocm_agent_enumerator"
    try:
        txt = subprocess.run([exe], capture_output=True, text=True, timeout=30).stdout
        archs = [ln.strip() for ln in txt.splitlines() if ln.strip().startswith("gfx") and ln.strip() != "gfx000"]
        return archs[0] if archs else None
    except Exception:                                     # noqa: BLE001
        return None


def pmc_traffic(kernel, n, p, live=None):
    """HBM bytes per launch of `kernel`: from the PMC passes of THIS run (run_pmc_children) when they succeeded, else
    from the COMMITTED rocprofv3 PMC passes (same command, taken on the builder's lease, labelled so) -- and those only
    when they were taken on this exact shape, else None."""
    if live and "kernels" in live and kernel in live["kernels"]:
        return (live["kernels"][kernel]["hbm_bytes_per_launch"],
                "measured in this run: two rocprofv3 --pmc child passes (FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024, "
                "MI355X_MICROARCH.md section HBM), per launch of this kernel")
    for name in ("r05_pmc_traffic_c2.json", "r04_pmc_traffic_c2.json", "r03_pmc_traffic_c2.json", "r02_pmc_traffic_c2.json",
                 "r01_pmc_traffic_c2.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            if d["config"]["n"] == n and d["config"]["p"] == p:
                return d["kernels"][kernel]["hbm_bytes_per_launch"], f"profiles/{name} (PMC passes of an earlier run of this command, not measured in this run)"
        except Exception:
            continue
    return None, None


def side_legs(ctx, mbpls_amd, a, mdl, X, Y, dT, r, n, p, K, q, single_block, barrier_sync, full_fit_s):
    """the side legs of the N = 1 line (never `value`): explicit deflation + the reference's iteration structure, shard proxy, the other
    solvers on the same matrix"""
    extras = {}
    if a.method == "NIPALS" and q == 1:
        proxy = None
        if p >= 8 * 1024:
            # (first: a millisecond-scale host measurement is best taken before the legs that leave tens of thousands
            # of Python objects and 32 GB of freed pool memory behind)
            note("extras: shard proxy")
            proxy = shard_proxy(ctx, mbpls_amd, dT, Y, r, n, p // 8, K, full_fit_s)
        note("extras: explicit deflation, reference iteration structure")
        extras = nipals_extras(ctx, mdl, X, Y, n, p, barrier_sync)
        if proxy is not None:
            extras["shard_proxy"] = proxy
        if single_block:
            note("extras: other solvers on the same matrix")
            extras["other_solvers_same_matrix"] = other_solvers(ctx, X, n, K, r)
    return extras


def extras_child(a):
    """bench.py --extras-child: the side legs in a process of their own (same matrix, generated again in HBM)"""
    import mbpls_amd
    from mbpls_amd import MBPLS, DeviceMatrix
    ctx = mbpls_amd.get_context()
    n, p, K, q = a.n, a.p, a.K, a.q
    Ts, Y = planted(n, q, 32, seed=0)
    dT = ctx.from_host(Ts)
    X = DeviceMatrix(ctx, n, p, [0, p])
    X.fill_latent(dT, 32, 0.5, 1234, 0)
    ctx.sync()

    def sync():
        ctx.sync()

    mdl = MBPLS(n_components=K, method=a.method, standardize=True, calc_all=True, copy=True)
    for _ in range(2):
        mdl.fit(X, Y)
    out = side_legs(ctx, mbpls_amd, a, mdl, X, Y, dT, 32, n, p, K, q, True, sync, a.full_fit_s)
    if a.e2e_cols != 0:
        note("end-to-end leg (host numpy in)")
        out["end_to_end_host_numpy"] = end_to_end(ctx, X, Y, n, p if a.e2e_cols < 0 else min(a.e2e_cols, p), K, a.method)
    out["side_legs_process"] = "child process of the timing run (same matrix generated again; a failure here cannot cost the line)"
    print(json.dumps(out), flush=True)
    X.close()
    return 0


def run_extras_child(a, full_fit_s):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--extras-child", "--n", str(a.n), "--p", str(a.p), "--K", str(a.K),
           "--q", str(a.q), "--method", a.method, "--e2e-cols", str(a.e2e_cols), "--full-fit-s", repr(float(full_fit_s))]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=None, text=True, timeout=1200)
        for line in reversed(r.stdout.splitlines()):
            if line.lstrip().startswith("{"):
                return json.loads(line)
        return {"side_legs_error": f"child exited with {r.returncode} and printed no JSON"}
    except Exception as e:                               # noqa: BLE001 -- reported in the line, never fatal
        return {"side_legs_error": f"{type(e).__name__}: {e}"}


def first_fit_child(a):
    """bench.py --first-fit-child: a fresh process.  Times import, context, data generation, the FIRST fit (fresh
    estimator, everything cold: code objects, pools, pinned buffers, helper threads) and the second."""
    t_start = time.perf_counter()
    import mbpls_amd
    from mbpls_amd import MBPLS, DeviceMatrix
    t_imp = time.perf_counter()
    ctx = mbpls_amd.get_context()
    t_ctx = time.perf_counter()
    Ts, Y = planted(a.n, a.q, 32, seed=0)
    dT = ctx.from_host(Ts)
    X = DeviceMatrix(ctx, a.n, a.p, [0, a.p])
    X.fill_latent(dT, 32, 0.5, 1234, 0)
    ctx.sync()
    t_gen = time.perf_counter()
    walls, phases, keep = [], [], []
    for _ in range(3):
        t0 = time.perf_counter()
        m = MBPLS(n_components=a.K, method=a.method, standardize=True, calc_all=True, copy=True).fit(X, Y)
        walls.append(time.perf_counter() - t0)
        keep.append(m)                       # (released after the timing, as in the parent's timed region)
        phases.append({k: round(1e3 * v, 2) for k, v in m.fit_timings_.items()})
    print(json.dumps({"first_fit_wall_s": round(walls[0], 5), "second_fit_wall_s": round(walls[1], 5),
                      "third_fit_wall_s": round(walls[2], 5), "import_s": round(t_imp - t_start, 3),
                      "context_s": round(t_ctx - t_imp, 3), "generate_s": round(t_gen - t_ctx, 3),
                      "host_phases_ms_first_fit": phases[0], "host_phases_ms_second_fit": phases[1],
                      "protocol": "fresh process (child started before the parent touched the GPU), X generated in HBM, "
                                  "fresh estimator per fit"}), flush=True)
    X.close()
    return 0


def run_first_fit_child(a):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--first-fit-child", "--n", str(a.n), "--p", str(a.p), "--K", str(a.K),
           "--q", str(a.q), "--method", a.method]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        for line in reversed(r.stdout.splitlines()):
            if line.lstrip().startswith("{"):
                return json.loads(line)
        return {"error": f"child exited with {r.returncode}: {r.stderr[-300:]}"}
    except Exception as e:                               # noqa: BLE001 -- reported in the line, never fatal
        return {"error": f"{type(e).__name__}: {e}"}


def measured_full_width(n, p, K, method):
    """The faithful oracle timed ONCE at the full width of this workload on a GPU box's host (tools/
    cpu_baseline_full_width.py: 164-199 GB of host RAM, four minutes -- not part of the default run); the committed record,
    labelled with its date and box, when it is for exactly this shape."""
    for name in ("r06_cpu_baseline_full_width.json",):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.loads(f.read().strip().splitlines()[-1])
        except (OSError, ValueError):
            continue
        if (d.get("n"), d.get("p"), d.get("K"), d.get("method")) == (n, p, K, method) and "measured_full_width_s" in d:
            return {"measured_full_width_s": d["measured_full_width_s"], "lean_mode_s": d.get("lean_fit_s"),
                    "blas_threads": d.get("blas_threads"), "peak_rss_GB": d.get("peak_rss_GB"),
                    "gpu_vs_oracle_worst_relerr_all_attributes_at_full_width":
                        d.get("gpu_vs_faithful_oracle_worst_relerr_all_attributes"),
                    "source": f"profiles/{name} (tools/cpu_baseline_full_width.py, {d.get('date')}, host {d.get('host')}, "
                              f"{d.get('cpus_visible')} CPUs visible) -- NOT measured in this run"}
    return None


def cpu_baseline(ctx, X, Y, n, p, cols, K, method, gpu_fit_s, gpu_check=True):
    """BASELINE.md section 3 protocol on the GPU box's host cores: the oracle in FAITHFUL mode (= the reference's numpy
    op sequence, n x p temporaries included) and in LEAN mode (the GPU's algorithm on the CPU) at three widths of the
    SAME device-generated matrix, after one warm-up fit; a line fitted in p gives the labelled extrapolation to the
    full width (the full size needs ~5x X of host RAM and ~4 minutes).  BLAS threads = the cgroup CPU quota.
    Also checks the GPU fit of the largest sample against the faithful oracle."""
    from mbpls_amd import MBPLS
    from oracle import mbpls_oracle as orc
    from mbpls_amd.runtime import cpu_quota
    sizes = sorted({max(2048, cols // 2), cols})
    big = 4 * cols if gpu_check else 0               # one larger sample (single run) when the host has the memory for it
    try:
        with open("/proc/meminfo") as f:
            avail = next(int(line.split()[1]) * 1024 for line in f if line.startswith("MemAvailable"))
    except Exception:
        avail = 0
    if big and big <= p and 6.5 * 8.0 * n * big < avail:     # faithful mode holds ~5 X-sized temporaries
        sizes.append(big)
    Xs = X.download(0, sizes[-1])
    cols_checked = cols
    thr = cpu_quota()      # CPUs this container may really use (cgroup quota), not the visible count
    try:
        import threadpoolctl
        limiter = threadpoolctl.threadpool_limits(limits=thr)
    except Exception:
        limiter = None
    runs = []
    ref = None
    try:
        orc.fit(np.ascontiguousarray(Xs[:, :sizes[0]]), Y, n_components=K, method=method, faithful=True)   # warm-up
        for c in sizes:
            Xc = np.ascontiguousarray(Xs[:, :c])
            reps = 3 if c <= cols else 1                 # median of >= 3 runs (the large sample: one)
            rec = {"p": c, "runs_per_mode": reps}
            for mode, faithful in (("faithful", True), ("lean", False)):
                tt = []
                for _ in range(reps if faithful else 1):
                    t0 = time.perf_counter()
                    mdl = orc.fit(Xc, Y, n_components=K, method=method, standardize=True, calc_all=True, faithful=faithful)
                    tt.append(time.perf_counter() - t0)
                rec[mode + "_s"] = round(float(np.median(tt)), 3)
                if faithful:
                    rec["faithful_s_each"] = [round(t, 3) for t in tt]
                    if c == cols_checked:
                        ref = mdl
                    if c > cols_checked and gpu_check:   # the large sample: the GPU fit of the same columns against it
                        g = MBPLS(n_components=K, method=method).fit(Xc, Y)
                        sg = np.sign(np.sum(mdl.Ts_ * g.Ts_, axis=0))
                        rec["gpu_vs_oracle_relerr"] = float(max(
                            np.max(np.abs(g.Ts_ * sg - mdl.Ts_)) / np.max(np.abs(mdl.Ts_)),
                            np.max(np.abs(g.beta_ - mdl.beta_)) / np.max(np.abs(mdl.beta_)),
                            np.max(np.abs(g.P_[0] * sg - mdl.P_[0])) / np.max(np.abs(mdl.P_[0]))))
                        del g
            runs.append(rec)
            del Xc
    finally:
        if limiter is not None:
            limiter.restore_original_limits()
    iters = ref.get("nipals_iterations_", [2] * K)
    units = sum(2 * i + 3 for i in iters) + 1 + 3
    t_cpu = runs[-1]["faithful_s"]
    cols_timed = runs[-1]["p"]
    gbps = units * 8.0 * n * cols_timed / t_cpu / 1e9
    fit = {}
    pp = np.array([r["p"] for r in runs], dtype=float)
    for mode in ("faithful", "lean"):
        tt = np.array([r[mode + "_s"] for r in runs])
        slope, icpt = np.polyfit(pp, tt, 1) if len(runs) > 1 else (tt[0] / pp[0], 0.0)
        full = float(slope * p + icpt)
        fit[mode] = {"s_per_1e6_columns": round(float(slope) * 1e6, 2), "extrapolated_full_size_fit_s": round(full, 1),
                     "extrapolated_fits_per_s": round(1.0 / full, 6), "gpu_speedup_fits_per_s": round(full / gpu_fit_s, 1)}
    err = None
    if gpu_check:
        mdl = MBPLS(n_components=K, method=method).fit(np.ascontiguousarray(Xs[:, :cols_checked]), Y)
        s = np.sign(np.sum(ref.Ts_ * mdl.Ts_, axis=0))
        err = float(max(np.max(np.abs(mdl.Ts_ * s - ref.Ts_)) / np.max(np.abs(ref.Ts_)),
                        np.max(np.abs(mdl.beta_ - ref.beta_)) / np.max(np.abs(ref.beta_)),
                        np.max(np.abs(mdl.P_[0] * s - ref.P_[0])) / np.max(np.abs(ref.P_[0]))))
    measured = measured_full_width(n, p, K, method)
    if measured is not None:
        ext = fit["faithful"]["extrapolated_full_size_fit_s"]
        measured["extrapolation_of_this_run_s"] = ext
        measured["extrapolation_over_measured"] = round(ext / measured["measured_full_width_s"], 4)
    return {"value": round(gbps, 3), "unit": "GB/s", "cores": thr, "kind": "port", "measured_full_width": measured,
            "sample": f"oracle faithful mode (reference numpy op sequence) on the first {cols_timed} columns of the same "
                      f"matrix, n={n} K={K}: {t_cpu:.1f} s wall = {units} reference-algorithm units of 8np bytes / wall; "
                      f"BLAS threads = CPU quota = {thr} ({os.cpu_count()} logical CPUs visible); after one warm-up fit; "
                      f"median of 3 runs at the two smaller widths, one run at the largest",
            "fit_wall_s": t_cpu, "runs": runs, "linear_fit_in_p": fit,
            "extrapolation": f"EXTRAPOLATED to p={p} from {len(runs)} sizes (linear in p); not measured at full size",
            "gpu_vs_oracle_relerr_on_sample": err, "gpu_vs_oracle_sample_columns": cols_checked if gpu_check else None}


if __name__ == "__main__":
    main()
