# round 4, second batch: new parity cases, rank-equality worker, cold start, host stalls, fixed-cost trace
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; mkdir -p $O
( timeout 900 python -m pytest "tests/test_gpu_parity.py::test_medium_vs_oracle" tests/test_gpu_rccl.py "tests/test_gpu_parity.py::test_exhausted_response_iterations_follow_the_oracle" -x -q --durations=10 ) > $O/tests.log 2>&1; echo tests rc=$?; tail -25 $O/tests.log
timeout 600 python3 bench.py --cpu-cols 4096 --e2e-cols 0 > $O/bench.log 2> $O/bench.err; echo bench rc=$?; tail -1 $O/bench.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','fit_wall_s','fit_wall_s_max','first_fit_wall_s')}); print(d['cold_start']); print(d['shard_proxy']); print(d['roofline']['frac'], d['kernels'])"
( timeout 600 python tools/time_shapes.py 8 UNIPALS 1 1200,2500,3000,5000,6000,7000 2>/dev/null | grep '^n=' ) > $O/shapes_unipals.txt; cat $O/shapes_unipals.txt
timeout 300 python tools/host_timeline.py --p 125000 --K 10 > $O/host_timeline_125k.txt 2>&1; head -60 $O/host_timeline_125k.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace125k -- python3 $GRAFT_REPO_ROOT/tools/gap_fit.py 125000 rccl > $GRAFT_REPO_ROOT/$O/gap125k.log 2>&1; echo trace rc=$?
cd $GRAFT_REPO_ROOT
python tools/trace_gaps.py $O/trace125k --list > $O/gaps_p125k.txt 2>&1; head -80 $O/gaps_p125k.txt
