R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03l
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_few_samples.py tests/test_gpu_long.py tests/test_gpu_rccl.py -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace125k -- python3 $R/tools/gap_fit.py 125000 rccl > $O/gap125k.log 2>&1; echo trace rc=$?
cd $R
python tools/trace_gaps.py $O/trace125k --list > $O/gaps125k.txt 2>&1; tail -14 $O/gaps125k.txt
timeout 600 python3 bench.py --cpu-cols 0 --e2e-cols 0 > $O/bench_c2.log 2>&1; echo bench rc=$?; tail -1 $O/bench_c2.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['fit_wall_s'], d['fit_wall_s_same_estimator_refit'], d['value'], d['roofline']['frac'], {k:v['avg_ms'] for k,v in d['kernels'].items()}); print(d['shard_proxy']); print(d['host_phases_ms_last_fit'])"
