R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03i
mkdir -p $O
cd $R
timeout 300 python tools/host_timeline.py 1000000 > $O/host_timeline_c2.txt 2>&1; grep -v "_res_\|rq.step\|rq.stream_out" $O/host_timeline_c2.txt | tail -22
timeout 600 python3 bench.py --cpu-cols 0 --e2e-cols 0 > $O/bench_c2.log 2>&1; echo bench rc=$?; tail -1 $O/bench_c2.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['fit_wall_s'], d['fit_wall_s_same_estimator_refit'], d['value'], d['roofline']['frac'], {k:v['avg_ms'] for k,v in d['kernels'].items()}); print(d['shard_proxy']); print(d['host_phases_ms_last_fit'])"
