# End-of-round check on the GPU box: full -m gpu suite, smoke(), the default bench line:
#   gpurun --timeout 3000 -- 'bash tools/gpu_final_check.sh'
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1700 python -m pytest tests -m gpu -x -q --durations=15 ) 2>&1 | tail -24
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 900 python3 bench.py > gpurun_out/bench_final.log 2> gpurun_out/bench_final.err ) 2>&1 | tail -3; echo bench rc=$?; tail -1 gpurun_out/bench_final.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['fit_wall_s'], d['roofline']['frac'], d['roofline']['traffic_source'][:40], d['shard_proxy'].get('fixed_ms_per_fit'), d['first_fit_wall_s'], d['cpu_baseline']['runs'][-1])"
