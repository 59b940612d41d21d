# End-of-round check on the GPU box: full -m gpu suite, smoke(), the default bench line, and the self-launched two-rank
# dry run on one GPU:   gpurun --timeout 3000 -- 'bash tools/gpu_final_check.sh'
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1700 python -m pytest tests -m gpu -x -q --durations=15 ) 2>&1 | tail -24
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python3 bench.py > gpurun_out/bench_final.log 2>&1; echo bench rc=$?; tail -1 gpurun_out/bench_final.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['fit_wall_s'], d['roofline']['frac'], d['shard_proxy'].get('fixed_ms_per_fit'), d['explicit_deflation']['deflate_sweep_read_write']['avg_ms'], d['cpu_baseline']['runs'][-1])"
MBPLS_BENCH_SHARED_GPU=1 timeout 600 python3 bench.py --gpus 2 --p 200000 --steps 2 --warmup 1 > gpurun_out/bench_shared2.log 2> gpurun_out/bench_shared2.err; echo shared2 rc=$?; tail -1 gpurun_out/bench_shared2.log | cut -c1-300
