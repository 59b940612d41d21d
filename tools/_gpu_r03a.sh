R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03a
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_rccl.py -x -q 2>&1 | tail -3
python -m pytest tests/test_gpu_parity.py -x -q -k "resident or quickstart or csv" 2>&1 | tail -3
MBPLS_BENCH_SHARED_GPU=1 timeout 600 python3 bench.py --gpus 2 --p 200000 --steps 2 --warmup 1 > $O/bench_shared2.log 2> $O/bench_shared2.err; echo shared2 rc=$?; tail -c 1500 $O/bench_shared2.log; tail -5 $O/bench_shared2.err
timeout 600 python3 bench.py --dist --p 125000 --steps 10 --warmup 3 --cpu-cols 0 --e2e-cols 0 --no-extras > $O/bench_dist125k.log 2>&1; echo dist rc=$?; tail -c 2500 $O/bench_dist125k.log
python tools/iters_deep.py 1000000 20 30 38 > $O/iters_deep.txt 2>&1; cat $O/iters_deep.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace125k -- python3 $R/tools/gap_fit.py 125000 > $O/gap125k.log 2>&1; echo trace rc=$?
cd $R
python tools/trace_gaps.py $O/trace125k --list > $O/gaps125k.txt 2>&1; tail -30 $O/gaps125k.txt
