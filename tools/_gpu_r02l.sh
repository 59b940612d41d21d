R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02l
mkdir -p $O
python tools/outlier_probe.py default > $O/probe_default.txt 2>&1; cat $O/probe_default.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_rccl.py -x -q -m gpu -k "resident or rccl or shared or offset" > $O/sel.log 2>&1; echo sel rc=$?; tail -3 $O/sel.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-cols 0 --e2e-cols 0 --no-extras > $O/bench_under_rocprof.log 2>&1; echo stats rc=$?
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc/FETCH_SIZE -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-cols 0 --e2e-cols 0 --no-extras > $O/pmc_fetch.log 2>&1; echo fetch rc=$?
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc/WRITE_SIZE -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-cols 0 --e2e-cols 0 --no-extras > $O/pmc_write.log 2>&1; echo write rc=$?
cd $R
find $O -name "*.csv" | head -20
python bench.py > $O/bench_full.log 2>&1; echo bench rc=$?; tail -1 $O/bench_full.log | cut -c1-600
