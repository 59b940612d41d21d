R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02r
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/all.log 2>&1; echo all rc=$?; tail -5 $O/all.log
for c in nipals_q10 unipals_q10; do timeout 300 python tools/bench_configs.py $c 2>/dev/null | tail -1 | cut -c1-1000; done
