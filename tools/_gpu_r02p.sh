R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02p
mkdir -p $O
cd $R
python -m pytest tests -x -q -m gpu > $O/all.log 2>&1; echo all rc=$?; tail -4 $O/all.log
python tools/bench_configs.py nipals_q10 2>/dev/null | tail -1 | cut -c1-900
