R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02q
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "subspace or multi_response or matrix or golden or medium or row_subset or tiny or edges" > $O/sel.log 2>&1; echo sel rc=$?; tail -5 $O/sel.log
timeout 300 python tools/bench_configs.py nipals_q10 2>/dev/null | tail -1 | cut -c1-1200
