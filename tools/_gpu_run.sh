R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03m
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "cross_val or row_subset" 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_interop.py -x -q 2>&1 | tail -3
timeout 900 python tools/time_cv.py > $O/time_cv.txt 2>&1; cat $O/time_cv.txt
