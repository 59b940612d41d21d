cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "cross_val" --durations=5 ) > $O/tests.log 2>&1; echo tests rc=$?; tail -12 $O/tests.log
( timeout 600 python -m pytest tests/test_gpu_sparse.py -x -q --durations=5 ) > $O/tests2.log 2>&1; echo tests2 rc=$?; tail -8 $O/tests2.log
timeout 900 python tools/time_cv.py > $O/time_cv.txt 2>&1; grep -A2 "q=3" $O/time_cv.txt
