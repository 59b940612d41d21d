# round 4, first GPU call: cluster passes -- parity of the long-column tests, then timings
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
( time timeout 900 python -m pytest tests/test_gpu_long.py -x -q --durations=25 ) > gpurun_out/r04a/test_long.log 2>&1; echo long rc=$?; tail -40 gpurun_out/r04a/test_long.log
timeout 400 python tools/time_long.py 16384 100000 > gpurun_out/r04a/long_16384.txt 2>&1; cat gpurun_out/r04a/long_16384.txt
timeout 400 python tools/time_long.py 1000000 2000 > gpurun_out/r04a/long_1e6.txt 2>&1; cat gpurun_out/r04a/long_1e6.txt
timeout 400 python tools/time_long.py 20000 20000 > gpurun_out/r04a/long_20000.txt 2>&1; cat gpurun_out/r04a/long_20000.txt
