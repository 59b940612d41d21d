mkdir -p gpurun_out/r02g
tools/micro/read_ceiling > gpurun_out/r02g/read_ceiling.txt 2>&1; cat gpurun_out/r02g/read_ceiling.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "multi_rhs or gram or kernel or KERNEL or matrix or golden" > gpurun_out/r02g/sel.log 2>&1; echo sel rc=$?; tail -4 gpurun_out/r02g/sel.log
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_sparse.py -x -q -m gpu > gpurun_out/r02g/full.log 2>&1; echo full rc=$?; tail -4 gpurun_out/r02g/full.log
python tools/bench_configs.py c4 > gpurun_out/r02g/c4.log 2>&1; echo c4 rc=$?; tail -1 gpurun_out/r02g/c4.log | cut -c1-1500
python tools/bench_configs.py unipals_q10 > gpurun_out/r02g/unipals_q10.log 2>&1; tail -1 gpurun_out/r02g/unipals_q10.log | cut -c1-1200
python tools/bench_configs.py c3 > gpurun_out/r02g/c3.log 2>&1; tail -1 gpurun_out/r02g/c3.log | cut -c1-1200
