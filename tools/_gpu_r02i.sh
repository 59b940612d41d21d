mkdir -p gpurun_out/r02i
python -m pytest tests -x -q -m gpu > gpurun_out/r02i/all.log 2>&1; echo all rc=$?; tail -4 gpurun_out/r02i/all.log
python tools/bench_configs.py c3 > gpurun_out/r02i/c3.log 2>&1; tail -1 gpurun_out/r02i/c3.log | cut -c1-1200
python tools/bench_configs.py c4 > gpurun_out/r02i/c4.log 2>&1; echo c4 rc=$?; tail -1 gpurun_out/r02i/c4.log | cut -c1-1500
python bench.py --steps 5 --warmup 2 --cpu-cols 0 --e2e-cols 0 --no-extras > gpurun_out/r02i/bench_quick.log 2>&1; tail -1 gpurun_out/r02i/bench_quick.log | cut -c1-1800
