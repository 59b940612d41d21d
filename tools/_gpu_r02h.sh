mkdir -p gpurun_out/r02h
python -m pytest tests -x -q -m gpu > gpurun_out/r02h/all.log 2>&1; echo all rc=$?; tail -4 gpurun_out/r02h/all.log
python tools/bench_configs.py c4 > gpurun_out/r02h/c4.log 2>&1; echo c4 rc=$?; tail -1 gpurun_out/r02h/c4.log | cut -c1-1500
python tools/bench_configs.py nipals_q10 > gpurun_out/r02h/nipals_q10.log 2>&1; tail -1 gpurun_out/r02h/nipals_q10.log | cut -c1-1500
python tools/bench_configs.py simpls > gpurun_out/r02h/simpls.log 2>&1; tail -1 gpurun_out/r02h/simpls.log | cut -c1-1200
