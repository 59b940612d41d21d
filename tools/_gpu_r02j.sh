mkdir -p gpurun_out/r02j
python -m pytest tests -x -q -m gpu > gpurun_out/r02j/all.log 2>&1; echo all rc=$?; tail -4 gpurun_out/r02j/all.log
python bench.py --steps 9 --warmup 2 --cpu-cols 0 --e2e-cols 0 --no-extras > gpurun_out/r02j/bench_quick.log 2>&1; tail -1 gpurun_out/r02j/bench_quick.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['fit_wall_s_each']); print(d['host_phases_ms_last_fit']); print(d['host_phases_ms_slowest_fit'])"
