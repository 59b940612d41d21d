R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02s
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/all.log 2>&1; echo all rc=$?; tail -4 $O/all.log
python bench.py --cpu-cols 0 --e2e-cols 0 --no-extras > $O/bench_quick.log 2>&1; grep -a '^{"metric"' $O/bench_quick.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['fit_wall_s'], d['fit_wall_s_each']); print(d['host_phases_ms_last_fit']); print(d['kernels'])"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 3 --warmup 6 --cpu-cols 0 --e2e-cols 0 --no-extras > $O/prof.log 2>&1
grep -h "k_reduce_chunks\|k_nipals_step\|k_scale_rows\|k_block_unit" $O/stats/*/*kernel_stats.csv | cut -c1-60,150-260
