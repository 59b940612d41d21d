cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j; mkdir -p $O
( time timeout 900 python3 bench.py > $O/bench_1.log 2> $O/bench_1.err ) 2>&1 | grep real; echo "bench rc=$?"; grep "bench \|Exception\|rror" $O/bench_1.err | tail -14
tail -1 $O/bench_1.log > $O/r04_bench_c2_line.json
python3 -c "
import json; d=json.load(open('$O/r04_bench_c2_line.json'))
print(sorted(d.keys()))
print(d['value'], d['fit_wall_s'], d['first_fit_wall_s'], d['explicit_deflation']['deflate_sweep_read_write'], d['shard_proxy']['fixed_ms_per_fit'], d['end_to_end_host_numpy'])
print(d.get('side_legs_process'), d['optional_q1_fixed_point'])"
