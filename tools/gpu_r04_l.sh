#!/bin/bash
# final refresh of the evidence that depends on the last code state: bench line, N > 1 dry-run lines, shard-sized timelines,
# other configurations, paper protocol, cross-validation timings
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04ev
mkdir -p $O
cd $R
timeout 900 python3 bench.py > $O/bench_full.log 2> $O/bench_full.err; echo bench rc=$?; tail -1 $O/bench_full.log > $O/r04_bench_c2_line.json
MBPLS_BENCH_SHARED_GPU=1 timeout 600 python3 bench.py --gpus 2 --p 200000 --steps 2 --warmup 1 --cpu-cols 8192 > $O/bench_shared2.log 2> $O/bench_shared2.err; echo shared2 rc=$?; tail -1 $O/bench_shared2.log > $O/r04_bench_selflaunch_shared_gpu_world2_line.json
timeout 600 python3 bench.py --dist --p 250000 --steps 5 --warmup 3 --cpu-cols 0 --e2e-cols 0 > $O/bench_dist1.log 2>&1; echo dist1 rc=$?; tail -1 $O/bench_dist1.log > $O/r04_bench_dist_rccl_world1_line.json
timeout 300 python3 tools/host_timeline.py --rccl --n 4096 --p 125000 --K 10 --fits 12 --fresh --show last > $O/r04_host_timeline_p125k_rccl.txt 2>&1
rm -f $O/r04_other_configs_lines.jsonl
for c in c3 c4 simpls nipals_q10 unipals_q10 nipals_mb; do timeout 600 python3 tools/bench_configs.py $c 2>/dev/null | tail -1 >> $O/r04_other_configs_lines.jsonl; done
( timeout 900 python3 tools/paper_protocol.py a --data both 2>/dev/null; timeout 1500 python3 tools/paper_protocol.py b --data both 2>/dev/null ) | grep '^{' > $O/r04_paper_protocol.jsonl
timeout 900 python3 tools/time_cv.py > $O/r04_time_cv.txt 2>&1
( timeout 900 python3 tools/time_shapes.py 8 NIPALS 1 2>/dev/null | grep '^n='; timeout 600 python3 tools/time_shapes.py 8 UNIPALS 1 1200,2500,3000,5000,6000,7000,8192,12000,16384 2>/dev/null | grep '^n=' ) > $O/r04_shapes.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $O/stats $O/trace125k
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 6 --cpu-cols 0 --e2e-cols 0 --no-extras > $O/bench_under_rocprof.log 2>&1; echo stats rc=$?
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace125k -- python3 $R/tools/gap_fit.py 125000 rccl > $O/gap125k.log 2>&1; echo trace rc=$?
cd $R
python3 tools/trace_gaps.py $O/trace125k --list > $O/r04_gaps_p125k.txt 2>&1
tail -1 $O/bench_full.log | cut -c1-200
