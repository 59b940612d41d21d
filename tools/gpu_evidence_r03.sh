# Round-3 evidence (everything named profiles/r03_* except the paper protocol, tools/paper_protocol.py, and the A/B notes):
# run on the GPU box from the repo root, e.g.
#   gpurun --timeout 3000 -- 'bash tools/gpu_evidence_r03.sh'
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03ev
mkdir -p $O
cd $R
timeout 900 python3 bench.py > $O/bench_full.log 2>&1; echo bench rc=$?; tail -1 $O/bench_full.log > $O/r03_bench_c2_line.json
MBPLS_BENCH_SHARED_GPU=1 timeout 600 python3 bench.py --gpus 2 --p 200000 --steps 2 --warmup 1 > $O/bench_shared2.log 2> $O/bench_shared2.err; echo shared2 rc=$?; tail -1 $O/bench_shared2.log > $O/r03_bench_selflaunch_shared_gpu_world2_line.json
timeout 600 python3 bench.py --dist --p 250000 --steps 5 --warmup 3 --cpu-cols 0 --e2e-cols 0 > $O/bench_dist1.log 2>&1; echo dist1 rc=$?; tail -1 $O/bench_dist1.log > $O/r03_bench_dist_rccl_world1_line.json
timeout 300 python tools/iters_deep.py 1000000 20 30 38 2>&1 | grep "^K=" > $O/r03_deep_iterations.txt
timeout 900 python tools/time_cv.py > $O/r03_time_cv.txt 2>&1
for c in c3 c4 simpls nipals_q10 unipals_q10 nipals_mb; do timeout 600 python tools/bench_configs.py $c 2>/dev/null | tail -1 >> $O/r03_other_configs_lines.jsonl; done
( timeout 900 python tools/time_shapes.py 8 NIPALS 1 2>/dev/null | grep '^n='; timeout 600 python tools/time_shapes.py 8 UNIPALS 1 1200,2500,3000,5000,6000,7000 2>/dev/null | grep '^n=' ) > $O/r03_shapes.txt
timeout 600 python tools/time_long.py 1000000 2000 > $O/r03_long_columns.txt 2>&1; timeout 600 python tools/time_long.py 16384 100000 >> $O/r03_long_columns.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 6 --cpu-cols 0 --e2e-cols 0 --no-extras > $O/bench_under_rocprof.log 2>&1; echo stats rc=$?
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc/FETCH_SIZE -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-cols 0 --e2e-cols 0 --no-extras > $O/pmc_fetch.log 2>&1; echo fetch rc=$?
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc/WRITE_SIZE -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-cols 0 --e2e-cols 0 --no-extras > $O/pmc_write.log 2>&1; echo write rc=$?
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace125k -- python3 $R/tools/gap_fit.py 125000 rccl > $O/gap125k.log 2>&1; echo trace rc=$?
cd $R
python tools/trace_gaps.py $O/trace125k --list > $O/r03_gaps_p125k.txt 2>&1
find $O -name "*kernel_stats.csv" | head; ls $O
