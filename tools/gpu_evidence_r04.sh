# Round-4 evidence (everything named profiles/r04_*): run on the GPU box from the repo root, e.g.
#   gpurun --timeout 3400 -- 'bash tools/gpu_evidence_r04.sh'
# then, back here:  python tools/collect_r04.py
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04ev
mkdir -p $O
cd $R
timeout 900 python3 bench.py > $O/bench_full.log 2> $O/bench_full.err; echo bench rc=$?; tail -1 $O/bench_full.log > $O/r04_bench_c2_line.json
MBPLS_BENCH_SHARED_GPU=1 timeout 600 python3 bench.py --gpus 2 --p 200000 --steps 2 --warmup 1 --cpu-cols 8192 > $O/bench_shared2.log 2> $O/bench_shared2.err; echo shared2 rc=$?; tail -1 $O/bench_shared2.log > $O/r04_bench_selflaunch_shared_gpu_world2_line.json
timeout 600 python3 bench.py --dist --p 250000 --steps 5 --warmup 3 --cpu-cols 0 --e2e-cols 0 > $O/bench_dist1.log 2>&1; echo dist1 rc=$?; tail -1 $O/bench_dist1.log > $O/r04_bench_dist_rccl_world1_line.json
( for s in "16384 100000" "1000000 2000" "20000 20000" "20000 1000" "65536 30000"; do timeout 400 python tools/time_long.py $s 2>&1 | grep -v "host phases"; done ) > $O/r04_long_columns.txt
timeout 900 python tools/time_cv.py > $O/r04_time_cv.txt 2>&1
for c in c3 c4 simpls nipals_q10 unipals_q10 nipals_mb; do timeout 600 python tools/bench_configs.py $c 2>/dev/null | tail -1 >> $O/r04_other_configs_lines.jsonl; done
( timeout 900 python tools/time_shapes.py 8 NIPALS 1 2>/dev/null | grep '^n='; timeout 600 python tools/time_shapes.py 8 UNIPALS 1 1200,2500,3000,5000,6000,7000,8192,12000,16384 2>/dev/null | grep '^n=' ) > $O/r04_shapes.txt
( timeout 900 python tools/paper_protocol.py a --data both 2>/dev/null; timeout 1500 python tools/paper_protocol.py b --data both 2>/dev/null ) | grep '^{' > $O/r04_paper_protocol.jsonl
( echo "== after the fix (this build): UNIPALS n=6000 p=166666 3 blocks K=5, fresh estimators"; timeout 300 python tools/stall_fit.py 6000 166666 3 5 UNIPALS 12 2>&1 | cut -c1-70; echo "== MBPLS_STREAM_STATS=0"; MBPLS_STREAM_STATS=0 timeout 300 python tools/stall_fit.py 6000 166666 3 5 UNIPALS 6 2>&1 | cut -c1-70 ) > $O/r04_stall_n6000_after.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 6 --cpu-cols 0 --e2e-cols 0 --no-extras > $O/bench_under_rocprof.log 2>&1; echo stats rc=$?
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc/FETCH_SIZE -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-cols 0 --e2e-cols 0 --no-extras > $O/pmc_fetch.log 2>&1; echo fetch rc=$?
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc/WRITE_SIZE -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-cols 0 --e2e-cols 0 --no-extras > $O/pmc_write.log 2>&1; echo write rc=$?
for s in "16384 100000" "1000000 2000"; do set -- $s; tag=n$1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/long_stats_$tag -- python3 $R/tools/long_one.py $1 $2 4 > $O/long_stats_$tag.log 2>&1; echo long stats $tag rc=$?
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/long_pmc_$tag/FETCH_SIZE -- python3 $R/tools/long_one.py $1 $2 > $O/long_pmc_fetch_$tag.log 2>&1; echo long fetch $tag rc=$?
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/long_pmc_$tag/WRITE_SIZE -- python3 $R/tools/long_one.py $1 $2 > $O/long_pmc_write_$tag.log 2>&1; echo long write $tag rc=$?
done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace125k -- python3 $R/tools/gap_fit.py 125000 rccl > $O/gap125k.log 2>&1; echo trace rc=$?
cd $R
python tools/trace_gaps.py $O/trace125k --list > $O/r04_gaps_p125k.txt 2>&1
python tools/pmc_to_json.py $O/pmc 4096 1000000 r04 > $O/pmc_c2_summary.txt 2>&1; cp profiles/r04_pmc_traffic_c2.json profiles/r04_pmc_*_bench_c2_counter_collection.csv $O/ 2>/dev/null
python tools/pmc_long_to_json.py $O/long_pmc_n16384 16384 100000 $O/r04_pmc_traffic_long_n16384_p100000.json > $O/pmc_long_n16384.txt 2>&1
python tools/pmc_long_to_json.py $O/long_pmc_n1000000 1000000 2000 $O/r04_pmc_traffic_long_n1000000_p2000.json > $O/pmc_long_n1000000.txt 2>&1
find $O -name "*kernel_stats.csv" | head; ls $O
