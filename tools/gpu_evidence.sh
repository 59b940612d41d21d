#!/bin/bash
# Evidence of one round (everything named profiles/<round>_*): run on the GPU box from the repo root,
#   gpurun --timeout 3400 -- 'bash tools/gpu_evidence.sh r05 [part ...]'      parts: bench dist configs shapes cv paper sparse fuzz prof long gaps sparselong kroutes (default: all)
#   (fuzz part: FUZZ_SMALL_SEEDS / FUZZ_BIG_SEEDS / FUZZ_LONG_SEEDS / FUZZ_WIDE_SEEDS / FUZZ_SHARD_SEED choose the seeds of a campaign)
# then, back on the build machine:  python tools/collect_evidence.py r05 [--table]
R=${GRAFT_REPO_ROOT:-$(pwd)}
RND=${1:-r05}
shift
PARTS=${*:-bench dist configs shapes cv paper sparse fuzz prof long gaps sparselong kroutes}
O=$R/gpurun_out/${RND}ev
mkdir -p $O
cd $R
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
if has bench; then
  ( time timeout 900 python3 bench.py > $O/bench_full.log 2> $O/bench_full.err ) 2> $O/bench_full.time; echo bench rc=$?; tail -1 $O/bench_full.log > $O/${RND}_bench_c2_line.json; tail -3 $O/bench_full.time
fi
if has dist; then
  MBPLS_BENCH_SHARED_GPU=1 timeout 600 python3 bench.py --gpus 2 --p 200000 --steps 2 --warmup 1 --cpu-cols 8192 > $O/bench_shared2.log 2> $O/bench_shared2.err; echo shared2 rc=$?; tail -1 $O/bench_shared2.log > $O/${RND}_bench_selflaunch_shared_gpu_world2_line.json
  timeout 600 python3 bench.py --dist --p 250000 --steps 5 --warmup 3 --cpu-cols 0 --e2e-cols 0 > $O/bench_dist1.log 2>&1; echo dist1 rc=$?; tail -1 $O/bench_dist1.log > $O/${RND}_bench_dist_rccl_world1_line.json
  timeout 300 python3 tools/host_timeline.py --rccl --n 4096 --p 125000 --K 10 --fits 12 --fresh --show last > $O/${RND}_host_timeline_p125k_rccl.txt 2>&1
fi
if has configs; then
  rm -f $O/${RND}_other_configs_lines.jsonl
  for c in c3 c4 simpls nipals_q10 unipals_q10 nipals_mb; do timeout 600 python3 tools/bench_configs.py $c 2>/dev/null | grep "^{" >> $O/${RND}_other_configs_lines.jsonl; done
fi
if has shapes; then
  ( timeout 900 python3 tools/time_shapes.py 8 NIPALS 1 2>/dev/null | grep '^n='; timeout 600 python3 tools/time_shapes.py 8 UNIPALS 1 1200,2500,3000,5000,6000,7000,8192,12000,16384 2>/dev/null | grep '^n=' ) > $O/${RND}_shapes.txt
fi
if has cv; then timeout 900 python3 tools/time_cv.py > $O/${RND}_time_cv.txt 2>&1; fi
if has paper; then
  ( timeout 900 python3 tools/paper_protocol.py a --data both 2>/dev/null; timeout 1500 python3 tools/paper_protocol.py b --data both 2>/dev/null ) | grep '^{' > $O/${RND}_paper_protocol.jsonl
fi
if has sparse; then timeout 600 python3 tools/time_sparse.py 1000000 0.05 > $O/${RND}_sparse_n4096_p1m.txt 2>&1; fi
if has fuzz; then
  ( echo "tests/fuzz.py on the code of $RND (random shapes / block splits / methods / settings incl. column offsets and missing values; HIP path vs the oracles, worst relative error over all fitted attributes and predictions):"
    for sd in ${FUZZ_SMALL_SEEDS:-505 606}; do echo -n "small seed $sd: "; timeout 900 python3 tools/fuzz_parity.py 200 $sd 2>&1 | tail -1; done
    for sd in ${FUZZ_BIG_SEEDS:-707}; do echo -n "register-resident columns (1024 < n <= 8192) seed $sd: "; timeout 900 python3 tools/fuzz_parity.py 60 $sd big 2>&1 | tail -1; done
    for sd in ${FUZZ_LONG_SEEDS:-808}; do echo -n "long columns (8192 < n <= 131072) seed $sd: "; timeout 900 python3 tools/fuzz_parity.py 40 $sd long 2>&1 | tail -1; done
    for sd in ${FUZZ_WIDE_SEEDS:-909}; do echo -n "wide (n <= 300, p up to 200000) seed $sd: "; timeout 900 python3 tools/fuzz_parity.py 40 $sd wide 2>&1 | tail -1; done
    echo "column shards, two ranks sharing the GPU (tests/_shared_gpu_fuzz_worker.py), 200 problems:"
    PYTHONPATH=$R MBPLS_FUZZ_CASES=200 MBPLS_FUZZ_SEED=${FUZZ_SHARD_SEED:-1001} timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29611 tests/_shared_gpu_fuzz_worker.py 2>&1 | grep "random sharded problems" ) > $O/${RND}_fuzz_suite_seeds.txt
fi
if has long; then
  ( for s in "16384 100000" "1000000 2000" "20000 20000" "20000 1000" "65536 30000"; do timeout 400 python3 tools/time_long.py $s 2>&1 | grep -v "host phases"; done ) > $O/${RND}_long_columns.txt
fi
cd /tmp && export TMPDIR=/tmp
if has prof; then
  rm -rf $O/stats $O/pmc
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 6 --cpu-cols 0 --e2e-cols 0 --no-extras > $O/bench_under_rocprof.log 2>&1; echo stats rc=$?
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc/FETCH_SIZE -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-cols 0 --e2e-cols 0 --no-extras > $O/pmc_fetch.log 2>&1; echo fetch rc=$?
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc/WRITE_SIZE -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-cols 0 --e2e-cols 0 --no-extras > $O/pmc_write.log 2>&1; echo write rc=$?
  ( cd $R && python3 tools/pmc_to_json.py $O/pmc 4096 1000000 $RND > $O/pmc_c2_summary.txt 2>&1; cp profiles/${RND}_pmc_traffic_c2.json profiles/${RND}_pmc_*_bench_c2_counter_collection.csv $O/ 2>/dev/null )
fi
if has sparselong; then
  # missing values on long columns (two streaming phases): timings, kernel stats, HBM traffic from the PMC counters
  ( cd $R && timeout 900 python3 tools/time_sparse_long.py 16384 32768 65536 200000 2>/dev/null | grep '^n=' > $O/${RND}_sparse_long_columns.txt )
  rm -rf $O/sl_stats $O/sl_pmc
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sl_stats -- python3 $R/tools/sparse_long_one.py 32768 61035 3 > $O/sl_stats.log 2>&1; echo sl stats rc=$?
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/sl_pmc/FETCH_SIZE -- python3 $R/tools/sparse_long_one.py 32768 61035 1 > $O/sl_fetch.log 2>&1; echo sl fetch rc=$?
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/sl_pmc/WRITE_SIZE -- python3 $R/tools/sparse_long_one.py 32768 61035 1 > $O/sl_write.log 2>&1; echo sl write rc=$?
  ( cd $R && python3 tools/pmc_kernels_to_json.py $O/sl_pmc $O/${RND}_pmc_traffic_sparse_long.json $(python3 -c "print(8.0*32768*61035)") k_nl_ > $O/${RND}_pmc_traffic_sparse_long.txt 2>&1
    cp $(find $O/sl_stats -name '*kernel_stats.csv' | head -1) $O/${RND}_sparse_long_rocprofv3_kernel_stats.csv 2>/dev/null
    for c in FETCH_SIZE WRITE_SIZE; do cp $(find $O/sl_pmc/$c -name '*counter_collection.csv' | head -1) $O/${RND}_pmc_${c}_sparse_long_counter_collection.csv 2>/dev/null; done )
fi
if has kroutes; then
  ( cd $R && ( timeout 600 python3 tools/time_kernel_routes.py small; timeout 900 python3 tools/time_kernel_routes.py big; timeout 600 python3 tools/time_kernel_routes.py c4 ) 2>/dev/null | grep '^{' > $O/${RND}_kernel_routes.jsonl )
fi
if has gaps; then
  rm -rf $O/trace125k
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace125k -- python3 $R/tools/gap_fit.py 125000 rccl > $O/gap125k.log 2>&1; echo trace rc=$?
  ( cd $R && python3 tools/trace_gaps.py $O/trace125k --list > $O/${RND}_gaps_p125k.txt 2>&1 )
fi
ls $O | head -60
