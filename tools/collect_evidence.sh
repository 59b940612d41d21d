#!/bin/bash
# Evidence for profiles/: run on the GPU box (gpurun) from the repo root.
#   bash tools/collect_evidence.sh [bench|pmc|configs|all]
# Writes under gpurun_out/ev/; tools/pmc_to_json.py and a copy into profiles/ follow on the build machine.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ev
mkdir -p $OUT
what=${1:-all}
cd /tmp && export TMPDIR=/tmp
if [ $what = stats ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --cpu-cols 0 > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stats.err
fi
if [ $what = bench ] || [ $what = all ]; then
  python3 $ROOT/bench.py > $OUT/bench_line.json 2> $OUT/bench.err
  # same timed region; --cpu-cols 0 only drops the CPU-baseline sample, whose small GPU check fit (65 536 columns)
  # would otherwise dilute the per-kernel averages of the summary
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --cpu-cols 0 > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stats.err
fi
if [ $what = pmc ] || [ $what = all ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc/$c -- python3 $ROOT/bench.py --steps 1 --warmup 0 --cpu-cols 0 > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err
  done
fi
if [ $what = configs ] || [ $what = all ]; then
  : > $OUT/other_configs.jsonl
  for c in c3 simpls nipals_mb c4 nipals_q10 unipals_q10; do
    python3 $ROOT/tools/bench_configs.py $c 2>> $OUT/configs.err | tail -1 >> $OUT/other_configs.jsonl
  done
  python3 $ROOT/tools/time_sparse.py 250000 0.01 > $OUT/sparse.txt 2>&1
  python3 $ROOT/tools/time_long.py 16384 100000 > $OUT/long.txt 2>&1
  python3 $ROOT/tools/time_long.py 1000000 2000 >> $OUT/long.txt 2>&1
  python3 $ROOT/tools/gram_sweep.py 8192 > $OUT/gram_sweep.txt 2>&1
fi
ls -R $OUT | head -40
