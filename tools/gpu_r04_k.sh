#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04ev
mkdir -p $O
cd $R
timeout 900 python3 bench.py > $O/bench_full.log 2> $O/bench_full.err; echo bench rc=$?; tail -1 $O/bench_full.log > $O/r04_bench_c2_line.json
timeout 300 python3 tools/host_timeline.py --rccl --n 4096 --p 125000 --K 10 --fits 12 --fresh --show last > $O/r04_host_timeline_p125k_rccl.txt 2>&1
MBPLS_COPIER_TRACE=1 timeout 300 python3 tools/host_timeline.py --rccl --n 4096 --p 125000 --K 10 --fits 6 --fresh --show last 2>&1 | grep "copier\] job" | tail -13 > $O/copier_trace.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $O/stats $O/trace125k
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 6 --cpu-cols 0 --e2e-cols 0 --no-extras > $O/bench_under_rocprof.log 2>&1; echo stats rc=$?
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace125k -- python3 $R/tools/gap_fit.py 125000 rccl > $O/gap125k.log 2>&1; echo trace rc=$?
cd $R
python3 tools/trace_gaps.py $O/trace125k --list > $O/r04_gaps_p125k.txt 2>&1
tail -1 $O/bench_full.log | cut -c1-200
