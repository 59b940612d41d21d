cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j; mkdir -p $O
for i in 1 2; do timeout 900 python3 bench.py > $O/bench_$i.log 2> $O/bench_$i.err; echo "bench $i rc=$?"; grep "bench \|Exception\|rror" $O/bench_$i.err | tail -12; done
tail -1 $O/bench_2.log > $O/r04_bench_c2_line.json
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "cross_val" ) 2>&1 | tail -2
timeout 900 python tools/time_cv.py > $O/r04_time_cv.txt 2>&1; cat $O/r04_time_cv.txt
