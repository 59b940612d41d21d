cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_rccl.py -x -q ) > $O/tests.log 2>&1; echo tests rc=$?; tail -5 $O/tests.log
timeout 300 python tools/host_timeline.py --p 1000000 --fits 3 --fresh --show first --min-us 50 > $O/cold_c2.txt 2>&1; cat $O/cold_c2.txt
timeout 300 python tools/host_timeline.py --n 6000 --p 166666 --blocks 3 --K 5 --method UNIPALS --fits 15 --fresh --show slowest --min-us 100 > $O/stall_6000.txt 2>&1; cat $O/stall_6000.txt
timeout 300 python tools/host_timeline.py --n 7000 --p 142857 --blocks 3 --K 5 --method UNIPALS --fits 15 --fresh --show slowest --min-us 100 > $O/stall_7000.txt 2>&1; cat $O/stall_7000.txt
timeout 300 python3 bench.py --first-fit-child --p 1000000 > $O/cold_child.txt 2>&1; cat $O/cold_child.txt
