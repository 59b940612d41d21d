cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
timeout 300 python3 bench.py --first-fit-child --p 1000000 > $O/cold_child.txt 2>&1; cat $O/cold_child.txt
timeout 300 python tools/host_timeline.py --p 1000000 --fits 3 --fresh --show first --min-us 50 > $O/cold_c2.txt 2>&1; cat $O/cold_c2.txt
timeout 300 python tools/host_timeline.py --p 125000 --K 10 --fits 6 --fresh --show last --min-us 20 > $O/warm_125k.txt 2>&1; head -40 $O/warm_125k.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/stall -- python3 $GRAFT_REPO_ROOT/tools/stall_fit.py 6000 166666 3 5 UNIPALS 12 > $GRAFT_REPO_ROOT/$O/stall_fit.log 2>&1; echo trace rc=$?
cd $GRAFT_REPO_ROOT
cat $O/stall_fit.log | tail -14
python tools/trace_gaps.py $O/stall --list --split-ms 60 --pick slowest > $O/stall_gaps.txt 2>&1; head -70 $O/stall_gaps.txt
ls $O/stall/*/ | head; 
