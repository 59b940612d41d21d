cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; mkdir -p $O
timeout 300 python tools/stall_fit.py 6000 166666 3 5 UNIPALS 12 > $O/stall_a.log 2>&1; cut -c1-16 $O/stall_a.log | tr '\n' ';'; echo
timeout 300 python tools/stall_fit.py 7000 142857 3 5 UNIPALS 12 > $O/stall_b.log 2>&1; cut -c1-16 $O/stall_b.log | tr '\n' ';'; echo
timeout 300 python tools/stall_fit.py 5500 180000 2 5 NIPALS 12 > $O/stall_c.log 2>&1; cut -c1-16 $O/stall_c.log | tr '\n' ';'; echo
( timeout 900 python tools/time_shapes.py 8 UNIPALS 1 1200,2500,3000,5000,6000,7000,8192 2>/dev/null | grep '^n=' ) > $O/shapes_unipals.txt; cut -c1-60 $O/shapes_unipals.txt
( timeout 900 python tools/time_shapes.py 8 NIPALS 1 40,64,128,256,500,1000,2048,3000,4096,5000,6000,8192 2>/dev/null | grep '^n=' ) > $O/shapes_nipals.txt; cat $O/shapes_nipals.txt
timeout 300 python3 bench.py --no-extras --no-first-fit --cpu-cols 0 --e2e-cols 0 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','fit_wall_s','fit_wall_s_max')}, d['fit_wall_s_each'], d['host_phases_ms_last_fit'])"
