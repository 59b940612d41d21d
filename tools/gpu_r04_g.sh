cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g; mkdir -p $O
timeout 300 python tools/long_one.py 1000000 2000 > $O/exit_1e6.log 2>&1; echo "long_one 1e6 plain rc=$?"; tail -2 $O/exit_1e6.log | cut -c1-200
timeout 300 python tools/long_one.py 65536 3000 > $O/exit_65536.log 2>&1; echo "long_one 65536 plain rc=$?"
( timeout 900 python tools/paper_protocol.py a --data both 2>/dev/null; timeout 1500 python tools/paper_protocol.py b --data both 2>/dev/null ) | grep '^{' > $O/r04_paper_protocol.jsonl; wc -l $O/r04_paper_protocol.jsonl
timeout 300 python3 bench.py --first-fit-child --p 1000000 > $O/cold_child.txt 2>&1; cut -c1-330 $O/cold_child.txt
for c in 4 2; do MBPLS_HIP_CHUNKS_PER_CU=$c timeout 300 python3 bench.py --no-first-fit --cpu-cols 0 --e2e-cols 0 --steps 5 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('chunks_per_cu=$c', d['fit_wall_s'], d['shard_proxy'])"; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace125k -- python3 $GRAFT_REPO_ROOT/tools/gap_fit.py 125000 rccl > $GRAFT_REPO_ROOT/$O/gap125k.log 2>&1; echo trace rc=$?
cd $GRAFT_REPO_ROOT
python tools/trace_gaps.py $O/trace125k --list > $O/r04_gaps_p125k.txt 2>&1; head -12 $O/r04_gaps_p125k.txt; tail -12 $O/r04_gaps_p125k.txt
