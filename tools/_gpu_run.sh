R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03j
mkdir -p $O
cd $R
timeout 900 python tools/paper_protocol.py a --cpu > $O/paper_a.jsonl 2> $O/paper_a.err; echo a rc=$?; cut -c1-900 $O/paper_a.jsonl; tail -3 $O/paper_a.err
timeout 1200 python tools/paper_protocol.py b --cpu > $O/paper_b.jsonl 2> $O/paper_b.err; echo b rc=$?; cut -c1-900 $O/paper_b.jsonl; tail -3 $O/paper_b.err
for v in 0 1; do for c in simpls nipals_q10 unipals_q10; do echo -n "IMPLICIT=$v "; MBPLS_KERNEL_IMPLICIT=$v timeout 300 python tools/bench_configs.py $c 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config'], d['fit_s'], {k:(v['launches'],v['avg_ms']) for k,v in d['kernels'].items()})"; done; done
