R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02t
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/all.log 2>&1; echo all rc=$?; tail -4 $O/all.log
rm -f $O/r02_other_configs_lines.jsonl
for c in c3 c4 simpls nipals_q10 unipals_q10 nipals_mb; do python tools/bench_configs.py $c 2>/dev/null | tail -1 >> $O/r02_other_configs_lines.jsonl; done
MBPLS_KERNEL_IMPLICIT=1 python tools/bench_configs.py c4 2>/dev/null | tail -1 | sed 's/"C4 KERNEL Gram"/"C4 KERNEL Gram, never-writes-X (MBPLS_KERNEL_IMPLICIT=1)"/' >> $O/r02_other_configs_lines.jsonl
cut -c1-330 $O/r02_other_configs_lines.jsonl
