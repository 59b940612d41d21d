R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02n
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python3 $R/tools/outlier_probe.py default > $O/probe.txt 2>&1; echo rc=$?
cat $O/probe.txt | tail -16
find $O/trace -name "*.csv" -exec ls -la {} \;
