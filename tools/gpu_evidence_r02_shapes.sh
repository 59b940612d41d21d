# rocprofv3 kernel statistics of the few-samples and the long-column kernel families (profiles/r02_*_rocprofv3_kernel_stats.csv)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02s
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/few -- python3 $R/tools/time_shapes.py 8 NIPALS 1 64,128,256,500,1000 > $O/few.log 2>&1; echo few rc=$?
rocprofv3 --kernel-trace --stats --output-format csv -d $O/long -- python3 $R/tools/time_long.py 1000000 2000 > $O/long.log 2>&1; echo long rc=$?
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sparse -- python3 -W ignore $R/tools/time_sparse_shapes.py > $O/sparse.log 2>&1; echo sparse rc=$?
cd $R
grep "^n=" $O/few.log | cut -c1-200; grep "^n=" $O/long.log | cut -c1-200
find $O -name "*kernel_stats.csv"
