cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/shapes
timeout 900 python tools/time_shapes.py 8 NIPALS 1 2>/dev/null | grep "^n=" > gpurun_out/shapes/r03_shapes.txt
timeout 600 python tools/time_shapes.py 8 UNIPALS 1 1200,2500,3000,5000,6000,7000 2>/dev/null | grep "^n=" >> gpurun_out/shapes/r03_shapes.txt
cat gpurun_out/shapes/r03_shapes.txt | cut -c1-120
