cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03n
( time timeout 1700 python -m pytest tests -m gpu -x -q --durations=12 ) 2>&1 | tail -30
