cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/suite
( time timeout 1700 python -m pytest tests -m gpu -q --durations=40 ) > gpurun_out/suite/suite.log 2>&1; echo suite rc=$?; tail -60 gpurun_out/suite/suite.log
