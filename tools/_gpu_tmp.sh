cd $GRAFT_REPO_ROOT
python tools/time_small.py 2>&1 | tail -8
