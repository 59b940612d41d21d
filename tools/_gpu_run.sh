cd $GRAFT_REPO_ROOT
PYTHONPATH=$GRAFT_REPO_ROOT/_ab_old timeout 300 python tools/_ab_explicit.py 2>/dev/null | tail -5
PYTHONPATH=$GRAFT_REPO_ROOT timeout 300 python tools/_ab_explicit.py 2>/dev/null | tail -5
PYTHONPATH=$GRAFT_REPO_ROOT/_ab_old timeout 300 python tools/_ab_explicit.py 2>/dev/null | tail -5
