cd $GRAFT_REPO_ROOT
timeout 300 python tools/_fp.py UNIPALS 2>/dev/null | grep "^fit\|finish\|wait" | cut -c1-200
