cd $GRAFT_REPO_ROOT; timeout 600 python tools/jobs/fill.py 2>&1 | tail -10
