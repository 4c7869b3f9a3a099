cd $GRAFT_REPO_ROOT; timeout 600 python tools/jobs/stepdeltas.py 2>&1 | tail -6
