cd $GRAFT_REPO_ROOT; timeout 600 python tools/jobs/dbg_compact.py 2>&1 | tail -12
