cd $GRAFT_REPO_ROOT
timeout 600 python tools/host_time.py 32 2>&1 | grep -v "^RCCL" | sed -n 22,70p
