cd $GRAFT_REPO_ROOT
for p in 1 0; do echo "== PREFETCH=$p"; RIFT_PREFETCH=$p timeout 600 python tools/host_time.py 256 2>&1 | grep "host issue\|empty queue"; done
