cd $GRAFT_REPO_ROOT
echo "== serial (no pipeline)"; RIFT_PIPELINE=0 timeout 600 python tools/jobs/stepdeltas.py 2>&1 | grep "deltas\|total"
echo "== tail deferred, no prefetch"; RIFT_PREFETCH=0 timeout 600 python tools/jobs/stepdeltas.py 2>&1 | grep "deltas\|total"
