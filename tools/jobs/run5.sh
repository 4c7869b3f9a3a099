cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -s -k "rtr or known_answers or rollout_fills" 2>&1 | grep -E "rtr vs|passed|failed|Error|error|assert" | head -30
