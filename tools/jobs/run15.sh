cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_host_rlft.py -m gpu -q -x -k second_stream 2>&1 | tail -30
