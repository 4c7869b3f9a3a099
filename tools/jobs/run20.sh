cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_shapes.py -m gpu -q 2>&1 | tail -12
