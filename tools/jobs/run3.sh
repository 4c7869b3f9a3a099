cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_update.py -m gpu -q -s -k "forward_eval or bf16_trunk or fp16 or benchmark_batch or thirty" > gpurun_out/r3_t3_full.txt 2>&1
grep -E "forward_eval\[|trunk, |fp16 vs|256-scene|30-step|passed|failed|bf16 vs" gpurun_out/r3_t3_full.txt
