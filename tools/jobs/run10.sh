cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_dropstats.py tests/test_gpu_parity.py -m gpu -q -x -k "drop or decoder or seeded or bit" 2>&1 | tail -4
timeout 300 python bench.py --no-cpu-baseline --no-full-update --no-precisions --steps 200 > gpurun_out/r3_b10.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r3_b10.json')); print(d['ms_per_step'], d['all_outputs']['ms_per_step'], d['final_loss'], d['roofline']['per_kernel_ms_per_step'])"
