cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_host_rlft.py tests/test_gpu_dp.py tests/test_gpu_update.py -m gpu -q -x 2>&1 | tail -6
for b in 256 32 64 128; do timeout 300 python bench.py --batch $b --no-cpu-baseline --no-full-update --no-precisions --no-roofline --steps 200 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('batch $b', d['ms_per_step'], d['all_outputs']['ms_per_step'], d['final_loss'])"; done
