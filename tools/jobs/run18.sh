cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -12
for es in 1 0; do RIFT_ENC_S=$es timeout 300 python bench.py --no-cpu-baseline --no-full-update --no-precisions --steps 200 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('enc_s=$es', round(d['ms_per_step'],4), round(d['all_outputs']['ms_per_step'],4), d['final_loss'], d['roofline']['per_kernel_ms_per_step'])"; done
