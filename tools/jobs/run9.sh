cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for e in 0 1; do if [ $e = 1 ]; then export RIFT_FORCE_ENC_W=1; fi; timeout 300 python bench.py --no-cpu-baseline --no-full-update --no-precisions --steps 100 > gpurun_out/r3_b9_$e.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r3_b9_$e.json')); print('force_encw=$e', d['ms_per_step'], d['final_loss'], d['roofline']['per_kernel_ms_per_step'])"; done
