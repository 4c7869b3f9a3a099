cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for q in 4 8 16 32; do for b in 256 32; do GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --batch $b --no-cpu-baseline --no-full-update --no-precisions --no-roofline --steps 200 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('queues $q batch $b', round(d['ms_per_step'],4), round(d['all_outputs']['ms_per_step'],4))"; done; done
