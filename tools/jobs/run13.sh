cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
RIFT_PROF_TOP=40 timeout 300 python bench.py --batch 32 --no-cpu-baseline --no-full-update --no-precisions --steps 100 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['gpu_ms_per_step_sum_of_kernels']); 
for k,v in d['roofline']['per_kernel_ms_per_step'].items(): print('  %-28s %.1f us' % (k, v*1e3))"
