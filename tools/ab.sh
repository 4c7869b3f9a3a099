#!/bin/bash
# Same-box A/B of two builds of the library:  tools/ab.sh <libA.so> <libB.so> [reps] [bench args...]   (run through gpurun from the repo root)
# Alternates the two builds `reps` times: ms per 256-scene step (200 steps) of each run, then one roofline leg per build (per-kernel HIP-event times).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
A=$1; B=$2; REPS=${3:-3}; shift 3
FAST="--steps 200 --warmup 5 --no-cpu-baseline --no-full-update --no-precisions --no-carla --no-tick --no-e2e"
cd $REPO
for r in $(seq $REPS); do
  for L in $A $B; do
    RIFT_LIB=$REPO/$L python bench.py $FAST --no-roofline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s %.4f ms/step' % ('$L', d['ms_per_step']))"
  done
done
for L in $A $B; do
  RIFT_LIB=$REPO/$L python bench.py $FAST --no-pmc "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$L', 'ms/step %.4f' % d['ms_per_step'], 'sum of kernels %.1f us' % (1e3*r.get('gpu_ms_per_step_sum_of_kernels',0)))
for k,v in sorted(r.get('per_kernel_ms_per_step',{}).items(), key=lambda kv:-kv[1])[:14]: print('   %-28s %8.2f us' % (k,1e3*v))"
done
