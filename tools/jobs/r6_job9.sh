#!/bin/bash
# round 6, job 9: level 2's size-dependent dealing (tests), batch sweep at the round's last state
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shapes.py tests/test_gpu_update.py -q -x 2>&1 | tail -4
{ echo "# step time against the minibatch on one GPU (python bench.py --batch B --steps 200): what one of N ranks runs under strong scaling"
  for b in 32 64 128 256; do python bench.py --batch $b --steps 200 --no-cpu-baseline --no-full-update --no-precisions --no-roofline --no-carla --no-tick 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch %4d: %.4f ms/step, %.0f scenes/s' % ($b, d['ms_per_step'], d['value']))"; done; } > $OUT/j9_batch_sweep.txt
cat $OUT/j9_batch_sweep.txt
