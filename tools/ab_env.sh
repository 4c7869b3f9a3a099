#!/bin/bash
# Same-box A/B of one environment switch:  tools/ab_env.sh <VAR> <valueA> <valueB> [reps] [bench args...]   (through gpurun, from the repo root)
VAR=$1; VA=$2; VB=$3; REPS=${4:-3}; shift 4
FAST="--steps 200 --warmup 5 --no-cpu-baseline --no-full-update --no-precisions --no-carla --no-tick --no-e2e --no-roofline"
for r in $(seq $REPS); do
  for V in $VA $VB; do
    env $VAR=$V timeout 300 python bench.py $FAST "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$V  %.4f ms/step' % d['ms_per_step'])"
  done
done
