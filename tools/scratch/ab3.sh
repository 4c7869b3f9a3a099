#!/bin/bash
# A/B of one environment switch over the small batches
VAR=${VAR:-RIFT_DEC_ASIDE}; ON=${ON:-64}; OFF=${OFF:-0}
for b in 32 64 128; do
  for i in 1 2; do
    for v in $ON $OFF; do
      env $VAR=$v python bench.py --batch $b --steps 300 --no-cpu-baseline --no-precisions --no-roofline --no-full-update --no-e2e --no-carla 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('b=$b $VAR=$v', round(d['ms_per_step'],4), round(d['all_outputs']['ms_per_step'],4) if d.get('all_outputs') else None)"
    done
  done
done
python -m pytest tests/test_gpu_update.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
