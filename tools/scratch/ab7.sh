#!/bin/bash
for q in "" 8; do
for b in 32 64; do
  for own in 0 1; do
      GPU_MAX_HW_QUEUES=$q RIFT_DEC_OWN=$own python bench.py --batch $b --steps 300 --no-cpu-baseline --no-precisions --no-roofline --no-full-update --no-e2e --no-carla 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hwq=$q b=$b own=$own', round(d['ms_per_step'],4), round(d['all_outputs']['ms_per_step'],4) if d.get('all_outputs') else None)"
  done
done
done
