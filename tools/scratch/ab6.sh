#!/bin/bash
for b in 32 64 128 256; do
  for i in 1 2; do
    for v in 256 0; do
      RIFT_DEC_DEFER=$v python bench.py --batch $b --steps 300 --no-cpu-baseline --no-precisions --no-roofline --no-full-update --no-e2e --no-carla 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('b=$b defer<=$v', round(d['ms_per_step'],4), round(d['all_outputs']['ms_per_step'],4) if d.get('all_outputs') else None)"
    done
  done
done
