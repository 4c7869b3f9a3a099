#!/bin/bash
run() { env "$@" python bench.py --steps 300 --no-cpu-baseline --no-precisions --no-roofline --no-full-update --no-carla 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],4), round(d['all_outputs']['ms_per_step'],4))"; }
for i in 1 2 3; do
  run RIFT_PASSA_NOFIT=1
  run RIFT_PASSA_NOFIT=0
done
