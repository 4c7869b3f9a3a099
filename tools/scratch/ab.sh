#!/bin/bash
VAR=${VAR:-RIFT_PE_PACK}
for i in 1 2 3; do
  for v in 1 0; do
    env $VAR=$v python bench.py --steps 300 --no-cpu-baseline --no-precisions --no-roofline --no-full-update 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['ms_per_step'],4), round(d['all_outputs']['ms_per_step'],4))"
  done
done
