#!/bin/bash
for v in 0 8 1 9 16 32 40; do
  RIFT_DEC_DBG=$v RIFT_PROF_TOP=3 python bench.py --steps 60 --no-cpu-baseline --no-precisions --no-full-update --no-e2e --no-carla 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dbg=$v', round(d['ms_per_step'],4), d['roofline']['per_kernel_ms_per_step'])"
done
