#!/bin/bash
for i in 1 2; do
  for v in none:none 0:0,0,0 -1:-1,-1,0 -1:0,-1,0 -1:0,0,0 -1:-1,0,0 0:-1,-1,0; do
    mp=${v%%:*}; sp=${v##*:}
    if [ $sp = none ]; then unset RIFT_STREAM_PRIO; else export RIFT_STREAM_PRIO=$sp; fi
    if [ $mp = none ]; then unset RIFT_BENCH_MAIN_PRIO; else export RIFT_BENCH_MAIN_PRIO=$mp; fi
    timeout 300 python bench.py --steps 300 --no-cpu-baseline --no-precisions --no-roofline --no-full-update --no-e2e --no-carla --no-tick 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('main=$mp prio=$sp', round(d['ms_per_step'],4), round(d['all_outputs']['ms_per_step'],4))"
  done
done
