#!/bin/bash
python -m pytest tests/test_gpu_update.py tests/test_gpu_dropstats.py tests/test_gpu_dp.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
for b in 32 64 128; do
  for i in 1 2; do
    for v in 1 0; do
      RIFT_DEC_SPLIT=$v python bench.py --batch $b --steps 300 --no-cpu-baseline --no-precisions --no-roofline --no-full-update --no-e2e --no-carla --no-tick 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('b=$b split=$v', round(d['ms_per_step'],4), round(d['all_outputs']['ms_per_step'],4) if d.get('all_outputs') else None)"
    done
  done
done
