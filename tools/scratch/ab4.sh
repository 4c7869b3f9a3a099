#!/bin/bash
# decoder FFN shared with the idle waves: parity + A/B (RIFT_DEC_DBG=32 turns the sharing off)
python -m pytest tests/test_gpu_parity.py tests/test_gpu_update.py tests/test_gpu_dropstats.py tests/test_gpu_shapes.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2 3; do
  for v in 0 32; do
    RIFT_DEC_DBG=$v python bench.py --steps 300 --no-cpu-baseline --no-precisions --no-full-update --no-e2e --no-carla 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dbg=$v', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_us'],1), d['roofline']['kernel'])"
  done
done
