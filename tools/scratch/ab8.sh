#!/bin/bash
for v in 0 64 32 0 64 32; do DEC_TS_BRIEF=1 RIFT_DEC_DBG=$v python tools/decw_ts.py 2>&1 | tail -1; done
for i in 1 2; do
  for v in 0 64 32; do
    RIFT_DEC_DBG=$v python bench.py --steps 300 --no-cpu-baseline --no-precisions --no-full-update --no-e2e --no-carla --no-tick 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dbg=$v', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_us'],1), d['roofline']['kernel'])"
  done
done
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropstats.py tests/test_gpu_shapes.py -m gpu -x -q 2>&1 | tail -3
