#!/bin/bash
for i in 1 2 3; do for b in 32 64 128; do python bench.py --batch $b --steps 200 --no-cpu-baseline --no-full-update --no-precisions --no-roofline --no-carla --no-tick 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch %4d: %.4f ms/step' % ($b, d['ms_per_step']))"; done; done
