#!/bin/bash
# long runs of the step pipeline with and without a process group: gpurun --timeout 1800 -- 'bash tools/jobs/soak.sh'
cd ${GRAFT_REPO_ROOT:-.}
B="--warmup 5 --no-cpu-baseline --no-precisions --no-roofline"
python bench.py --steps 5000 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('5000 steps: %.4f ms loss %s full_update %s' % (d['ms_per_step'], d['final_loss'], d['full_update']['seconds']))"
RIFT_BENCH_FORCE_PG=1 python bench.py --steps 5000 $B --no-full-update 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('5000 steps, one-rank group: %.4f ms loss %s' % (d['ms_per_step'], d['final_loss']))"
python bench.py --batch 32 --steps 5000 $B --no-full-update 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('5000 steps of 32 scenes: %.4f ms loss %s' % (d['ms_per_step'], d['final_loss']))"
