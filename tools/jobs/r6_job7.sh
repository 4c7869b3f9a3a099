#!/bin/bash
# round 6, job 7: the full GPU suite at the round's state + smoke + the driver's bench line
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -30 > $OUT/j7_tests.txt
cat $OUT/j7_tests.txt | cut -c1-300
python -c "
import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null > $OUT/j7_bench20.json
python tools/bench_digest.py < $OUT/j7_bench20.json | cut -c1-600
python -c "
import json; d=json.load(open('$OUT/j7_bench20.json')); print({k:v for k,v in d['config'].items() if isinstance(v,(int,float)) or k.endswith('dtype')})"
