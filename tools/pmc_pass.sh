#!/bin/bash
# One rocprofv3 PMC pass over a short bench run (kernel-trace only, as the pool requires):  tools/pmc_pass.sh TAG "COUNTER1 COUNTER2 ..."
# -> gpurun_out/pmc_TAG.txt (per-kernel counter sums per dispatch)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$TAG
timeout 600 rocprofv3 --kernel-trace --pmc $1 -d /tmp/pmc_$TAG -o pmc -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-full-update > $OUT/pmc_$TAG.json 2> $OUT/pmc_$TAG.err
DB=$(find /tmp/pmc_$TAG -name '*.db' | head -1)
python $REPO/tools/rocpd_pmc.py "$DB" > $OUT/pmc_$TAG.txt 2>&1
