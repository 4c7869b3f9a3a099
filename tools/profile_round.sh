#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root):
#   1. bench.py (the judged command, with cpu_baseline)                    -> gpurun_out/bench.json
#   2. rocprofv3 --kernel-trace --stats of the same bench command          -> gpurun_out/kt_summary.txt
#   3. PMC passes (each in its own run, kernel-trace only): FETCH_SIZE, WRITE_SIZE, MFMA/VALU activity
# Copy the summaries into profiles/ afterwards.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=${STEPS:-30}
python $REPO/bench.py --steps $STEPS --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.err
# (the kernel trace and the counter passes run with the two chains of the forward on ONE stream: per-kernel times and counters of serial launches)
export RIFT_TWO_STREAMS=0
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $REPO/bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-full-update > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
DB=$(find /tmp/kt -name '*.db' | head -1)
python $REPO/tools/rocpd_summary.py "$DB" > $OUT/kt_summary.txt 2>&1
for f in $(find /tmp/kt -name '*stats*.csv' | head -4); do cp $f $OUT/; done
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pmc_$tag
  timeout 600 rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_$tag -o pmc -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-full-update > $OUT/pmc_$tag.json 2> $OUT/pmc_$tag.err
  DB=$(find /tmp/pmc_$tag -name '*.db' | head -1)
  python $REPO/tools/rocpd_pmc.py "$DB" > $OUT/pmc_$tag.txt 2>&1
done
ls -la $OUT
