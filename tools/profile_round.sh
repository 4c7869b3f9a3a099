#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root):
#   1. bench.py (the judged command, with cpu_baseline and the fp16 / fp32 legs)   -> gpurun_out/bench.json
#   2. rocprofv3 --kernel-trace --stats of the same bench command                  -> gpurun_out/kt_summary.txt
#   3. PMC passes (each in its own run, kernel-trace only): FETCH_SIZE, WRITE_SIZE, MFMA/VALU activity, LDS
#   4. step time against the minibatch (32 / 64 / 128 scenes), the dense-traffic step
# Copy the summaries into profiles/ afterwards (TAG=rNN tools/profile_round.sh does it).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
TAG=${TAG:-}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=${STEPS:-200}
python $REPO/bench.py --steps $STEPS --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python $REPO/bench.py --steps 20 --warmup 5 --no-pmc > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
tail -c 600 $OUT/bench.err
# (the kernel trace and the counter passes run with the forward's two chains on ONE stream and the step's tail behind its trunk: per-kernel
# times and counters of serial launches)
export RIFT_TWO_STREAMS=0 RIFT_PIPELINE=0
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-full-update --no-precisions --no-carla --no-tick --no-e2e --no-pmc > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
DB=$(find /tmp/kt -name '*.db' | head -1)
python $REPO/tools/rocpd_summary.py "$DB" > $OUT/kt_summary.txt 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pmc_$tag
  timeout 600 rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_$tag -o pmc -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-full-update --no-precisions --no-carla --no-tick --no-e2e > $OUT/pmc_$tag.json 2> $OUT/pmc_$tag.err
  DB=$(find /tmp/pmc_$tag -name '*.db' | head -1)
  python $REPO/tools/rocpd_pmc.py "$DB" > $OUT/pmc_$tag.txt 2>&1
done
unset RIFT_TWO_STREAMS RIFT_PIPELINE
python $REPO/tools/pmc_traffic.py $OUT $OUT/pmc_traffic.json
{ echo "# step time against the minibatch on one GPU (python bench.py --batch B --steps 200): what one of N ranks runs under strong scaling"
  for b in 32 64 128 256; do python $REPO/bench.py --batch $b --steps 200 --no-cpu-baseline --no-full-update --no-precisions --no-roofline --no-carla --no-tick 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch %4d: %.4f ms/step, %.0f scenes/s' % ($b, d['ms_per_step'], d['value']))"; done; } > $OUT/batch_sweep.txt
python $REPO/tools/dense_step.py > $OUT/dense_step.txt 2>/dev/null
python $REPO/tools/shape_step.py bench carla carla-ragged > $OUT/carla_step.txt 2>/dev/null
{ echo "# rollout tick (python tools/tick_latency.py): host wall time of get_action for K CBVs of one environment, CARLA shapes, fp16 operands"
  python $REPO/tools/tick_latency.py --ticks 60 2>/dev/null | grep -v "^{"; } > $OUT/tick_latency.txt
if [ -x $REPO/tools/ubench/group_gemm.bin ]; then
  { echo "# tools/ubench/group_gemm.bin: s_memtime ticks per group GEMM (32 fragments from LDS under 32 MFMAs) of wave 0 against the working waves of a workgroup"
    $REPO/tools/ubench/group_gemm.bin; } > $OUT/group_gemm.txt 2>/dev/null
fi
{ echo "# one rank over RCCL with the three exchanges forced (RIFT_BENCH_FORCE_PG=1 python bench.py --steps 200): the data-parallel step pipeline on one GPU"
  RIFT_BENCH_FORCE_PG=1 python $REPO/bench.py --steps 200 --no-cpu-baseline --no-full-update --no-precisions --no-roofline --no-carla --no-tick 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('forced exchanges: %.4f ms/step (rccl_ranks %d)' % (d['ms_per_step'], d['rccl_ranks']))"
  python $REPO/bench.py --steps 200 --no-cpu-baseline --no-full-update --no-precisions --no-roofline --no-carla --no-tick 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no group:         %.4f ms/step' % d['ms_per_step'])"; } > $OUT/forced_pg.txt
{ echo "# tools/fill_drain.py 256: what a K-step timed region costs beyond K steady-state steps"; python $REPO/tools/fill_drain.py 256 2>/dev/null | grep -v amdgpu.ids; } > $OUT/fill_drain.txt
BATCHES=256 bash $REPO/tools/timeline.sh > /dev/null 2>&1
if [ -n "$TAG" ]; then
  P=$REPO/gpurun_out/profiles_$TAG; mkdir -p $P
  cp $OUT/bench.json $P/${TAG}_bench.json; cp $OUT/bench_steps20.json $P/${TAG}_bench_steps20.json; cp $OUT/kt_summary.txt $P/${TAG}_rocprof_kernel_stats.txt; cp $OUT/pmc_traffic.json $P/${TAG}_pmc_traffic.json
  for t in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT; do cp $OUT/pmc_$t.txt $P/${TAG}_pmc_$t.txt; done
  cp $OUT/batch_sweep.txt $P/${TAG}_batch_sweep.txt; cp $OUT/dense_step.txt $P/${TAG}_dense_step.txt
  cp $OUT/carla_step.txt $P/${TAG}_carla_step.txt; cp $OUT/forced_pg.txt $P/${TAG}_forced_pg.txt
  cp $OUT/fill_drain.txt $P/${TAG}_fill_drain.txt; cp $OUT/timeline_256.txt $P/${TAG}_timeline_256.txt
  cp $OUT/tick_latency.txt $P/${TAG}_tick_latency.txt; [ -f $OUT/group_gemm.txt ] && cp $OUT/group_gemm.txt $P/${TAG}_group_gemm.txt
fi
ls -la $OUT | tail -30
