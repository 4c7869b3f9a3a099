#!/bin/bash
# round 6, job 3: the ranking inside the preparation's launch (tests + A/B), the fp16 build on the packed-fp16 GELU (precision + speed A/B)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -25 > $OUT/j3_tests.txt
cat $OUT/j3_tests.txt
bash tools/ab_env.sh RIFT_RANK_IN_PREP 0 1 3 > $OUT/j3_ab_rank.txt 2>&1
cat $OUT/j3_ab_rank.txt
{ echo "# fp16 build: rational fp32 GELU (librift_hip_f16gelu32.so) against the packed-fp16 GELU (librift_hip.so), --precision fp16"
  bash tools/ab.sh rift_amd/librift_hip_f16gelu32.so rift_amd/librift_hip.so 2 --precision fp16; } > $OUT/j3_ab_f16gelu.txt 2>&1
cat $OUT/j3_ab_f16gelu.txt
{ echo "# packed-fp16 GELU (default)"; python tests/diagnostics/fp16_margin.py fp16 2>/dev/null | grep "worst over"
  echo "# rational fp32 GELU"; RIFT_LIB=$REPO/rift_amd/librift_hip_f16gelu32.so python tests/diagnostics/fp16_margin.py fp16 2>/dev/null | grep "worst over"; } > $OUT/j3_fp16_margin.txt
cat $OUT/j3_fp16_margin.txt
python bench.py --steps 20 --warmup 5 --no-carla --no-tick --no-e2e --no-full-update 2>/dev/null > $OUT/j3_bench20.json
python tools/bench_digest.py < $OUT/j3_bench20.json
