#!/bin/bash
# round 6, job 4: look-back ranking inside the preparation's launch (tests + A/B at 256 and 32 scenes), fp16 build on the packed-fp16 GELU
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -25 > $OUT/j4_tests.txt
cat $OUT/j4_tests.txt
{ bash tools/ab_env.sh RIFT_RANK_IN_PREP 0 1 3; echo "# 32 scenes"; bash tools/ab_env.sh RIFT_RANK_IN_PREP 0 1 3 --batch 32; } > $OUT/j4_ab_rank.txt 2>&1
cat $OUT/j4_ab_rank.txt
{ echo "# fp16 build: rational fp32 GELU (librift_hip_f16gelu32.so) against the packed-fp16 GELU (librift_hip.so), --precision fp16"
  bash tools/ab.sh rift_amd/librift_hip_f16gelu32.so rift_amd/librift_hip.so 2 --precision fp16; } > $OUT/j4_ab_f16gelu.txt 2>&1
cat $OUT/j4_ab_f16gelu.txt
{ echo "# packed-fp16 GELU (default)"; python tests/diagnostics/fp16_margin.py fp16 2>/dev/null | grep "worst over"
  echo "# rational fp32 GELU"; RIFT_LIB=$REPO/rift_amd/librift_hip_f16gelu32.so python tests/diagnostics/fp16_margin.py fp16 2>/dev/null | grep "worst over"; } > $OUT/j4_fp16_margin.txt
cat $OUT/j4_fp16_margin.txt
python bench.py --steps 20 --warmup 5 --no-carla --no-tick --no-e2e --no-full-update 2>/dev/null > $OUT/j4_bench20.json
python tools/bench_digest.py < $OUT/j4_bench20.json
python -c "
import json; d=json.load(open('$OUT/j4_bench20.json')); print(json.dumps(d['config'])[-600:]); print(json.dumps(d.get('precision_contract',{}).get('fp16')), json.dumps(d.get('precision_contract',{}).get('bf16')))"
