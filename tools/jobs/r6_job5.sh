#!/bin/bash
# round 6, job 5: look-back ranking with relaxed atomics (tests + A/B at 256 / 32 scenes), LayerNorm guard A/B, fp16 margins, host time at 32 scenes
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -25 > $OUT/j5_tests.txt
cat $OUT/j5_tests.txt
{ timeout 600 bash tools/ab_env.sh RIFT_RANK_IN_PREP 0 1 3; echo "# 32 scenes"; timeout 600 bash tools/ab_env.sh RIFT_RANK_IN_PREP 0 1 3 --batch 32; } > $OUT/j5_ab_rank.txt 2>&1
cat $OUT/j5_ab_rank.txt
{ echo "# one-pass LayerNorm without (librift_hip_noguard.so) / with (librift_hip.so) the cancellation guard"
  timeout 600 bash tools/ab.sh rift_amd/librift_hip_noguard.so rift_amd/librift_hip.so 3; } > $OUT/j5_ab_lnguard.txt 2>&1
cat $OUT/j5_ab_lnguard.txt
{ echo "# packed-fp16 GELU (default)"; timeout 600 python tests/diagnostics/fp16_margin.py fp16 2>/dev/null | grep "worst over"; } > $OUT/j5_fp16_margin.txt
cat $OUT/j5_fp16_margin.txt
timeout 300 python tools/host_time.py 32 2>/dev/null > $OUT/j5_host_time_32.txt
grep -E "host issue|host cost" $OUT/j5_host_time_32.txt
sed -n 1,60p $OUT/j5_host_time_32.txt | cut -c1-160
