#!/bin/bash
# round 6, job 6: full GPU suite; the 112-row fused encoder at the CARLA shapes (A/B); LayerNorm guard with one branch per call (A/B)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -40 > $OUT/j6_tests.txt
cat $OUT/j6_tests.txt | cut -c1-300
{ echo "# RIFT_ENC112=0 (enc_w_kernel)"; RIFT_ENC112=0 timeout 300 python tools/shape_step.py bench carla carla-ragged 2>/dev/null
  echo "# RIFT_ENC112=1 (enc_fused112_kernel)"; RIFT_ENC112=1 timeout 300 python tools/shape_step.py bench carla carla-ragged 2>/dev/null
  echo "# RIFT_ENC112=0"; RIFT_ENC112=0 timeout 300 python tools/shape_step.py carla carla-ragged 2>/dev/null
  echo "# RIFT_ENC112=1"; RIFT_ENC112=1 timeout 300 python tools/shape_step.py carla carla-ragged 2>/dev/null; } > $OUT/j6_carla.txt 2>&1
cat $OUT/j6_carla.txt | cut -c1-400
{ echo "# one-pass LayerNorm without (librift_hip_noguard.so) / with (librift_hip.so) the cancellation guard, one branch per call in levels 0 / 1"
  timeout 600 bash tools/ab.sh rift_amd/librift_hip_noguard.so rift_amd/librift_hip.so 3; } > $OUT/j6_ab_lnguard.txt 2>&1
cat $OUT/j6_ab_lnguard.txt
