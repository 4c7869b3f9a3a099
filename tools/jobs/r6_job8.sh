#!/bin/bash
# round 6, job 8: what the driver's command measures (bench.py --gpus 1 --steps 20 --warmup 5, all legs) run to run, and against two switches
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
dig() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'ms_per_step %.4f' % d['ms_per_step'], 'steady %.4f' % d['config'].get('steady_state_ms_per_step', 0), 'all_outputs %.4f' % d['all_outputs']['ms_per_step'], 'fp16 %.4f' % d.get('precisions', {}).get('fp16', {}).get('ms_per_step', 0))"; }
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | dig default; done
for i in 1 2; do RIFT_PROBE_KEEP_CACHE=1 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | dig keep_cache; done
RIFT_RANK_IN_PREP=0 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | dig rank_own_launch
python bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc 2>/dev/null | dig no_pmc
{ echo "# level 2 dealt tile-major (librift_hip_l2old.so, rounds 2 - 5) against wave-major with idle waves skipping the arithmetic (librift_hip.so)"
  timeout 600 bash tools/ab.sh rift_amd/librift_hip_l2old.so rift_amd/librift_hip.so 3
  echo "# 32 scenes"; timeout 300 bash tools/ab.sh rift_amd/librift_hip_l2old.so rift_amd/librift_hip.so 2 --batch 32; } > $OUT/j8_ab_l2.txt 2>&1
cat $OUT/j8_ab_l2.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "nat or ranking or forward_eval or compacted_history" 2>&1 | tail -3
