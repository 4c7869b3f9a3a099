#!/bin/bash
# round 6, job 11: the full GPU suite at the round's last state, rollout tick with the one-copy staging, batch sweep, the driver's command
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -12 > $OUT/j11_tests.txt
cat $OUT/j11_tests.txt | cut -c1-250
{ echo "# rollout tick (python tools/tick_latency.py): host wall time of get_action for K CBVs of one environment, CARLA shapes, fp16 operands"
  python tools/tick_latency.py --ticks 60 2>/dev/null | grep -v "^{"; } > $OUT/j11_tick_latency.txt
cat $OUT/j11_tick_latency.txt
{ echo "# step time against the minibatch on one GPU (python bench.py --batch B --steps 200): what one of N ranks runs under strong scaling"
  for b in 32 64 128 256; do python bench.py --batch $b --steps 200 --no-cpu-baseline --no-full-update --no-precisions --no-roofline --no-carla --no-tick 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch %4d: %.4f ms/step, %.0f scenes/s' % ($b, d['ms_per_step'], d['value']))"; done; } > $OUT/j11_batch_sweep.txt
cat $OUT/j11_batch_sweep.txt
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null > $OUT/j11_bench20.json
python tools/bench_digest.py < $OUT/j11_bench20.json | cut -c1-500
