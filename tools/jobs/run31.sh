cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_update.py tests/test_gpu_dp.py -q -m gpu -x 2>&1 | tail -5
B="--no-cpu-baseline --no-full-update --no-precisions --no-roofline --steps 300"
for cfg in "0 1" "1 1" "1 0"; do set -- $cfg
for b in 256 32; do
  RIFT_PREFETCH=$1 RIFT_SIDE_GATE=$2 python bench.py --batch $b $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prefetch $1 side_gate $2 batch $b: %.4f ms  loss %s' % (d['ms_per_step'], d.get('final_loss')))"
done; done
