cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropstats.py tests/test_gpu_shapes.py -q -m gpu 2>&1 | tail -8
B="--no-cpu-baseline --no-full-update --no-precisions --no-roofline --steps 300"
for cmp in 0 1; do for b in 256 32; do
  RIFT_NAT_COMPACT=$cmp python bench.py --batch $b $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('compact $cmp batch $b: %.4f ms' % d['ms_per_step'])"
done; done
