cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shapes.py tests/test_gpu_dp.py -q -m gpu -x 2>&1 | tail -8
B="--no-cpu-baseline --no-full-update --no-precisions --no-roofline --steps 300"
for lv in 0 1; do for b in 256 32; do
  RIFT_PE_LIVE=$lv python bench.py --batch $b $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pe_live $lv batch $b: %.4f ms' % d['ms_per_step'])"
done; done
