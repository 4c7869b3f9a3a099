cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_host_rlft.py tests/test_gpu_update.py tests/test_gpu_dp.py -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
B="--no-cpu-baseline --no-full-update --no-precisions --no-roofline --steps 300"
for a in 1 0; do for b in 32 64 128 192; do
  RIFT_NAT_ASIDE=$a python bench.py --batch $b $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nat_aside $a batch $b: %.4f ms loss %s' % (d['ms_per_step'], d.get('final_loss')))"
done; done
RIFT_SIDE_GATE=0 python bench.py --batch 256 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ungated+aside batch 256: %.4f ms' % d['ms_per_step'])"
