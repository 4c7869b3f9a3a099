cd $GRAFT_REPO_ROOT
python bench.py --steps 30 --no-cpu-baseline --no-full-update --no-precisions 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print({k: v for k, v in d['roofline']['per_kernel_ms_per_step'].items()})"
B="--no-cpu-baseline --no-full-update --no-precisions --no-roofline --steps 300"
for b in 256 256 32; do python bench.py --batch $b $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $b: %.4f ms' % d['ms_per_step'])"; done
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -6
