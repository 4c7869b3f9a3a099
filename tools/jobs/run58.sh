cd $GRAFT_REPO_ROOT
timeout 600 python tools/host_time.py 32 2>&1 | grep "host issue\|empty queue"
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
B="--no-cpu-baseline --no-full-update --no-precisions --no-roofline --steps 300"
for b in 32 64 128 256; do python bench.py --batch $b $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $b: %.4f ms' % d['ms_per_step'])"; done
