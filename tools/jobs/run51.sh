cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -4
python -c "
import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --steps 3000 --warmup 5 --no-cpu-baseline --no-full-update --no-precisions --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('3000 steps: %.4f ms loss %s' % (d['ms_per_step'], d['final_loss']))"
RIFT_BENCH_FORCE_PG=1 python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-full-update --no-precisions --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('forced PG: %.4f ms rccl_ranks %s loss %s' % (d['ms_per_step'], d['rccl_ranks'], d['final_loss']))"
