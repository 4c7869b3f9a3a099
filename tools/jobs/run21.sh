cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dp.py -m gpu -q -k rccl 2>&1 | tail -8
RIFT_BENCH_FORCE_PG=1 timeout 300 python bench.py --no-cpu-baseline --no-full-update --no-precisions --no-roofline --steps 100 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('forced pg (1 rank, exchanges on):', round(d['ms_per_step'],4), d['rccl_ranks'], d['final_loss'])"
