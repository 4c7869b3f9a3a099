cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for ts in 0 1; do export RIFT_TRUNK_STREAMS=$ts; echo "== trunk streams $ts"; for rep in 1 2 3; do timeout 600 python -m pytest tests/test_host_rlft.py -m gpu -q -x -k second_stream 2>&1 | tail -1; done
for b in 256 32; do timeout 300 python bench.py --batch $b --no-cpu-baseline --no-full-update --no-precisions --no-roofline --steps 200 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('batch $b', d['ms_per_step'], d['all_outputs']['ms_per_step'], d['final_loss'])"; done; done
