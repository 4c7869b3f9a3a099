cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -12
for pl in 1 0; do RIFT_PIPELINE=$pl timeout 300 python bench.py --no-cpu-baseline --no-full-update --no-precisions --steps 200 > gpurun_out/r3_b7_pl$pl.json 2>gpurun_out/r3_b7_pl$pl.err; python -c "
import json; d=json.load(open('gpurun_out/r3_b7_pl$pl.json')); print('pipeline=$pl', d['ms_per_step'], d['all_outputs']['ms_per_step'], d['final_loss'])"; done
