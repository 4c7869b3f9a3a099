cd $GRAFT_REPO_ROOT
for rep in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/bench20.err | tee gpurun_out/bench20.json | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'all', d['all_outputs']['ms_per_step'], 'prec', {k:(v['ms_per_step'] if isinstance(v,dict) else 0) for k,v in d['precisions'].items()}, 'full', d['full_update']['seconds'], 'roof', d['roofline']['achieved'], d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'])"; done
tail -2 gpurun_out/bench20.err
