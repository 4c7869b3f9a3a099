cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_update.py -q -m gpu -x -k validation_between 2>&1 | tail -3
timeout 600 python tools/jobs/bugcheck.py 2>&1 | tail -3
python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/bench20.err | tee gpurun_out/bench20.json | cut -c1-400
tail -2 gpurun_out/bench20.err
B="--no-cpu-baseline --no-full-update --no-precisions --no-roofline"
for s in 20 50 100 300; do python bench.py --steps $s --warmup 5 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps $s: %.4f ms' % d['ms_per_step'])"; done
