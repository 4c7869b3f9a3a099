cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/bench20.err | tee gpurun_out/bench20.json | cut -c1-1500
tail -3 gpurun_out/bench20.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-update --no-precisions --no-roofline 2>/dev/null | cut -c1-300
