cd $GRAFT_REPO_ROOT
for rep in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/bench20.err | tee gpurun_out/bench20.json | python tools/bench_digest.py; done
tail -2 gpurun_out/bench20.err
