cd $GRAFT_REPO_ROOT
RIFT_BENCH_FORCE_PG=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-update --no-roofline 2>/dev/null | python tools/bench_digest.py
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python tools/bench_digest.py
