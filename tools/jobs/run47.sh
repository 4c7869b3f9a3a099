cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-full-update --no-roofline --warmup 5"
for s in 20 20 300; do
  python bench.py $B --steps $s 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps $s: %.4f ms' % d['ms_per_step'])"
done
for b in 32 128; do python bench.py $B --no-precisions --steps 300 --batch $b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $b: %.4f ms' % d['ms_per_step'])"; done
