cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-full-update --no-precisions --no-roofline --warmup 5"
for ra in 1 0; do for s in 20 20 20 300; do
  RIFT_RUN_AHEAD=$ra python bench.py $B --steps $s 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('run_ahead_one $ra steps $s: %.4f ms' % d['ms_per_step'])"
done; done
for b in 32 128; do python bench.py $B --steps 300 --batch $b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $b: %.4f ms' % d['ms_per_step'])"; done
