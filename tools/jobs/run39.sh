cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-full-update --no-precisions --no-roofline --steps 400"
for rep in 1 2; do for t in 0 1; do
  RIFT_PE_TAIL_SKIP=$t python bench.py --batch 256 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tail_skip $t: %.4f ms' % d['ms_per_step'])"
done; done
