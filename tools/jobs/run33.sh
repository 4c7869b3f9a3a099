cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-full-update --no-precisions --no-roofline --steps 300"
for g in 1 0; do for b in 64 128 192; do
  RIFT_SIDE_GATE=$g python bench.py --batch $b $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('side_gate $g batch $b: %.4f ms' % d['ms_per_step'])"
done; done
