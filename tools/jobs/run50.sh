cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-full-update --no-roofline --steps 20"
for w in 5 5 40 40; do
  python bench.py $B --warmup $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('warmup $w: %.4f ms steady %.4f' % (d['ms_per_step'], d['steady_state']['ms_per_step']))"
done
