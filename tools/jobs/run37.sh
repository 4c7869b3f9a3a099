cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-full-update --no-precisions --no-roofline --steps 300"
for m in 0 1; do for b in 32 128 256; do
  RIFT_MAP_ON_PREP=$m python bench.py --batch $b $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('map_on_prep $m batch $b: %.4f ms loss %s' % (d['ms_per_step'], d.get('final_loss')))"
done; done
for q in 4 6 16; do for b in 32 256; do
  GPU_MAX_HW_QUEUES=$q python bench.py --batch $b $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hw queues $q batch $b: %.4f ms' % d['ms_per_step'])"
done; done
