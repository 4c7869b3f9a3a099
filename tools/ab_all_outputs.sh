REPO=${GRAFT_REPO_ROOT:-$(pwd)}
FAST="--steps 200 --warmup 5 --no-cpu-baseline --no-full-update --no-precisions --no-carla --no-tick --no-e2e --no-roofline"
for r in 1 2; do for L in rift_amd/librift_hip_base.so rift_amd/librift_hip.so; do
  RIFT_LIB=$REPO/$L python bench.py $FAST 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-34s headline %.4f  all_outputs %.4f ms/step' % ('$L', d['ms_per_step'], d['all_outputs']['ms_per_step']))"
done; done
