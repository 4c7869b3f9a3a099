cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-full-update --no-precisions --no-roofline --steps 20 --warmup 5"
for cfg in "0 1" "1 1" "0 0"; do set -- $cfg
  for rep in 1 2 3; do
  RIFT_SIDE_GATE=$1 RIFT_NAT_ASIDE=$2 python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gate $1 aside $2: %.4f ms' % d['ms_per_step'])"
  done
done
for rep in 1 2 3; do RIFT_PREFETCH=0 python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no prefetch: %.4f ms' % d['ms_per_step'])"; done
