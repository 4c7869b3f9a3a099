cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-full-update --no-precisions --no-roofline --steps 300"
for cfg in "0 0" "1 0" "1 -1" "1 -2"; do
  set -- $cfg
  for b in 256 32; do
    RIFT_NAT_MAIN=$1 RIFT_SIDE_PRIO=$2 python bench.py --batch $b $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nat_main $1 prio $2 batch $b: %.4f ms' % d['ms_per_step'])"
  done
done
python -c "
import torch
print(torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else '')"
