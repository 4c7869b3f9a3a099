#!/bin/bash
# two (three) gloo ranks sharing GPU 0: gpurun --timeout 1800 -- 'bash tools/jobs/dp_two_ranks_one_gpu.sh'
cd ${GRAFT_REPO_ROOT:-.}
for n in ${RANKS:-2 3}; do
RIFT_DP_SAME_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$n --master-addr 127.0.0.1 --master-port 2963$n tests/dp_worker.py > /tmp/dpw_$n.log 2>&1
grep "^\[rank0\]" /tmp/dpw_$n.log | head -12
grep "DP_WORKER" /tmp/dpw_$n.log
done
