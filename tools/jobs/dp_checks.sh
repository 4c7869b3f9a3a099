#!/bin/bash
# data-parallel checks on a one-GPU box: gpurun --timeout 1800 -- 'bash tools/jobs/dp_checks.sh'
cd ${GRAFT_REPO_ROOT:-.}
timeout 1500 python -m pytest tests/test_gpu_dp.py tests/test_host_rlft.py tests/test_gpu_update.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29631 tests/dp_worker.py 2>&1 | grep "DP_WORKER\|Error\|error" | tail -3
RIFT_BENCH_FORCE_PG=1 python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-full-update --no-precisions --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('forced PG: %.4f ms rccl_ranks %s loss %s' % (d['ms_per_step'], d['rccl_ranks'], d['final_loss']))"
