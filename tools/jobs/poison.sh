#!/bin/bash
# the parity / shape / property tests with the scratch arena and every CU's LDS refilled with the NaN pattern before each forward / launch:
# gpurun --timeout 3000 -- 'bash tools/jobs/poison.sh'
cd ${GRAFT_REPO_ROOT:-.}
RIFT_POISON_ARENA=0xFF timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shapes.py tests/test_gpu_properties.py -q -m gpu 2>&1 | tail -4
RIFT_POISON_ARENA=0xFF RIFT_POISON_LDS=0xFF timeout 2400 python -m pytest tests/test_gpu_shapes.py -q -m gpu 2>&1 | tail -3
# the step pipeline likewise (the refill runs on the prepare stream, in front of the preparation): stream-placement equalities, interleaved validation,
# changing batch shapes, the data-parallel late tail, real ranks sharing the GPU
RIFT_POISON_ARENA=0xFF timeout 2400 python -m pytest tests/test_gpu_update.py tests/test_host_rlft.py tests/test_gpu_dp.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
