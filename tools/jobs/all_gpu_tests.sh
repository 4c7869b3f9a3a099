#!/bin/bash
# the round-end checks in one call: gpurun --timeout 3600 -- 'bash tools/jobs/all_gpu_tests.sh'
cd ${GRAFT_REPO_ROOT:-.}
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -4
python -c "
import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python tools/bench_digest.py
