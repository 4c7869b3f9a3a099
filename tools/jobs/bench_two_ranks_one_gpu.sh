#!/bin/bash
# bench.py's N = 2 code path with both ranks on GPU 0 (gloo): gpurun --timeout 1800 -- 'bash tools/jobs/bench_two_ranks_one_gpu.sh'
cd ${GRAFT_REPO_ROOT:-.}
RIFT_BENCH_SAME_GPU=1 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 20 --warmup 5 > /tmp/b2.json 2> /tmp/b2.err
echo "rc $?"; grep "^\[rank0\]\|Error" /tmp/b2.err | head -12
python - <<'PY'
import json
try:
    d = json.loads(open('/tmp/b2.json').read())
    print({k: d[k] for k in ("n_gpus", "ms_per_step", "value", "scaling", "rccl_ranks")}, {k: round(d[k]["ms_per_step"], 3) for k in ("weak", "strong") if k in d})
except Exception as e:
    print("no json:", e, open('/tmp/b2.json').read()[:300])
PY
