#!/bin/bash
# GPU job: the new host-arena tests + the end-to-end update figure with its host timeline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_host_rlft.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/e2e_tests.log
python bench.py --steps 20 --no-cpu-baseline --no-precisions --no-roofline > gpurun_out/e2e_bench.json 2> gpurun_out/e2e_bench.err
tail -3 gpurun_out/e2e_tests.log; python -c "
import json; d=json.load(open('gpurun_out/e2e_bench.json')); print(json.dumps(d.get('full_update_e2e'), indent=1)); print(d['ms_per_step'], d.get('full_update'))"
tail -5 gpurun_out/e2e_bench.err
