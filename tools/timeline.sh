#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for b in ${BATCHES:-256 128}; do
  rm -rf /tmp/tl_$b
  rocprofv3 --kernel-trace -d /tmp/tl_$b -o tl -- python $REPO/bench.py --batch $b --steps 60 --warmup 5 --no-cpu-baseline --no-roofline --no-full-update --no-precisions --no-carla --no-tick --no-e2e > /tmp/tl_$b.json 2> /tmp/tl_$b.err
  DB=$(find /tmp/tl_$b -name '*.db' | head -1)
  python $REPO/tools/rocpd_timeline.py "$DB" 30 2 > $REPO/gpurun_out/timeline_$b.txt 2>&1
  python -c "import json;d=json.loads([l for l in open('/tmp/tl_$b.json') if l.startswith('{')][-1]);print('batch $b under rocprof', d['ms_per_step'])"
done
tail -n 12 $REPO/gpurun_out/timeline_256.txt
