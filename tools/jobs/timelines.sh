#!/bin/bash
# kernel timelines of two steady-state steps of the pipelined update (rocprofv3 kernel trace), 256 and 32 scenes:
# gpurun --timeout 1500 -- 'bash tools/jobs/timelines.sh'   ->  gpurun_out/timeline_256.txt, timeline_32.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for b in 256 32; do
  rm -rf /tmp/kt_$b && rocprofv3 --kernel-trace -d /tmp/kt_$b -o kt -- python $R/bench.py --batch $b --steps 40 --warmup 5 --no-cpu-baseline --no-full-update --no-precisions --no-roofline > /tmp/tl_$b.json 2> /tmp/tl_$b.err
  DB=$(find /tmp/kt_$b -name '*.db' | head -1)
  { echo "# python bench.py --batch $b --steps 40 under rocprofv3 --kernel-trace (the tracer slows the host: the device waits for it where it would not otherwise);"
    echo "# q0 = the caller's queue (token assembly, encoder, decoder), q1 = prepare stream (gather, preparation, ranking, history encoder), q2 = map chain, q3 = update stream (head .. AdamW)"
    python $R/tools/rocpd_timeline.py $DB 20 2; } > $R/gpurun_out/timeline_$b.txt 2>&1
done
