#!/bin/bash
# round 6: the idle-device ramp -- in-kernel shader-clock probes beside the steps, the same protocol without them, and the serial one-stream
# step under a kernel trace (clean per-kernel durations region by region)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
python tools/ramp_trace.py $OUT/ramp_probe.json 2>/dev/null > $OUT/ramp_probe.txt
cat $OUT/ramp_probe.txt
RAMP_NO_PROBE=1 python tools/ramp_trace.py $OUT/ramp_noprobe.json 2>/dev/null > $OUT/ramp_noprobe.txt
cat $OUT/ramp_noprobe.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ramp && RAMP_NO_PROBE=1 RIFT_TWO_STREAMS=0 RIFT_PIPELINE=0 rocprofv3 --kernel-trace -d /tmp/ramp -o ramp -- python $REPO/tools/ramp_trace.py $OUT/ramp_serial.json > $OUT/ramp_serial_run.txt 2>&1
DB=$(find /tmp/ramp -name '*.db' | head -1)
python $REPO/tools/ramp_analyze.py "$DB" $OUT/ramp_serial.json > $OUT/ramp_serial_analysis.txt 2>&1
head -18 $OUT/ramp_serial_analysis.txt | cut -c1-250
grep "ms/step" $OUT/ramp_serial_run.txt
