#!/bin/bash
# round 6, job 1: VALU issue rate against waves per SIMD; NAT levels 0 / 1 at 12 waves per workgroup (A/B); the idle-device ramp under a kernel trace
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
tools/ubench/valu_occupancy.bin > $OUT/valu_occupancy.txt 2>&1
cat $OUT/valu_occupancy.txt
RIFT_LIB=$REPO/rift_amd/librift_hip_w12.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "forward_eval or nat_level or compacted_history" 2>&1 | tail -5 > $OUT/w12_parity.txt
cat $OUT/w12_parity.txt
bash tools/ab.sh rift_amd/librift_hip.so rift_amd/librift_hip_w12.so 2 > $OUT/ab_w12.txt 2>&1
cat $OUT/ab_w12.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ramp && rocprofv3 --kernel-trace -d /tmp/ramp -o ramp -- python $REPO/tools/ramp_trace.py $OUT/ramp_host.json > $OUT/ramp_run.txt 2>&1
DB=$(find /tmp/ramp -name '*.db' | head -1)
python $REPO/tools/ramp_analyze.py "$DB" $OUT/ramp_host.json > $OUT/ramp_analysis.txt 2>&1
tail -60 $OUT/ramp_analysis.txt
# the same protocol without the profiler (host-side numbers only)
RAMP_NO_SAMPLER=1 python $REPO/tools/ramp_trace.py $OUT/ramp_host_noprof.json > $OUT/ramp_run_noprof.txt 2>&1
cat $OUT/ramp_run_noprof.txt | tail -22
