cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
rm -rf /tmp/kt && rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-full-update --no-precisions --no-roofline > $R/gpurun_out/tl_bench.json 2> /tmp/kt.err
DB=$(find /tmp/kt -name '*.db' | head -1)
python $R/tools/rocpd_timeline.py $DB 20 2 > $R/gpurun_out/timeline256b.txt 2>&1
