cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
rm -rf /tmp/kt2 && rocprofv3 --kernel-trace -d /tmp/kt2 -o kt -- python $R/bench.py --batch 32 --steps 40 --warmup 5 --no-cpu-baseline --no-full-update --no-precisions --no-roofline > $R/gpurun_out/tl_bench32.json 2> /tmp/kt2.err
DB=$(find /tmp/kt2 -name '*.db' | head -1)
python $R/tools/rocpd_timeline.py $DB 20 2 > $R/gpurun_out/timeline32b.txt 2>&1
