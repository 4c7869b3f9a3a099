cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
rm -rf /tmp/kt && rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-full-update --no-precisions --no-roofline > $R/gpurun_out/tl_bench.json 2> /tmp/kt.err
tail -c 300 /tmp/kt.err
DB=$(find /tmp/kt -name '*.db' | head -1)
python $R/tools/rocpd_timeline.py $DB 20 2 > $R/gpurun_out/timeline256.txt 2>&1
rm -rf /tmp/kt2 && rocprofv3 --kernel-trace -d /tmp/kt2 -o kt -- python $R/bench.py --batch 32 --steps 40 --warmup 5 --no-cpu-baseline --no-full-update --no-precisions --no-roofline > $R/gpurun_out/tl_bench32.json 2> /tmp/kt2.err
DB=$(find /tmp/kt2 -name '*.db' | head -1)
python $R/tools/rocpd_timeline.py $DB 20 2 > $R/gpurun_out/timeline32.txt 2>&1
cat $R/gpurun_out/tl_bench.json | cut -c1-200
