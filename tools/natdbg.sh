for d in 0; do
  RIFT_NAT_DBG=$d RIFT_PROF_TOP=40 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/natdbg_$d.json 2> gpurun_out/natdbg_$d.err
done
echo done
