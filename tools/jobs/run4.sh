cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python bench.py --steps 100 > gpurun_out/r3_b4.json 2>gpurun_out/r3_b4.err; echo "rc=$?"
tail -3 gpurun_out/r3_b4.err
python bench.py --gpus 2 --steps 5; echo "rc(gpus 2 on this box)=$?"
for b in 32 64 128; do timeout 300 python bench.py --batch $b --steps 100 --no-cpu-baseline --no-full-update --no-precisions --no-roofline > gpurun_out/r3_batch_$b.json 2>/dev/null; done
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3_b4.json"))
print({k:v for k,v in d.items() if k not in ("config","roofline","cpu_baseline")})
print(d["roofline"]); print(d["cpu_baseline"])
for b in (32,64,128):
    x=json.load(open(f"gpurun_out/r3_batch_{b}.json")); print(b, x["ms_per_step"], x["value"])
PY
