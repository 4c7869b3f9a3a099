set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3_t1.txt
timeout 600 python tests/diagnostics/operand_report.py > gpurun_out/operand_report.txt 2>gpurun_out/operand_report.err
for p in bf16 fp16; do timeout 300 python bench.py --precision $p --no-cpu-baseline --no-full-update --steps 100 > gpurun_out/r3_b1_$p.json 2>gpurun_out/r3_b1_$p.err; done
cat gpurun_out/r3_t1.txt gpurun_out/operand_report.txt
python - <<'PY'
import json
for p in ("bf16","fp16"):
    try:
        d=json.load(open(f"gpurun_out/r3_b1_{p}.json")); print(p, d["ms_per_step"], d["all_outputs"]["ms_per_step"], d["final_loss"], d["roofline"]["per_kernel_ms_per_step"])
    except Exception as e: print(p, "ERR", e)
PY
