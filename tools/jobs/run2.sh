cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tests/diagnostics/operand_report.py 2>gpurun_out/operand_report.err | grep "256 scenes" > gpurun_out/operand_report256.txt
for p in bf16 fp16; do timeout 300 python bench.py --precision $p --no-cpu-baseline --no-full-update --steps 100 > gpurun_out/r3_b2_$p.json 2>gpurun_out/r3_b2_$p.err; done
timeout 600 python -m pytest tests -m gpu -x -q -k "points or pe_ or forward_eval or poison or reprod" 2>&1 | tail -5
cat gpurun_out/operand_report256.txt; tail -3 gpurun_out/operand_report.err
python - <<'PY'
import json
for p in ("bf16","fp16"):
    try:
        d=json.load(open(f"gpurun_out/r3_b2_{p}.json")); print(p, d["ms_per_step"], d["all_outputs"]["ms_per_step"], d["final_loss"], d["roofline"]["per_kernel_ms_per_step"])
    except Exception as e: print(p, "ERR", e)
PY
