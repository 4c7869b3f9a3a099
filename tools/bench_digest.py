"""One-line digest of a bench.py JSON line (stdin)."""
import json, sys
d = json.loads(sys.stdin.read())
print("ms_per_step %.4f" % d["ms_per_step"], "| steady %s" % (d.get("steady_state", {}).get("ms_per_step"),), "| all_outputs %.4f" % d["all_outputs"]["ms_per_step"],
      "| precisions", {k: round(v["ms_per_step"], 4) for k, v in d.get("precisions", {}).items() if isinstance(v, dict)},
      "| full_update %s" % (d.get("full_update", {}).get("seconds"),), "| roofline", {k: d["roofline"].get(k) for k in ("kernel", "achieved", "frac", "avg_launch_us", "traffic")} if "roofline" in d else None,
      "| cpu %s" % (d.get("cpu_baseline", {}).get("value"),))
