"""Which kernels is the step waiting for?  Runs the bench's headline leg once per kernel with RIFT_DELAY=<label>:<us> (one lane spinning on
the stream right behind every launch of that label: the kernel's successors start later, no CU is taken away) and prints
d(ms per step) / d(delay) -- ~1 for a kernel on the step's critical path, ~0 where the step pipeline has slack.

    python tools/critical_path.py [--delay 40] [--steps 300] [--batch 256] [labels ...]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LABELS = ["collate_kernel", "prep_kernel", "nat_rank_kernel", "nat_l0w_kernel", "nat_l1w_kernel", "nat_l2w_kernel", "fpn_tail_kernel",
          "pe_stats1_kernel", "pe_w_kernel", "bn_finalize_t_kernel", "pe_out_kernel", "fo_w_kernel", "ego_fused_kernel", "token_kernel",
          "enc_fused_kernel", "q0_fused_kernel", "dec_w_kernel", "pi_forward_kernel", "loss_kernel", "loss_reduce_kernel",
          "pi_backward_kernel", "loss_finalize_clip_kernel", "adamw_kernel"]


def step_ms(env, steps, batch):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--batch", str(batch), "--no-cpu-baseline", "--no-precisions",
           "--no-roofline", "--no-full-update", "--no-e2e", "--no-carla", "--no-tick"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600).stdout
    line = [ln for ln in out.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    return d["ms_per_step"], (d.get("all_outputs") or {}).get("ms_per_step")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--delay", type=float, default=40.0, help="microseconds behind every launch of the label")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("labels", nargs="*")
    a = ap.parse_args()
    env = dict(os.environ)
    env.pop("RIFT_DELAY", None)
    base = [step_ms(env, a.steps, a.batch) for _ in range(2)]
    b0 = min(b[0] for b in base)
    b1 = min(b[1] for b in base) if base[0][1] else None
    print(f"batch {a.batch}, {a.steps} steps, no delay: {b0:.4f} ms per step (every output: {b1})   [runs: {base}]", flush=True)
    print(f"{'label':28s} {'ms/step':>8s} {'d/delay':>8s}   {'all out':>8s} {'d/delay':>8s}")
    for lab in a.labels or LABELS:
        env["RIFT_DELAY"] = f"{lab}:{a.delay}"
        t, ta = step_ms(env, a.steps, a.batch)
        n = 2 if lab == "bn_finalize_t_kernel" else 1          # (launched twice per step)
        s = (t - b0) * 1e3 / (n * a.delay)
        sa = (ta - b1) * 1e3 / (n * a.delay) if (ta and b1) else float("nan")
        print(f"{lab:28s} {t:8.4f} {s:8.2f}   {ta if ta else float('nan'):8.4f} {sa:8.2f}", flush=True)


if __name__ == "__main__":
    main()
