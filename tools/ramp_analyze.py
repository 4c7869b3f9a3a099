"""Clocks or dispatch?  Per 20-step region of tools/ramp_trace.py (regions are separated by the host's synchronize: segments of the kernel
trace split at device-idle gaps > 60 us that precede a collate_kernel... the region boundaries are taken from the step counts): the mean
duration of every large kernel, the sum of all kernel durations per step (busy), the wall span per step, and the per-queue idle share.
If the ramp is a clock ramp the large kernels' durations shrink from region to region; if it is dispatch / overlap they stay and span does not.
    python tools/ramp_analyze.py <results.db> <ramp_host.json>"""
import json, sqlite3, sys
from collections import defaultdict

con = sqlite3.connect(sys.argv[1])
host = json.load(open(sys.argv[2]))
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = con.execute(f"select name, start, end, {qcol} from kernels order by start").fetchall()
BIG = ("dec_w_kernel", "nat_l0w", "nat_l1w", "nat_l2w", "enc_fused", "pe_w_kernel", "pe_out", "pe_stats1p", "prep_kernel", "nat_rank", "fpn_tail", "collate", "pi_backward", "update_tail")
col = [i for i, r in enumerate(rows) if "collate_kernel" in r[0]]
# steps in program order: 5 warm-up, then the regions of the host file
bounds, at = [], 5
for r in host["regions"]:
    bounds.append((r["tag"], at, at + r["steps"], r["ms_per_step"]))
    at += r["steps"]
if at > len(col):
    print(f"trace holds {len(col)} steps, host ran {at}: analysing what is there")
print(f"{'region':16s} {'host ms/step':>12s} {'span/step':>10s} {'busy/step':>10s} {'busy/span':>9s} | " + " ".join(f"{k[:9]:>9s}" for k in BIG))
base = None
for tag, a, b, hms in bounds:
    if b > len(col):
        break
    i0 = col[a]
    i1 = col[b] if b < len(col) else len(rows)
    seg = rows[i0:i1]
    # the region ends with the last kernel before the host's synchronize: cut the tail at the last kernel that starts before the next region's collate
    span = max(e for _, s, e, _ in seg) - seg[0][1]
    busy = sum(e - s for _, s, e, _ in seg)
    dur = defaultdict(list)
    for n, s, e, q in seg:
        for k in BIG:
            if k in n:
                dur[k].append(e - s)
    n = b - a
    means = [sum(dur[k]) / max(len(dur[k]), 1) / 1e3 for k in BIG]
    if base is None:
        base = means
    print(f"{tag:16s} {hms:12.4f} {span / n / 1e3:10.1f} {busy / n / 1e3:10.1f} {busy / span:9.3f} | " + " ".join(f"{m:9.2f}" for m in means))
# per-step detail of the first region: duration of the dominant kernels step by step
print("\nstep-by-step (first 40 timed steps): span to next collate, dec_w / nat_l2w / enc_fused / pe_w durations (us)")
for st in range(5, min(45, len(col) - 1)):
    seg = rows[col[st]:col[st + 1]]
    d = {k: [e - s for n, s, e, _ in seg if k in n] for k in ("dec_w_kernel", "nat_l2w", "enc_fused", "pe_w_kernel", "nat_l0w")}
    print(f"  step {st - 5:3d}: to next collate {(rows[col[st + 1]][1] - rows[col[st]][1]) / 1e3:8.1f}  " + "  ".join(f"{k[:8]} {sum(v) / max(len(v), 1) / 1e3:7.2f}" for k, v in d.items()))
# clock samples
keys, smp = host.get("sample_keys", []), host.get("samples", [])
if smp:
    print(f"\n{len(smp)} sysfs samples ({keys}); around each region start (t - t0 in ms: values)")
    for r in host["regions"][:14]:
        near = [(t, v) for t, v in smp if r["t0"] - 0.004 <= t <= r["t1"] + 0.001]
        pick = near[:: max(1, len(near) // 12)]
        print(f"  {r['tag']:16s} " + "  ".join(f"{(t - r['t0']) * 1e3:+.1f}:{'/'.join(v)}" for t, v in pick))
    dt = [smp[i + 1][0] - smp[i][0] for i in range(len(smp) - 1)]
    dt.sort()
    print(f"  sample period median {dt[len(dt) // 2] * 1e3:.3f} ms, p90 {dt[int(len(dt) * 0.9)] * 1e3:.3f} ms")
