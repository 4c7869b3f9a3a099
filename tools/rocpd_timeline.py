"""Timeline of steady-state steps from a rocprofv3 kernel trace (rocpd sqlite) of the pipelined, two-stream step:
every kernel's start offset, duration and queue, the time with no kernel running and the time with only `small` kernels running.
Usage: python tools/rocpd_timeline.py <results.db> [first_step] [n_steps]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = con.execute(f"select name, start, end, {qcol or 0} from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "collate_kernel" in r[0]]
first = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2
a, b = starts[first], starts[first + n]
t0 = rows[a][1]
BIG = ("dec_w", "nat_l", "enc_fused", "pe_w_kernel", "pe_out")
seg = rows[a:b]
qs = sorted({r[3] for r in seg})
print(f"steps {first}..{first + n - 1}: span {(rows[b][1] - t0) / 1e3 / n:.1f} us per step; queues {qs}")
for nm, s, e, q in seg:
    short = nm.replace("rift_bf::", "").replace("void ", "").split("(")[0][:34]
    print(f"  {(s - t0) / 1e3:9.2f} .. {(e - t0) / 1e3:9.2f}  {(e - s) / 1e3:7.2f} us  q{qs.index(q)}  {short}")
# occupancy classes over the window
ev = []
for nm, s, e, q in seg:
    big = any(k in nm for k in BIG)
    ev.append((s, 1, big)); ev.append((e, -1, big))
ev.sort()
nb = ns = 0
last = ev[0][0]
acc = {"idle": 0, "small only": 0, "one big": 0, "two+ big": 0}
for t, d, big in ev:
    dt = t - last
    key = "idle" if nb + ns == 0 else ("small only" if nb == 0 else ("one big" if nb == 1 else "two+ big"))
    acc[key] += dt
    last = t
    if big: nb += d
    else: ns += d
tot = sum(acc.values())
print("time by what is running (us per step):", {k: round(v / 1e3 / n, 1) for k, v in acc.items()}, "total", round(tot / 1e3 / n, 1))
