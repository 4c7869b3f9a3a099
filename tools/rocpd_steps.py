"""Per-step GPU busy time and idle gaps from a rocprofv3 kernel trace (rocpd sqlite): steps are delimited by collate_kernel.
Usage: python tools/rocpd_steps.py <results.db>"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "collate_kernel" in r[0]]
steps = []
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a:b]
    span = rows[b][1] - seg[0][1]
    busy = sum(e - s for _, s, e in seg)
    gaps = [(seg[i + 1][1] - seg[i][2], seg[i][0], seg[i + 1][0]) for i in range(len(seg) - 1)] + [(rows[b][1] - seg[-1][2], seg[-1][0], rows[b][0])]
    steps.append((span, busy, len(seg), gaps))
steps.sort(key=lambda s: s[0])
print(f"{len(steps)} steps; span (us): min {steps[0][0] / 1e3:.1f} median {steps[len(steps) // 2][0] / 1e3:.1f} max {steps[-1][0] / 1e3:.1f}")
med = steps[len(steps) // 4]     # a fast (steady-state) step
print(f"a steady-state step: span {med[0] / 1e3:.1f} us, busy {med[1] / 1e3:.1f} us, {med[2]} kernels, idle {(med[0] - med[1]) / 1e3:.1f} us")
for g, a, b in sorted(med[3], key=lambda t: -t[0])[:12]:
    print(f"  gap {g / 1e3:7.2f} us  after {a[:60]:60s} before {b[:50]}")
print("kernels of that step, in order:")
a = starts[len(starts) // 2]
b = starts[len(starts) // 2 + 1]
for n, s_, e_ in rows[a:b]:
    print(f"  {(e_ - s_) / 1e3:8.2f} us  {n[:110]}")
