"""Why a fresh trainer's first 20-step region is slower than the same region repeated: per-dispatch durations of the large kernels over the first
N dispatches of a rocprofv3 kernel trace (clock ramp shows as longer kernels, host / runtime warm-up as longer gaps).
    python tools/first_steps.py <results.db> [kernel substring]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
key = sys.argv[2] if len(sys.argv) > 2 else "dec_w_kernel"
rows = con.execute("select name, start, end from kernels order by start").fetchall()
k = [(s, e) for n, s, e in rows if key in n]
print(f"{len(k)} dispatches of {key}")
for i in range(0, min(len(k), 120), 5):
    chunk = k[i:i + 5]
    dur = sum(e - s for s, e in chunk) / len(chunk) / 1e3
    per = (chunk[-1][0] - chunk[0][0]) / max(len(chunk) - 1, 1) / 1e3
    print(f"  dispatches {i:3d}..{i + len(chunk) - 1:3d}: mean duration {dur:7.2f} us, start-to-start {per:8.2f} us")
