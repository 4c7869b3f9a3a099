"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average duration,
GPU busy time and inter-kernel gaps.  Usage: python tools/rocpd_summary.py <results.db> [steps]"""
import sqlite3
import sys


def main(path, steps=None):
    con = sqlite3.connect(path)
    cur = con.cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    if not rows:
        print("no kernel records")
        return
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(name, [0, 0])
        a[0] += 1
        a[1] += e - s
    tot = sum(v[1] for v in agg.values())
    span = rows[-1][2] - rows[0][1]
    gaps = sum(max(0, rows[i + 1][1] - rows[i][2]) for i in range(len(rows) - 1))
    print(f"kernels: {len(rows)} dispatches, busy {tot / 1e6:.3f} ms, span {span / 1e6:.3f} ms, gaps {gaps / 1e6:.3f} ms")
    small = [max(0, rows[i + 1][1] - rows[i][2]) for i in range(len(rows) - 1)]
    small = [g for g in small if g < 200_000]     # inter-kernel gaps inside steps (host pauses between phases excluded)
    if small:
        small.sort()
        print(f"inter-kernel gaps < 0.2 ms: n {len(small)}, sum {sum(small) / 1e6:.3f} ms, median {small[len(small) // 2] / 1e3:.2f} us, "
              f"p90 {small[int(len(small) * 0.9)] / 1e3:.2f} us")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>9} {'%':>6}  name")
    for name, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = name if len(name) < 110 else name[:107] + "..."
        print(f"{n:7d} {d / 1e6:10.3f} {d / n / 1e3:9.2f} {100 * d / tot:6.2f}  {short}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else None)
