"""Per-kernel PMC totals from a rocprofv3 rocpd sqlite database (one --pmc pass).
Usage: python tools/rocpd_pmc.py <results.db>   Prints: kernel, dispatches, per-counter sum and per-dispatch mean."""
import sqlite3
import sys


def tables(cur):
    return [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    tb = tables(cur)
    pm = [t for t in tb if "pmc" in t.lower() or "counter" in t.lower()]
    print("# tables:", ", ".join(tb))
    view = next((t for t in tb if t == "counters_collection"), None)
    if view is None:
        for t in pm:
            cols = [c[1] for c in cur.execute(f"pragma table_info('{t}')")]
            print(f"# {t}: {cols}")
            for row in cur.execute(f"select * from '{t}' limit 3"):
                print("#   ", row)
        return
    cols = [c[1] for c in cur.execute(f"pragma table_info('{view}')")]
    print(f"# {view}: {cols}")
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    ncol = "counter_name" if "counter_name" in cols else "pmc_name"
    vcol = "value" if "value" in cols else "counter_value"
    did = "dispatch_id" if "dispatch_id" in cols else None
    q = f"select {kcol}, {ncol}, count(*), sum({vcol}){', count(distinct ' + did + ')' if did else ''} from {view} group by {kcol}, {ncol}"
    agg = {}
    for row in cur.execute(q):
        k, c, n, s = row[:4]
        nd = row[4] if did else n
        agg.setdefault(k, {})[c] = (nd, s)
    ctrs = sorted({c for v in agg.values() for c in v})
    print(f"{'dispatches':>10} " + " ".join(f"{c + '/disp':>28}" for c in ctrs) + "  kernel")
    tot = {c: 0.0 for c in ctrs}
    for k, v in sorted(agg.items(), key=lambda kv: -max(x[1] for x in kv[1].values())):
        nd = max(x[0] for x in v.values())
        print(f"{nd:10d} " + " ".join(f"{(v[c][1] / v[c][0] if c in v else 0):28.1f}" for c in ctrs) + "  " + (k if len(k) < 100 else k[:97] + "..."))
        for c in ctrs:
            tot[c] += v[c][1] if c in v else 0.0
    print("# totals over the run: " + ", ".join(f"{c}={tot[c]:.0f}" for c in ctrs))


if __name__ == "__main__":
    main(sys.argv[1])
