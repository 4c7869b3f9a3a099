import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
print([t for t in tabs if not t.startswith('rocpd_')][:40])
rows = cur.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
t0 = rows[0][1]
mx = rows[0][2]; gaps=[]
for i,(n,s,e,q,st) in enumerate(rows[1:],1):
    if s - mx > 30e6: gaps.append((i, mx, s))
    mx = max(mx,e)
print("n gaps", len(gaps), [round((g[1]-t0)/1e6) for g in gaps])
api = None
for t in ("regions", "hip_api", "api"):
    if t in tabs:
        api = t; break
print("api table", api)
if api:
    cols = [r[1] for r in cur.execute(f"pragma table_info({api})").fetchall()]
    print(cols)
for gi, g0, g1 in gaps[-4:]:
    print("---- gap", (g0-t0)/1e6, "->", (g1-t0)/1e6)
    for n,s,e,q,st in rows[gi-6:gi+4]:
        print(f"  {(s-t0)/1e6:10.3f} +{(e-s)/1e3:8.1f}us q={q} s={st} {n[:70]}")
    if api:
        for r in cur.execute(f"select name, start, end, tid from {api} where end > ? and start < ? order by start", (g0 - 200000, g1 + 200000)).fetchall()[:60]:
            if r[2]-r[1] > 200000 or True:
                print(f"     api {(r[1]-t0)/1e6:10.3f} +{(r[2]-r[1])/1e3:9.1f}us tid={r[3]} {r[0][:60]}")
