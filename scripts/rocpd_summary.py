"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / total / avg, like --stats."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'").fetchall()]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})").fetchall()]
scol = [r[1] for r in cur.execute(f"pragma table_info({ks})").fetchall()]
namecol = 'kernel_name' if 'kernel_name' in scol else ('display_name' if 'display_name' in scol else scol[-1])
q = f"select s.{namecol}, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.{namecol} order by 3 desc"
rows = cur.execute(q).fetchall()
tot = sum(r[2] for r in rows); n = sum(r[1] for r in rows)
t0, t1 = cur.execute(f"select min(start), max(end) from {kd}").fetchone()
print(f"kernels: {n} dispatches, total kernel time {tot/1e6:.1f} ms, trace span {(t1-t0)/1e6:.1f} ms")
print(f"{'name':90s} {'count':>8s} {'total_ms':>10s} {'avg_us':>9s} {'pct':>6s}")
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for name, c, s, mn, mx in rows[:top]:
    nm = re.sub(r'\s+', ' ', name)[:90]
    print(f"{nm:90s} {c:8d} {s/1e6:10.2f} {s/c/1e3:9.1f} {100*s/tot:6.1f}")
