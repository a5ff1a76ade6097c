"""Per-kernel stats over the LAST `--ms` milliseconds of a rocprofv3 kernel trace (steady-state graph replays only)."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor(); win_ms = float(sys.argv[2]); nsteps = float(sys.argv[3]) if len(sys.argv) > 3 else 1
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'").fetchall()]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]; ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
scol = [r[1] for r in cur.execute(f"pragma table_info({ks})").fetchall()]
namecol = 'kernel_name' if 'kernel_name' in scol else 'display_name'
t1 = cur.execute(f"select max(end) from {kd}").fetchone()[0]; t0 = t1 - int(win_ms * 1e6)
rows = cur.execute(f"select s.{namecol}, count(*), sum(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id where d.start >= {t0} group by s.{namecol} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows); n = sum(r[1] for r in rows)
print(f"window {win_ms} ms ({nsteps} steps): {n/nsteps:.0f} dispatches/step, kernel time {tot/1e6/nsteps:.1f} ms/step")
print(f"{'name':84s} {'count/step':>10s} {'ms/step':>9s} {'avg_us':>8s} {'pct':>5s}")
for name, c, s in rows[:int(sys.argv[4]) if len(sys.argv) > 4 else 30]:
    nm = re.sub(r'\s+', ' ', name)[:84]
    print(f"{nm:84s} {c/nsteps:10.0f} {s/1e6/nsteps:9.2f} {s/c/1e3:8.1f} {100*s/tot:5.1f}")
