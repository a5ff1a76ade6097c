"""Per-kernel PMC counter totals from a rocprofv3 --pmc run (rocpd sqlite)."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'").fetchall()]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
kd, ks, pe, pi = T('rocpd_kernel_dispatch'), T('rocpd_info_kernel_symbol'), T('rocpd_pmc_event'), T('rocpd_info_pmc')
scol = [r[1] for r in cur.execute(f"pragma table_info({ks})").fetchall()]
namecol = 'kernel_name' if 'kernel_name' in scol else 'display_name'
# optional third argument: only the LAST <win_ms> milliseconds of the trace (the timed graph replays of bench.py --no-host-timing) —
# totals over the whole command also contain the eager warm-up step and the capture pass, whose launch mix differs
where = ""
if len(sys.argv) > 3:
    t1 = cur.execute(f"select max(end) from {kd}").fetchone()[0]
    where = f"where d.start >= {t1 - int(float(sys.argv[3]) * 1e6)}"
q = f"select s.{namecol}, i.name, count(*), sum(e.value), sum(d.end-d.start) from {pe} e join {pi} i on e.pmc_id=i.id join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id {where} group by s.{namecol}, i.name order by 4 desc"
for n, c, k, v, t in cur.execute(q).fetchall()[:int(sys.argv[2]) if len(sys.argv) > 2 else 10]:
    print(f"{re.sub(r'[ ]+', ' ', n)[:140]:140s} {c:12s} n={k:6d} total={v:.4e} per_launch={v/k:.1f} avg_us={t/k/1e3:.1f}")
