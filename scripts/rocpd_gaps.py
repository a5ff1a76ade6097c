"""Idle time between consecutive kernels of each HW queue over the LAST `ms` milliseconds of a rocprofv3 kernel trace:
    python scripts/rocpd_gaps.py <db> <ms> <steps>
Per queue: dispatches/step, busy ms/step, idle ms/step inside the queue's own span (start of kernel i+1 minus the latest end seen so
far, when positive), a histogram of those gaps, and the kernels that most often FOLLOW a gap > 3 us."""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor(); win_ms = float(sys.argv[2]); nsteps = float(sys.argv[3])
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'").fetchall()]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]; ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
dcol = [r[1] for r in cur.execute(f"pragma table_info({kd})").fetchall()]
scol = [r[1] for r in cur.execute(f"pragma table_info({ks})").fetchall()]
namecol = 'kernel_name' if 'kernel_name' in scol else 'display_name'
qcol = 'queue_id' if 'queue_id' in dcol else ('stream_id' if 'stream_id' in dcol else None)
print("dispatch columns:", dcol)
t1 = cur.execute(f"select max(end) from {kd}").fetchone()[0]; t0 = t1 - int(win_ms * 1e6)
rows = cur.execute(f"select d.start, d.end, {('d.' + qcol) if qcol else '0'}, s.{namecol} from {kd} d join {ks} s on d.kernel_id=s.id "
                   f"where d.start >= {t0} order by d.start").fetchall()
byq = collections.defaultdict(list)
for st, en, q, nm in rows:
    byq[q].append((st, en, nm))
edges = [0, 0.5, 1, 1.5, 2, 3, 5, 10, 20, 50, 1e9]
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, _ in rs) / 1e6 / nsteps
    last_end, idle, hist, after = rs[0][1], 0.0, [0] * (len(edges) - 1), collections.Counter()
    overlap = 0.0
    for s, e, nm in rs[1:]:
        g = (s - last_end) / 1e3
        if g > 0:
            idle += g
            for i in range(len(edges) - 1):
                if edges[i] <= g < edges[i + 1]:
                    hist[i] += 1
                    break
            if g > 3:
                after[re.sub(r'\s+', ' ', nm)[:70]] += g
        else:
            overlap += -g
        last_end = max(last_end, e)
    big = [i for i in range(1, len(rs)) if rs[i][0] - max(r[1] for r in rs[max(0, i - 4):i]) > 500e3]
    for i in big[-6:]:                     # context of the last few large gaps: three kernels either side
        print(f"   gap of {(rs[i][0] - rs[i - 1][1]) / 1e3:.0f} us between")
        for j in range(max(0, i - 3), min(len(rs), i + 3)):
            print(f"      {'>>' if j == i else '  '} {(rs[j][1] - rs[j][0]) / 1e3:9.1f} us  {re.sub(r'[ ]+', ' ', rs[j][2])[:110]}")
    span = (rs[-1][1] - rs[0][0]) / 1e6 / nsteps
    print(f"queue {q}: {len(rs)/nsteps:.0f} dispatches/step, busy {busy:.2f} ms/step, idle-in-span {idle/1e3/nsteps:.2f} ms/step, "
          f"span {span:.2f} ms/step, back-to-back overlap {overlap/1e3/nsteps:.2f} ms/step")
    print("   gap histogram (us): " + "  ".join(f"[{edges[i]}-{edges[i+1] if edges[i+1] < 1e8 else 'inf'}):{hist[i]/nsteps:.0f}" for i in range(len(hist))))
    for nm, g in after.most_common(8):
        print(f"   after-gap us/step {g/nsteps:8.1f}  {nm}")
