"""Merge the per-signature GEMM table of one bench step (T2V_BENCH_SHAPE_TABLE: kernel timestamps inside the instrumented eager
step, default train mode) with the isolated back-to-back timings of scripts/gemm_vs_library.py on the same signatures:
    python scripts/instep_vs_isolated.py <shape table> <gemm_vs_library output> > profiles/rNN_gemm_instep_vs_isolated.txt
In-step rows of one (M, N, K, window) signature are summed over their variants (residual or not, rank fragment or not: forward and
backward-data launches of the layers with that geometry)."""
import re
import sys

shape_file, iso_file = sys.argv[1], sys.argv[2]
instep = {}
for line in open(shape_file):
    m = re.match(r"\((\d+), (\d+), (\d+), (\d+), (\d+), '([^']+)', '(\w+)', '([^']+)'\)\s+launches\s+(\d+)\s+ms\s+([\d.]+).*us/launch\s+([\d.]+)\s+GFLOP\s+([\d.]+)", line)
    if not m:
        continue
    M, N, rc, K, z, win, kern, res, n, ms, us, gf = m.groups()
    if kern != "nn":
        continue
    key = (int(M), int(N), int(K))
    a = instep.setdefault(key, [0, 0.0, 0.0])
    a[0] += int(n); a[1] += float(ms); a[2] += float(gf) * int(n)
print("# GEMM signatures of one C2 step (default train mode): in-step kernel time (all launches of the geometry: forward, backward-data,")
print("# with / without residual and rank fragment) next to the same geometry launched back to back in isolation (ours / vendor library)")
print(f"# {'M':>7s} {'N':>6s} {'K':>6s} | {'launches':>8s} {'in-step us':>10s} {'TF/s':>7s} | {'isolated us':>11s} {'TF/s':>7s} | {'library us':>10s} | in-step / isolated")
for line in open(iso_file):
    m = re.match(r"M=\s*(\d+) N=\s*(\d+)\+\d+ K=\s*(\d+) taps=\d+: ours\s+([\d.]+) us\s+([\d.]+) TF/s \| library\s+([\d.]+) us", line)
    if not m:
        continue
    M, N, K, us, tf, lus = m.groups()
    a = instep.get((int(M), int(N), int(K)))
    if a is None:
        continue
    ius = a[1] / a[0] * 1e3
    print(f"  {int(M):7d} {int(N):6d} {int(K):6d} | {a[0]:8d} {ius:10.1f} {a[2] / a[1]:7.1f} | {float(us):11.1f} {float(tf):7.1f} | {float(lus):10.1f} | {ius / float(us):.2f}")
