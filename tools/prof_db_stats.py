"""Per-kernel totals and inter-kernel gaps from a rocprofv3 results .db (kernel-trace): python tools/prof_db_stats.py x.db [--csv out.csv]"""
import collections, sqlite3, sys
import numpy as np
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start"))
def short(n):
    for k in ("oplist", "newton", "pmat", "copyBuffer", "reduce", "eigfrags", "fitch", "gather", "k_sh", "gamma20"):
        if k in n: return k
    return n[:30]
tot = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in rows: tot[n][0] += 1; tot[n][1] += (e - s) / 1e3
span = (rows[-1][2] - rows[0][1]) / 1e6; busy = sum(v[1] for v in tot.values()) / 1e3
lines = ["Name,Calls,TotalDurationUs,AverageUs,Percentage"]
for n, (c, t) in sorted(tot.items(), key=lambda x: -x[1][1]):
    print("%-64s calls %6d total %10.1f us avg %8.2f us" % (n[:64], c, t, t / c)); lines.append('"%s",%d,%.3f,%.3f,%.3f' % (n, c, t, t / c, 100 * t / (busy * 1e3)))
print("span %.1f ms, kernels busy %.1f ms (%.0f %%)" % (span, busy, 100 * busy / span))
gaps = collections.defaultdict(list); prev = None
for n, s, e in rows:
    if prev is not None: gaps[short(prev[0]) + "->" + short(n)].append((s - prev[2]) / 1e3)
    prev = (n, s, e)
for k, l in sorted(gaps.items(), key=lambda x: -sum(x[1]))[:8]:
    l = np.array(l); print("gap %-22s n %5d total %8.1f ms  median %8.2f us  max %9.1f us" % (k, len(l), l.sum() / 1e3, np.median(l), l.max()))
if "--csv" in sys.argv: open(sys.argv[sys.argv.index("--csv") + 1], "w").write("\n".join(lines) + "\n")
