"""Per-kernel totals of a rocprofv3 --kernel-trace run written as a rocpd database (the default output format):
  python tools/db_summary.py RESULTS.db [N]"""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name,start,end from kernels"))
tot = {}
for n, s, e in rows:
    k = re.sub(r"\(.*", "", n)
    t = tot.setdefault(k, [0, 0])
    t[0] += 1
    t[1] += e - s
span = max(r[2] for r in rows) - min(r[1] for r in rows)
print("kernels: %d launches, busy %.2f ms, span %.2f ms" % (len(rows), sum(v[1] for v in tot.values()) / 1e6, span / 1e6))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("%-60s calls %6d total_ms %10.3f avg_us %10.1f" % (k[:60], v[0], v[1] / 1e6, v[1] / 1e3 / v[0]))
