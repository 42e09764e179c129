import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1])
rows = [(s, e, re.sub(r"\(.*", "", n)) for n, s, e in c.execute("select name,start,end from kernels")]
rows.sort()
acc = [(s, e) for s, e, n in rows if "accum0_f9" in n]
sc = [(s, e, n) for s, e, n in rows if "scatter_lds" in n or "hist_lds" in n]
tot = ov = 0
for s, e, n in sc:
    tot += e - s
    for a, b in acc:
        lo, hi = max(s, a), min(e, b)
        if hi > lo:
            ov += hi - lo
print("sort kernels: %.2f ms total, %.2f ms inside a gather kernel's interval (%.0f %%)" % (tot / 1e6, ov / 1e6, 100.0 * ov / max(1, tot)))
# timeline of the last step
last = [r for r in rows if "accum0_f9" in r[2] or "scatter_lds<0" in r[2]][-14:]
t0 = last[0][0]
for s, e, n in last:
    print("%-32s start %8.3f ms  dur %7.3f ms" % (n[:32], (s - t0) / 1e6, (e - s) / 1e6))
