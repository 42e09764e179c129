"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals and, with --seq PATTERN, the launch-ordered
durations of the kernels whose name matches (to see which launches of a multi-level schedule are slow)."""
import argparse, csv, glob, os, re, sys
ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--seq", default=None)
ap.add_argument("--last", type=int, default=400, help="only the last N matching launches")
args = ap.parse_args()
files = glob.glob(os.path.join(args.dir, "**", "*kernel_trace.csv"), recursive=True)
if not files:
    sys.exit("no kernel_trace.csv under " + args.dir)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
tot = {}
for s, e, n in rows:
    k = re.sub(r"\(.*", "", n)
    t = tot.setdefault(k, [0, 0])
    t[0] += 1
    t[1] += e - s
span = rows[-1][1] - rows[0][0]
print("kernels: %d launches, busy %.2f ms, span %.2f ms" % (len(rows), sum(v[1] for v in tot.values()) / 1e6, span / 1e6))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-60s calls %6d total_ms %10.3f avg_us %10.1f" % (k[:60], v[0], v[1] / 1e6, v[1] / 1e3 / v[0]))
if args.seq:
    pat = re.compile(args.seq)
    sel = [(s, e, n) for s, e, n in rows if pat.search(n)][-args.last:]
    prev_end = None
    for s, e, n in sel:
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        print("%-44s dur_us %9.1f gap_us %8.1f" % (re.sub(r"\(.*", "", n)[:44], (e - s) / 1e3, gap))
        prev_end = e
