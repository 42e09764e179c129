"""Per-kernel averages of rocprofv3 --pmc passes (counter_collection.csv files under the given directories).
Usage: python tools/pmc_summary.py OUT.json DIR [DIR ...]"""
import csv, glob, json, os, re, sys
out, dirs = sys.argv[1], sys.argv[2:]
acc = {}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = re.sub(r"\(.*", "", r["Kernel_Name"])
                c = r["Counter_Name"]
                e = acc.setdefault(k, {}).setdefault(c, [0.0, set()])
                e[0] += float(r["Counter_Value"])
                e[1].add(r.get("Dispatch_Id", len(e[1])))
res = {}
for k, cs in acc.items():
    res[k] = {}
    for c, (tot, ids) in cs.items():
        res[k][c + "_per_launch"] = round(tot / max(1, len(ids)), 1)
        res[k]["launches"] = len(ids)
json.dump({"kernels": res}, open(out, "w"), indent=1)
top = sorted(res.items(), key=lambda kv: -sum(v for n, v in kv[1].items() if n.endswith("_per_launch")) * kv[1]["launches"])[:12]
for k, v in top:
    print(k[:50], v)
