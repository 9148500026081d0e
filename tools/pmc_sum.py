#!/usr/bin/env python3
"""Per-dispatch means of the counters of one rocprofv3 --pmc pass (counter_collection.csv), by kernel name and grid size.
usage: pmc_sum.py <dir or csv> [kernel substring]      -> one JSON line per (kernel, grid)"""
import collections, csv, glob, json, os, sys
path = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in files:
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            key = (r["Kernel_Name"].split("(")[0][:90], int(r.get("Grid_Size", 0)))
            acc[key][r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
for (k, g), disp in sorted(acc.items()):
    names = sorted({c for d in disp.values() for c in d})
    vals = list(disp.values())
    drop = 2 if len(vals) > 4 else 0           # warm-up launches
    vals = vals[drop:]
    print(json.dumps({"kernel": k, "grid": g, "dispatches": len(vals), **{c: sum(v[c] for v in vals) / len(vals) for c in names}}))
