#!/usr/bin/env python3
"""Print per-dispatch PMC counters of the conv kernel from a rocprofv3 --pmc CSV directory."""
import csv, glob, os, sys
from collections import defaultdict
path = sys.argv[1]
files = sorted(glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True))
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
by = defaultdict(dict)
for r in rows:
    if "conv_mfma" not in r["Kernel_Name"]:
        continue
    by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    by[int(r["Dispatch_Id"])]["_grid"] = r.get("Grid_Size", "?")
ids = sorted(by)
names = sorted({k for d in by.values() for k in d if not k.startswith("_")})
print("dispatch grid " + " ".join(f"{n:>22s}" for n in names))
for i in ids[-int(sys.argv[2]) if len(sys.argv) > 2 else -6:]:
    print(f"{i:8d} {by[i]['_grid']:>8s} " + " ".join(f"{by[i].get(n, float('nan')):22.0f}" for n in names))
