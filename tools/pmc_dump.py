#!/usr/bin/env python3
"""Mean per-dispatch value of every counter for kernels whose name contains a pattern.  python tools/pmc_dump.py pattern dir..."""
import csv, glob, os, sys
from collections import defaultdict
pat = sys.argv[1]
acc = defaultdict(list)
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:36s} n={len(v):3d} mean={sum(v) / len(v):16.1f}")
