#!/usr/bin/env python3
"""Condense a rocprofv3 kernel trace (CSV) into a per-(kernel, grid) table.

    python tools/prof_summary.py <dir-or-csv> [--skip-first N] > profiles/<name>.txt

Groups dispatches by (short kernel name, grid, workgroup, LDS) and prints count, total ms, mean/min/max us and
share of the total, sorted by total time.
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"conv_mfma_kernel<(\d+), *(\d+), *(\d+), *(\d+)>", name)
    if m:
        return "conv_mfma<WM%s,WN%s,MR%s,NR%s>" % m.groups()
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:70]


def main():
    path = sys.argv[1]
    skip = int(sys.argv[sys.argv.index("--skip-first") + 1]) if "--skip-first" in sys.argv else 0
    files = [path] if path.endswith(".csv") else sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))
    if not files:
        sys.exit("no *kernel_trace.csv under " + path)
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[skip:]
    groups = defaultdict(list)
    for r in rows:
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        key = (short(r["Kernel_Name"]), "x".join(r.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")),
               r.get("Workgroup_Size_X", "?"), r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?"), r.get("Accum_VGPR_Count", "?"))
        groups[key].append(dur)
    total = sum(sum(v) for v in groups.values())
    span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6 if rows else 0
    print(f"# {len(rows)} dispatches, kernel time {total / 1e3:.3f} ms, wall span {span:.3f} ms, files: {', '.join(os.path.basename(f) for f in files)}")
    print(f"{'kernel':52s} {'grid(threads)':>22s} {'wg':>4s} {'lds':>6s} {'vgpr':>5s} {'agpr':>5s} {'n':>5s} {'total_ms':>9s} {'mean_us':>9s} {'min_us':>8s} {'max_us':>8s} {'share':>6s}")
    for key, v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        print(f"{key[0]:52s} {key[1]:>22s} {key[2]:>4s} {key[3]:>6s} {key[4]:>5s} {key[5]:>5s} {len(v):5d} {sum(v) / 1e3:9.3f} {sum(v) / len(v):9.1f} {min(v):8.1f} {max(v):8.1f} {100 * sum(v) / total:5.1f}%")
    byname = defaultdict(float)
    for key, v in groups.items():
        byname[key[0]] += sum(v)
    print("\n# by kernel")
    for k, t in sorted(byname.items(), key=lambda kv: -kv[1]):
        print(f"{k:52s} {t / 1e3:9.3f} ms {100 * t / total:5.1f}%")


if __name__ == "__main__":
    main()
