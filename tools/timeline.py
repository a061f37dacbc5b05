#!/usr/bin/env python3
"""Kernel timeline of the last traced step from a rocprofv3 --kernel-trace CSV directory: start offset, duration, queue
and the number of kernels running concurrently; plus total busy time vs the sum of kernel durations."""
import csv, glob, os, sys
path = sys.argv[1]
files = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Grid_Size", "?")) for r in rows))
# the last step: find the last sequence_mask kernel
idx = [i for i, k in enumerate(ks) if "sequence_mask" in k[2]]
ks = ks[idx[-1]:] if idx else ks
t0 = ks[0][0]
end = max(k[1] for k in ks)
busy = 0
cur_s, cur_e = None, None
for s, e, *_ in ks:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(k[1] - k[0] for k in ks)
print(f"# {len(ks)} kernels, span {(end - t0) / 1e6:.3f} ms, busy (union) {busy / 1e6:.3f} ms, sum of durations {tot / 1e6:.3f} ms")
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 400
print(f"{'start_us':>10} {'dur_us':>9} {'queue':>6} {'grid':>9}  kernel")
for s, e, n, q, g in ks[:lim]:
    short = n.split("(")[0].replace("void svoc::", "")[:70]
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} {q:>6} {g:>9}  {short}")
