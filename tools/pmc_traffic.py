#!/usr/bin/env python3
"""Sum HBM traffic per bench step from rocprofv3 --pmc CSVs (two passes: read-request and write-request counters).
    python tools/pmc_traffic.py <read_dir> <write_dir> <dispatch-skip> <steps>   -> JSON on stdout
Bytes are resolved by request size (TCC_EA0_RDREQ_{32B,64B,128B}, WRREQ / WRREQ_64B), which avoids the 2x ambiguity of
FETCH_SIZE on gfx950 (MI355X_MICROARCH.md, HBM section).  bench.py imports traffic() for its live `roofline.traffic`."""
import csv, glob, json, os, sys
from collections import defaultdict

READ_COUNTERS = ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"]
WRITE_COUNTERS = ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"]


def load(path):
    by = defaultdict(dict)
    names = {}
    for f in sorted(glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
            names[int(r["Dispatch_Id"])] = r["Kernel_Name"]
    return by, names


def is_model(n):
    return any(k in n for k in ("conv_mfma", "conv_group", "conv_ksplit", "conv_wino", "convt_wino", "convt_tail", "resblock_fused",
                                "wn_layer_fused", "wn_layer_f25", "wn_stack_f25", "wn_small_f25", "conv_post", "copy2d", "sequence_mask"))


def gemm(n):
    return any(k in n for k in ("conv_mfma", "conv_group", "conv_ksplit", "conv_wino", "convt_wino", "resblock_fused", "wn_layer_fused", "wn_layer_f25",
                                "wn_stack_f25", "wn_small_f25"))


def rbytes(c):
    n32, n64, n128, tot = c.get("TCC_EA0_RDREQ_32B_sum", 0), c.get("TCC_EA0_RDREQ_64B_sum", 0), c.get("TCC_EA0_RDREQ_128B_sum", 0), c.get("TCC_EA0_RDREQ_sum", 0)
    other = max(0.0, tot - n32 - n64 - n128)
    return 32 * n32 + 64 * n64 + 128 * n128 + 64 * other


def wbytes(c):
    n64, tot = c.get("TCC_EA0_WRREQ_64B_sum", 0), c.get("TCC_EA0_WRREQ_sum", 0)
    return 64 * n64 + 32 * max(0.0, tot - n64)


def traffic(read_dir, write_dir, per_step=0, steps=1):
    """HBM bytes per inference step over the last `steps` steps of the two traces.  A step starts at its sequence_mask dispatch
    (as tools/timeline.py cuts the kernel trace); `per_step` (kernels per step) is only the fallback when there is none."""
    rd, rn = load(read_dir)
    wr, wn = load(write_dir)
    if not rd or not wr:
        raise RuntimeError("no counter_collection.csv under the given directories")

    def tail(by, names):
        ids = sorted(by)
        marks = [k for k, i in enumerate(ids) if "sequence_mask" in names[i]]
        if len(marks) >= steps:
            return ids[marks[-steps]:]
        ids = [i for i in ids if is_model(names[i])]
        return ids[-per_step * steps:] if per_step > 0 else ids

    rid, wid = tail(rd, rn), tail(wr, wn)
    R = sum(rbytes(rd[i]) for i in rid) / steps
    W = sum(wbytes(wr[i]) for i in wid) / steps
    Rg = sum(rbytes(rd[i]) for i in rid if gemm(rn[i])) / steps
    Wg = sum(wbytes(wr[i]) for i in wid if gemm(wn[i])) / steps
    ng = sum(1 for i in rid if gemm(rn[i])) / steps
    return {"hbm_read_bytes_per_step": R, "hbm_write_bytes_per_step": W, "gemm_family_read_bytes_per_step": Rg,
            "gemm_family_write_bytes_per_step": Wg, "gemm_family_launches_per_step": ng,
            "gemm_family_bytes_per_launch": (Rg + Wg) / max(1.0, ng), "dispatches_counted_per_step": len(rid) / steps,
            "method": "rocprofv3 --pmc, TCC_EA0_RDREQ_{32B,64B,128B}_sum and TCC_EA0_WRREQ{,_64B}_sum, separate passes, bench.py 16x512"}


def counter_per_step(path, name, steps=1):
    """Sum of counter `name` over the dispatches of the last `steps` inference steps (a step starts at its sequence_mask
    dispatch), per step.  bench.py: SQ_INSTS_VALU_MFMA_MOPS_F32 of the default launch plan -> `roofline.executed_flops_pmc`."""
    by, names = load(path)
    if not by:
        raise RuntimeError("no counter_collection.csv under " + path)
    ids = sorted(by)
    marks = [k for k, i in enumerate(ids) if "sequence_mask" in names[i]]
    if len(marks) < steps:
        raise RuntimeError("fewer sequence_mask dispatches than steps in " + path)
    return sum(by[i].get(name, 0.0) for i in ids[marks[-steps]:]) / steps


if __name__ == "__main__":
    print(json.dumps(traffic(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 0, int(sys.argv[4]) if len(sys.argv) > 4 else 1)))
