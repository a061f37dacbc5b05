#!/usr/bin/env python3
"""Instruction mix per kernel family from rocprofv3 --pmc CSVs (SQ_INSTS_VALU/SALU/LDS/VMEM/MFMA, SQ_VALU_MFMA_BUSY_CYCLES,
GRBM_GUI_ACTIVE collected in separate passes into the given directories).
    python tools/pmc_inst_mix.py <dir> [<dir> ...]"""
import csv, glob, os, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
def fam(n):
    for k in ("conv_wino4_group", "conv_wino4_acc3", "conv_wino4_pairacc", "conv_wino4_pair", "conv_wino4", "wn_stack_f25", "wn_layer_f25", "wn_small_f25", "convt_wino", "conv_wino_group", "conv_wino", "conv_group", "conv_ksplit", "conv_mfma", "resblock_fused", "wn_layer_fused", "conv_post"):
        if k in n:
            m = re.search(r"<([\d, ]+)>", n)
            return k + (("<" + m.group(1).replace(" ", "") + ">") if m else "")
    return None
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = fam(r["Kernel_Name"])
            if k:
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
names = ["SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]
print(f"{'kernel':40s} {'disp':>5s} {'MFMA(M)':>9s} {'VALU/MFMA':>9s} {'SALU/MFMA':>9s} {'LDS/MFMA':>9s} {'VMEM/MFMA':>9s} {'MfmaBusy%':>9s}")
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_INSTS_MFMA", 0)):
    a = acc[k]; m = a.get("SQ_INSTS_MFMA", 0) or 1
    valu = a.get("SQ_INSTS_VALU", 0) - a.get("SQ_INSTS_MFMA", 0)      # SQ_INSTS_VALU includes the MFMAs
    busy = 100.0 * a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (a.get("GRBM_GUI_ACTIVE", 0) / 8) if a.get("GRBM_GUI_ACTIVE") else float("nan")
    print(f"{k:40s} {cnt[k].get('SQ_INSTS_MFMA', 0):5d} {m / 1e6:9.1f} {valu / m:9.2f} {a.get('SQ_INSTS_SALU', 0) / m:9.2f} {a.get('SQ_INSTS_LDS', 0) / m:9.2f} "
          f"{a.get('SQ_INSTS_VMEM', 0) / m:9.2f} {busy:9.1f}")
