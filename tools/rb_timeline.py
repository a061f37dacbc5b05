#!/usr/bin/env python3
"""Per-CU phase timeline of the fused ResBlock kernel from its wall-clock stamps (100 MHz) and HW_ID / XCC_ID:
are co-resident workgroups in lock step (all staging, all MFMA, all storing at the same time)?
    python tools/rb_timeline.py C k d [B] [frames]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cases import sw
from smart_vocoder_amd import modules, _native as N
C, k, d = (int(v) for v in sys.argv[1:4])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 16
frames = int(sys.argv[5]) if len(sys.argv) > 5 else 512
L = frames * (128 if C == 64 else 256)
lib = N.lib()
m = modules.ResBlock1(C, k, (d,))
m.load_state_dict({n: torch.from_numpy(v) for n, v in sw.fill_state_dict({n: tuple(p.shape) for n, p in m.state_dict().items()}, 7, 1.0).items()})
m = m.cuda().eval()
x = torch.randn(B, C, L, device="cuda") * 0.5
for _ in range(30):
    m(x)
buf = torch.zeros(1 << 16, 8, dtype=torch.long, device="cuda")
torch.cuda.synchronize()
N.check(lib.svoc_debug_set_stamp_buffer(N.ptr(buf)))
m(x); torch.cuda.synchronize()
N.check(lib.svoc_debug_set_stamp_buffer(None))
D = buf.cpu().numpy(); D = D[D[:, 5] != 0]
t0 = D[:, 0].min()
T = (D[:, :6] - t0) * 0.01          # us
hw, xcc = D[:, 6], D[:, 7]
cu = ((xcc & 15) << 8) | (((hw >> 13) & 3) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
ph = np.diff(T, axis=1)
print(f"C={C} k={k} d={d} L={L} B={B}: {len(D)} workgroups on {len(np.unique(cu))} CUs, span {T[:, 5].max():.1f} us")
print("mean phase us: stage %.2f | c1 %.2f | exchange %.2f | c2 %.2f | epilogue %.2f | life %.2f" % (*ph.mean(axis=0), (T[:, 5] - T[:, 0]).mean()))
# per CU: time with >=1 workgroup inside an MFMA phase (c1: T1..T2, c2: T3..T4) and mean concurrency in MFMA phases
span = T[:, 5].max()
grid = np.arange(0, span, 0.05)
busy_any, conc_m, conc_s = [], [], []
hist = np.zeros(8)
for c in np.unique(cu):
    w = T[cu == c]
    in_m = np.zeros_like(grid); in_s = np.zeros_like(grid); alive = np.zeros_like(grid)
    for r in w:
        in_m += ((grid >= r[1]) & (grid < r[2])) | ((grid >= r[3]) & (grid < r[4]))
        in_s += ((grid >= r[0]) & (grid < r[1])) | ((grid >= r[4]) & (grid < r[5]))
        alive += (grid >= r[0]) & (grid < r[5])
    act = alive > 0
    hist += np.bincount(in_m[act].astype(int), minlength=8)[:8]
    busy_any.append((in_m[act] > 0).mean()); conc_m.append(in_m[act].mean()); conc_s.append(in_s[act].mean())
print("per CU: fraction of time with >=1 workgroup in an MFMA phase: mean %.3f min %.3f max %.3f" % (np.mean(busy_any), np.min(busy_any), np.max(busy_any)))
print("per CU: mean #workgroups in MFMA phases %.2f, in memory phases (stage/epilogue) %.2f" % (np.mean(conc_m), np.mean(conc_s)))
print("time share with n workgroups of the CU inside an MFMA phase, n = 0..5: " + " ".join(f"{v:.3f}" for v in (hist / hist.sum())[:6]))
# lock step: for the first CU list the first 12 workgroups
c = np.unique(cu)[0]
w = T[cu == c]; w = w[np.argsort(w[:, 0])]
print("CU %#x first workgroups (start | staged | c1 done | exchanged | c2 done | end):" % c)
for r in w[:14]:
    print("   " + " ".join(f"{v:8.2f}" for v in r))
# chip-wide: number of workgroups in staging as a function of time (is HBM demand bursty?)
st = np.zeros_like(grid)
for r in T:
    st += (grid >= r[0]) & (grid < r[1])
q = np.percentile(st, [5, 25, 50, 75, 95])
print("chip-wide workgroups in the staging phase at a time: p5 %.0f p25 %.0f median %.0f p75 %.0f p95 %.0f (uniform would be ~ constant)" % tuple(q))
