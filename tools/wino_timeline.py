#!/usr/bin/env python3
"""Where does a Winograd convolution workgroup spend its life?  Per-workgroup stamps of conv_wino_kernel (wall clock start / end,
shader cycles of wave 0 inside MFMA phases / in staging + transform / in the epilogue) and HW_ID -> per-CU concurrency.
    python tools/wino_timeline.py C k d [B] [L]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smart_vocoder_amd import _native as N
C, k, d = (int(v) for v in sys.argv[1:4])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 16
L = int(sys.argv[5]) if len(sys.argv) > 5 else (32768 if C == 128 else (4096 if C == 256 else 65536))
lib = N.lib()
g = torch.Generator(device="cpu").manual_seed(1)
x = (torch.randn(B, C, L, generator=g) * 0.5).cuda()
wv = (torch.randn(C, C, k, generator=g) * 0.05).cuda()
wg = wv.flatten(1).norm(dim=1).view(C, 1, 1).contiguous()
bias = torch.zeros(C, device="cuda")
y = torch.empty_like(x)
def run():
    N.check(lib.svoc_conv1d_winograd(N.stream_ptr(), N.ptr(x), N.ptr(wv), N.ptr(wg), N.ptr(bias), N.ptr(x), N.ptr(y), B, C, C, L, k, d, ctypes.c_float(0.1)))
for _ in range(20):
    run()
buf = torch.zeros(1 << 16, 16, dtype=torch.long, device="cuda")
torch.cuda.synchronize()
N.check(lib.svoc_debug_set_stamp_buffer(N.ptr(buf)))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
N.check(lib.svoc_debug_set_stamp_buffer(None))
D = buf.cpu().numpy(); D = D[D[:, 5] != 0]
G = (k + 1) // 4; ND = G - 1; slots = 4 * G + ND
nch = C // 32
n_mfma = nch * 16 * (4 * G + 2 * ND)
t0 = D[:, 0].min()
S, E = (D[:, 0] - t0) * 0.01, (D[:, 1] - t0) * 0.01
life = E - S
clk = 2400.0   # MHz (shader clock; tools/clock_trace.py shows it pinned under this load)
mf, prep, epi = D[:, 2] / clk, D[:, 3] / clk, D[:, 4] / clk
pub, b1, tr = D[:, 8] / clk, D[:, 9] / clk, D[:, 10] / clk
hw, xcc = D[:, 6], D[:, 7]
cu = ((xcc & 15) << 8) | (((hw >> 13) & 3) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
print(f"C={C} k={k} d={d} L={L} B={B}: {len(D)} workgroups on {len(np.unique(cu))} CUs, kernel {e0.elapsed_time(e1) * 1e3:.1f} us (stamped), span {E.max():.1f} us")
print(f"per workgroup (wave 0), us: life {life.mean():.2f} | MFMA phases {mf.mean():.2f} | publish+transform+barriers {prep.mean():.2f} | epilogue {epi.mean():.2f} | other {np.mean(life - mf - prep - epi):.2f}")
print(f"   of which: wait for raw loads + publish {pub.mean():.2f} | barrier 1 {b1.mean():.2f} | transform {tr.mean():.2f} | barrier 2 {np.mean(prep - pub - b1 - tr):.2f}   ({nch} chunks)")
print(f"MFMAs per wave {n_mfma}: {mf.mean() * clk / n_mfma:.1f} shader cycles per MFMA inside MFMA phases (64 = the SIMD to itself, 128 = shared by two waves)")
conc = []
for c in np.unique(cu):
    s, e = S[cu == c], E[cu == c]
    grid = np.arange(s.min(), e.max(), 0.1)
    alive = ((grid[:, None] >= s[None]) & (grid[:, None] < e[None])).sum(1)
    conc.append(np.bincount(alive, minlength=4)[:4] / len(grid))
conc = np.mean(conc, axis=0)
print("per CU share of time with n resident workgroups, n = 0..3: " + " ".join(f"{v:.3f}" for v in conc))
pipe = len(D) * 4 * n_mfma * 64 / clk / (256 * 4)
print(f"matrix-pipe time {pipe:.1f} us = {pipe / E.max():.3f} of the span")
