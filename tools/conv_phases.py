#!/usr/bin/env python3
"""Per-workgroup phase timing (cycles) of single convolutions.
plain kernel (SVOC_WS=0): staging / MFMA / epilogue per workgroup; wave-specialised kernel: consumer wave 0's total
barrier-wait / MFMA / epilogue cycles over all its tiles."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smart_vocoder_amd import _native as N
lib = N.lib()
B = 16
shapes = [(int(a), int(b)) for a, b in (s.split("x") for s in sys.argv[1:])] or [(32, 131072), (64, 65536), (128, 32768), (256, 4096)]
print(f"{'C':>4} {'k':>2} {'d':>2} {'L':>7} {'res':>3} | {'phase0':>9} {'phase1':>9} {'phase2':>9} {'span':>10} {'span/blk':>8}")
for (C, L) in shapes:
    for k in (3, 11):
        for res in (0, 1):
            x = torch.randn(B, C, L, device="cuda") * 0.5
            w = torch.randn(C, C, k, device="cuda") / (C * k) ** 0.5
            b = torch.randn(C, device="cuda") * 0.1
            y = torch.empty_like(x)
            out = (ctypes.c_double * 4)()
            N.check(lib.svoc_debug_conv_timing(N.stream_ptr(), N.ptr(x), N.ptr(w), N.ptr(b), N.ptr(x) if res else None, N.ptr(y), B, C, L, k, 3, out))
            for _ in range(3):      # warm clocks: the call launches the convolution twice and reports the second
                N.check(lib.svoc_debug_conv_timing(N.stream_ptr(), N.ptr(x), N.ptr(w), N.ptr(b), N.ptr(x) if res else None, N.ptr(y), B, C, L, k, 3, out))
            blk = out[0] + out[1] + out[2]
            print(f"{C:4d} {k:2d} {3:2d} {L:7d} {res:3d} | {out[0]:9.0f} {out[1]:9.0f} {out[2]:9.0f} {out[3]:10.0f} {out[3] / blk if blk else 0:8.2f}")
