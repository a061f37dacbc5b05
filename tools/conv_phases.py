#!/usr/bin/env python3
"""Per-workgroup phase timing (cycles) of single convolutions: staging / MFMA / epilogue."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smart_vocoder_amd import _native as N
lib = N.lib()
B = 16
print(f"{'C':>4} {'k':>2} {'d':>2} {'L':>7} {'res':>3} | {'stage0':>8} {'mfma':>8} {'epilog':>8} | span_cycles")
for (C, L) in ((32, 131072), (64, 65536), (128, 32768), (256, 4096)):
    for k in (3, 11):
        for res in (0, 1):
            x = torch.randn(B, C, L, device="cuda") * 0.5
            w = torch.randn(C, C, k, device="cuda") / (C * k) ** 0.5
            b = torch.randn(C, device="cuda") * 0.1
            y = torch.empty_like(x)
            out = (ctypes.c_double * 4)()
            N.check(lib.svoc_debug_conv_timing(N.stream_ptr(), N.ptr(x), N.ptr(w), N.ptr(b), N.ptr(x) if res else None, N.ptr(y), B, C, L, k, 3, out))
            print(f"{C:4d} {k:2d} {3:2d} {L:7d} {res:3d} | {out[0]:8.0f} {out[1]:8.0f} {out[2]:8.0f} | {out[3]:.0f}")
