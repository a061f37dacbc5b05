#!/usr/bin/env python3
"""Per-launch time of the Winograd convolution kernel at the decoder's stage sizes (library event profiler).
    python tools/wino_bench.py [C=128] [L=32768] [B=16]   (SVOC_WINO_SKIP=1/2/4 for the phase-removal timing experiments)"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from smart_vocoder_amd import _native as N
C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
lib = N.lib()
x = torch.randn(B, C, L, device="cuda") * 0.5
y = torch.empty_like(x)
for k in [int(v) for v in os.environ.get("WB_K", "3,7,11").split(",")]:
    for d in [int(v) for v in os.environ.get("WB_D", "1,3,5").split(",")]:
        v = torch.randn(C, C, k, device="cuda") / (C * k) ** 0.5
        g = torch.rand(C, 1, 1, device="cuda") + 0.5
        b = torch.randn(C, device="cuda") * 0.1
        for _ in range(2):
            N.check(lib.svoc_conv1d_winograd(N.stream_ptr(), N.ptr(x), N.ptr(v), N.ptr(g), N.ptr(b), N.ptr(x), N.ptr(y), B, C, C, L, k, d, ctypes.c_float(0.1)))
        N.profile_enable(True)
        for _ in range(5):
            N.check(lib.svoc_conv1d_winograd(N.stream_ptr(), N.ptr(x), N.ptr(v), N.ptr(g), N.ptr(b), N.ptr(x), N.ptr(y), B, C, C, L, k, d, ctypes.c_float(0.1)))
        torch.cuda.synchronize()
        rep = N.profile_report().splitlines()
        print(rep[1], flush=True)
        N.profile_enable(False)
