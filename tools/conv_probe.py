#!/usr/bin/env python3
"""Run one ResBlock1(C, k, (d,)) = [lrelu->conv(d) ; lrelu->conv(1)+res] on a decoder-stage-sized tensor, repeatedly.
    python tools/conv_probe.py C k d L [B] [iters]        (for rocprofv3 --pmc / --kernel-trace)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from cases import sw
from smart_vocoder_amd import modules, _native
C, k, d, L = (int(a) for a in sys.argv[1:5])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 16
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
m = modules.ResBlock1(C, k, (d,))
m.load_state_dict({n: torch.from_numpy(v) for n, v in sw.fill_state_dict({n: tuple(p.shape) for n, p in m.state_dict().items()}, 7, 1.0).items()})
m = m.cuda().eval()
x = torch.randn(B, C, L, device="cuda") * 0.5
y = m(x); torch.cuda.synchronize()
_native.profile_enable(True)
t0 = time.perf_counter()
for _ in range(iters):
    y = m(x)
torch.cuda.synchronize()
print(f"# C={C} k={k} d={d} L={L} B={B}: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms per pair")
print(_native.profile_report())
