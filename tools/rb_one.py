#!/usr/bin/env python3
"""Run one fused ResBlock iteration shape a few times (target for rocprofv3 --pmc).  python tools/rb_one.py C k d [iters]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cases import sw
from smart_vocoder_amd import modules
C, k, d = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
L = 512 * (128 if C == 64 else 256)
m = modules.ResBlock1(C, k, (d,))
m.load_state_dict({n: torch.from_numpy(v) for n, v in sw.fill_state_dict({n: tuple(p.shape) for n, p in m.state_dict().items()}, 7, 1.0).items()})
m = m.cuda().eval()
x = torch.randn(16, C, L, device="cuda") * 0.5
for _ in range(iters):
    m(x)
torch.cuda.synchronize()
