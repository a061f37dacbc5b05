#!/usr/bin/env python3
"""Phase cycle stamps of resblock_fused_kernel (stage | c1 MFMA | residual+activation exchange | c2 MFMA | epilogue).
    python tools/fused_phases.py [C k d L B]..."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from cases import sw
from smart_vocoder_amd import modules, _native as N
lib = N.lib()
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(64, 3, 1, 65536, 16), (64, 11, 5, 65536, 16), (32, 3, 1, 131072, 16), (32, 11, 5, 131072, 16)]
print(f"{'C':>3} {'k':>2} {'d':>2} {'L':>7} | {'stage':>7} {'c1':>7} {'xchg':>7} {'c2':>7} {'epi':>7} {'total':>8} | blocks")
for (C, k, d, L, B) in shapes:
    m = modules.ResBlock1(C, k, (d,))
    m.load_state_dict({n: torch.from_numpy(v) for n, v in sw.fill_state_dict({n: tuple(p.shape) for n, p in m.state_dict().items()}, 7, 1.0).items()})
    m = m.cuda().eval()
    x = torch.randn(B, C, L, device="cuda") * 0.5
    for _ in range(5):
        m(x)
    buf = torch.zeros(1 << 20, 8, dtype=torch.long, device="cuda")
    torch.cuda.synchronize()
    N.check(lib.svoc_debug_set_stamp_buffer(N.ptr(buf)))
    m(x); torch.cuda.synchronize()
    N.check(lib.svoc_debug_set_stamp_buffer(None))
    d_ = buf.cpu().numpy(); d_ = d_[d_[:, 5] != 0]
    ph = np.diff(d_[:, :6], axis=1).mean(axis=0)
    print(f"{C:3d} {k:2d} {d:2d} {L:7d} | " + " ".join(f"{v:7.0f}" for v in ph) + f" {ph.sum():8.0f} | {len(d_)}")
