#!/usr/bin/env python3
"""Per-layer-shape timing of one SynthesizerTrn.infer using the library's event profiler.
    python tools/profile_infer.py [B] [T] [iters]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from cases import sw
from smart_vocoder_amd import models, _native
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
net = models.SynthesizerTrn(513, 32, n_speakers=109, **cases.IITP_MODEL)
net.load_state_dict({k: torch.from_numpy(v) for k, v in cases.full_model_weights().items()}, strict=False)
net = net.cuda().eval()
mel = torch.from_numpy(sw.synthetic_mel(1001, B, T)).cuda(); eps = torch.from_numpy(sw.synthetic_eps(1001, B, T)).cuda()
ln = torch.full((B,), T, dtype=torch.int64).cuda()
net.infer(mel, ln, noise_scale=0.667, eps=eps); torch.cuda.synchronize()
_native.profile_enable(True)
for _ in range(iters):
    net.infer(mel, ln, noise_scale=0.667, eps=eps)
torch.cuda.synchronize()
print(f"# B={B} T={T} iters={iters}")
print(_native.profile_report())
_native.profile_enable(False)
