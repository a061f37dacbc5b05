#!/usr/bin/env python3
"""Mid-size batches: SynthesizerTrn.infer latency at a few (B, T) under the library's current environment (run once per
switch setting: the library reads its switches once per process).  Usage: [SVOC_WN_SMALL=0 ...] python tools/mid_batch_ab.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from cases import sw
from smart_vocoder_amd import models
net = models.SynthesizerTrn(513, 32, n_speakers=109, **cases.IITP_MODEL)
net.load_state_dict({k: torch.from_numpy(v) for k, v in cases.full_model_weights().items()}, strict=False)
net = net.cuda().eval()
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("SVOC_")) or "default"
for (B, T) in ((2, 512), (4, 512), (6, 512), (8, 512), (4, 1024), (12, 512)):
    mel = torch.from_numpy(sw.synthetic_mel(1, B, T)).cuda(); eps = torch.from_numpy(sw.synthetic_eps(1, B, T)).cuda()
    ln = torch.full((B,), T, dtype=torch.int64).cuda()
    for _ in range(4):
        net.infer(mel, ln, noise_scale=0.667, eps=eps)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            net.infer(mel, ln, noise_scale=0.667, eps=eps)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 10)
    print(f"[{tag}] B={B:3d} T={T:5d}: {best * 1e3:8.3f} ms  {B * T * 256 / best / 1e6:8.2f} M samples/s", flush=True)
