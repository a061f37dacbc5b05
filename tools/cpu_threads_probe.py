#!/usr/bin/env python3
"""Times the CPU oracle at several torch thread counts (to choose bench.py's cpu_baseline thread count)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from cases import sw
from oracle import vocoder_oracle as O
sd = {k: torch.from_numpy(v) for k, v in cases.full_model_weights().items()}
B, T = 1, 256
mel = torch.from_numpy(sw.synthetic_mel(1, B, T)); eps = torch.from_numpy(sw.synthetic_eps(1, B, T)); ln = torch.tensor([T] * B)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread' ")
for n in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64, 128]:
    torch.set_num_threads(n)
    with torch.no_grad():
        O.infer(sd, mel[:, :, :32], torch.tensor([32]), eps[:, :, :32], 0.667)
        t0 = time.perf_counter(); o, *_ = O.infer(sd, mel, ln, eps, 0.667); dt = time.perf_counter() - t0
    print(f"threads {n:4d}: {dt:7.2f} s  {o.numel() / dt:10.0f} samples/s", flush=True)
