#!/usr/bin/env python3
"""Micro-benchmark of ONE fused ResBlock1 iteration (resblock_fused kernels) per (C, k, d) at the headline stage sizes,
with a same-process A/B against another kernel generation (SVOC_FUSE_V) and a bit-equality check.
    python tools/rb_bench.py [--b 16] [--ab 1,2] [--iters 20]"""
import argparse, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cases import sw
from smart_vocoder_amd import modules, _native as N
ap = argparse.ArgumentParser()
ap.add_argument("--b", type=int, default=16)
ap.add_argument("--ab", default="1,2")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--frames", type=int, default=512)
args = ap.parse_args()
vers = [v for v in args.ab.split(",")]
lib = N.lib()
print(f"{'C':>3} {'k':>2} {'d':>2} {'L':>7} | " + " | ".join(f"v{v}: {'us':>8} {'TFLOP/s':>7}" for v in vers) + " | max|diff| vs first")
tot = {v: 0.0 for v in vers}
for C, L in ((64, args.frames * 128), (32, args.frames * 256)):
    for k in (3, 7, 11):
        for d in (1, 3, 5):
            m = modules.ResBlock1(C, k, (d,))
            m.load_state_dict({n: torch.from_numpy(v) for n, v in sw.fill_state_dict({n: tuple(p.shape) for n, p in m.state_dict().items()}, 7, 1.0).items()})
            m = m.cuda().eval()
            x = torch.randn(args.b, C, L, device="cuda") * 0.5
            flops = 2 * 2.0 * C * C * k * args.b * L
            outs, cols = [], []
            best = {v: 1e30 for v in vers}

            def setv(v):
                os.environ["SVOC_FUSE_V"] = v.split(":")[0]
                os.environ["SVOC_RB_STAGGER"] = "0"; os.environ["SVOC_RB_PRIO"] = "0"; os.environ["SVOC_RB_FORCEU"] = "0"
                for kv in v.split(":")[1:]:
                    kk, vv = kv.split("=")
                    os.environ["SVOC_RB_" + kk] = vv
            setv(vers[0])
            for _ in range(40):                      # clocks up after the host-side module setup
                m(x)
            torch.cuda.synchronize()
            for rep in range(3):
                order = vers[rep % len(vers):] + vers[:rep % len(vers)]
                for v in order:
                    setv(v)
                    for _ in range(3):
                        y = m(x)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(args.iters):
                        y = m(x)
                    e1.record(); torch.cuda.synchronize()
                    best[v] = min(best[v], e0.elapsed_time(e1) * 1e3 / args.iters)
            for v in vers:
                setv(v)
                outs.append(m(x).clone())
                tot[v] += best[v]
                cols.append(f"    {best[v]:8.1f} {flops / best[v] / 1e6:7.1f}")
            diff = max(float((o - outs[0]).abs().max()) for o in outs)
            print(f"{C:3d} {k:2d} {d:2d} {L:7d} | " + " | ".join(cols) + f" | {diff:.2e}", flush=True)
print("sum us: " + "  ".join(f"v{v}={tot[v]:.0f}" for v in vers))
