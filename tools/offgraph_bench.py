#!/usr/bin/env python3
"""Time and achieved HBM rate of the module-level (off-graph) ops: DDSConv, ConvFlow, LayerNorm, the rational-quadratic spline
and the mel front-end.  Bytes = algorithmic traffic of the op as a whole (every tensor a layer must read / write once), so the
rate is a lower bound on what the kernels move.  python tools/offgraph_bench.py [B] [T]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from smart_vocoder_amd import modules, transforms, mel_processing
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = "cuda"
g = torch.Generator().manual_seed(3)

def init(m):
    for p in m.parameters():
        with torch.no_grad():
            p.copy_(torch.randn(p.shape, generator=g) * 0.1 + (1.0 if p.dim() == 1 and "gamma" in "gamma" else 0.0) * 0)
    return m.to(dev).eval()

def bench(name, fn, nbytes, n=20, flops=0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    extra = f"  {flops / ms * 1e-9:6.1f} TFLOP/s (MFMA-bound op)" if flops else ""
    print(f"{name:58s} {ms * 1e3:9.1f} us  {nbytes / ms * 1e-6:8.1f} GB/s ({nbytes / ms * 1e-6 / 8000 * 100:4.1f} % of 8 TB/s){extra}", flush=True)

print(f"# B={B} T={T}")
C = 192
x = torch.randn(B, C, T, generator=g).to(dev); mask = torch.ones(B, 1, T, device=dev)
ct = B * C * T * 4
dds = init(modules.DDSConv(C, 3, 3))
# per layer: dw+LN+GELU (read x, write y1), 1x1 (read y1, write y2), LN+GELU+residual (read y2, read+write x)  -> 7 passes; + add/copy 4
bench(f"DDSConv(192, k3, 3 layers) [{B},{C},{T}]", lambda: dds(x, mask), (7 * 3 + 4) * ct)
ln = init(modules.LayerNorm(C))
bench(f"LayerNorm(192) [{B},{C},{T}]", lambda: ln(x), 2 * ct)
cf = init(modules.ConvFlow(2, C, 3, 3))
x2 = torch.randn(B, 2, T, generator=g).to(dev) * 0.5
# pre 1x1 (1 -> 192), DDSConv, proj 1x1 (192 -> 29), spline on T elements of 29 parameters
bench(f"ConvFlow(2, 192, k3, 3 layers, 10 bins) [{B},2,{T}]", lambda: cf(x2, mask), (7 * 3 + 4 + 2) * ct + 2 * B * 29 * T * 4)
nb = 10
n = B * T
xs = (torch.rand(n, generator=g) * 2 - 1).to(dev) * 4
uw = torch.randn(n, nb, generator=g).to(dev); uh = torch.randn(n, nb, generator=g).to(dev); ud = torch.randn(n, nb - 1, generator=g).to(dev)
bench(f"rq spline, linear tails, {nb} bins, {n} elements", lambda: transforms.piecewise_rational_quadratic_transform(xs, uw, uh, ud, inverse=False, tails="linear", tail_bound=5.0),
      n * (1 + 3 * nb - 1 + 2) * 4)
wav = (torch.rand(B, T * 256, generator=g) * 2 - 1).to(dev) * 0.9
nfr = T
bench(f"mel_spectrogram_torch(1024, 80, hop 256) [{B},{T * 256}]", lambda: mel_processing.mel_spectrogram_torch(wav, 1024, 80, 22050, 256, 1024, 0, None),
      B * T * 256 * 4 + B * 80 * nfr * 4, n=5, flops=2.0 * B * nfr * (2 * 513 * 1024 + 80 * 513))
