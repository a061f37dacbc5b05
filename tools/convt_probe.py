"""A few launches of one upsampler (svoc_conv_transpose1d) for rocprofv3: python tools/convt_probe.py Ci Co k s L [B iters]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smart_vocoder_amd import _native as N

ci, co, k, s, L = [int(a) for a in sys.argv[1:6]]
B = int(sys.argv[6]) if len(sys.argv) > 6 else 16
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 3
g = torch.Generator(device="cpu").manual_seed(1)
x = torch.randn(B, ci, L, generator=g).cuda()
v = (torch.randn(ci, co, k, generator=g) / (ci * k / s) ** 0.5).cuda()
gg = torch.ones(ci, 1, 1).cuda()
b = torch.zeros(co).cuda()
y = torch.empty(B, co, L * s, device="cuda")
for _ in range(iters):
    N.check(N.lib().svoc_conv_transpose1d(N.stream_ptr(), N.ptr(x), N.ptr(v), N.ptr(gg), N.ptr(b), N.ptr(y), B, ci, co, L, k, s, ctypes.c_float(0.1)))
torch.cuda.synchronize()
print("ok", float(y.abs().mean()))
