import ctypes, os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from smart_vocoder_amd import _native as N
C, L, B = 128, 32768, 16
lib = N.lib()
x = torch.randn(B, C, L, device="cuda") * 0.5
y = torch.empty_like(x)
for k in (3, 7, 11):
    v = torch.randn(C, C, k, device="cuda") / (C * k) ** 0.5
    g = torch.rand(C, 1, 1, device="cuda") + 0.5
    b = torch.randn(C, device="cuda") * 0.1
    for _ in range(3):
        N.check(lib.svoc_conv1d_winograd(N.stream_ptr(), N.ptr(x), N.ptr(v), N.ptr(g), N.ptr(b), N.ptr(x), N.ptr(y), B, C, C, L, k, 1, ctypes.c_float(0.1)))
torch.cuda.synchronize()
