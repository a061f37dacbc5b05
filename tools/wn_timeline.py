"""Phase stamps of the WN layer kernel (wn_layer_fused_ks_kernel, wave 0 of every workgroup, shader cycles):
    python tools/wn_timeline.py [B] [T]"""
import os, sys
import numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from cases import sw
from smart_vocoder_amd import modules, _native as N
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
lib = N.lib()
m = modules.WN(192, 5, 1, 16)
m.load_state_dict({n: torch.from_numpy(v) for n, v in sw.fill_state_dict({n: tuple(p.shape) for n, p in m.state_dict().items()}, 7, 0.5).items()})
m = m.cuda().eval()
x = torch.randn(B, 192, T, device="cuda") * 0.5; mask = torch.ones(B, 1, T, device="cuda")
for _ in range(5): m(x, mask)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): m(x, mask)
e1.record(); torch.cuda.synchronize()
print(f"WN(192, k5, 16 layers) B={B} T={T}: {e0.elapsed_time(e1) / 160 * 1e3:.1f} us per layer (events, unstamped)")
buf = torch.zeros(1 << 14, 8, dtype=torch.long, device="cuda"); torch.cuda.synchronize()
N.check(lib.svoc_debug_set_stamp_buffer(N.ptr(buf))); m(x, mask); torch.cuda.synchronize(); N.check(lib.svoc_debug_set_stamp_buffer(None))
D = buf.cpu().numpy(); D = D[D[:, 6] != 0]          # stamps of the LAST layer that wrote each slot
if not len(D): print("(no stamps: the persistent stack launch (SVOC_WN_STACK=0 for the per-layer kernel) or the short-input path ran)"); sys.exit(0)
if (D[:, 6] < 0).all():      # wn_layer_f25_kernel (round 4): Winograd F(2,5) in_layer
    D[:, 6] = -D[:, 6]
    names = ["staging (loads -> LDS, barrier)", "input transform -> planes, barrier", "phase A (in_layer, F(2,5): 576 MFMAs 16x16x4 per wave)",
             "output transform + exchange + gate + barrier", "phase B (res_skip MFMAs)", "exchange + barrier + epilogue"]
    tot = D[:, 6] - D[:, 0]
    print(f"{len(D)} workgroups (wn_layer_f25_kernel); wave 0, shader cycles: total {tot.mean():.0f} (= {tot.mean() / 2400:.1f} us at 2.4 GHz)")
    sub = D[:, 7]
    a, b_, c = (sub >> 42) & 0x1fffff, (sub >> 21) & 0x1fffff, sub & 0x1fffff
    print(f"   (after the stream: output transform + partial-sum writes {a.mean():.0f} | barrier {b_.mean():.0f} | gate {c.mean():.0f} | bias loads + barrier {(D[:, 4] - D[:, 3] - a - b_ - c).mean():.0f})")
    for i, n in enumerate(names):
        d = D[:, i + 1] - D[:, i]
        print(f"   {n:60s} {d.mean():8.0f}  ({100 * d.mean() / tot.mean():4.1f} %)")
    sys.exit(0)
names = ["staging (loads -> LDS, barrier)", "phase A (in_layer MFMAs)", "exchange + gate + barrier", "phase B (res_skip MFMAs)", "exchange + barrier", "epilogue"]
tot = D[:, 6] - D[:, 0]
print(f"{len(D)} workgroups; wave 0, shader cycles: total {tot.mean():.0f} (= {tot.mean() / 2400:.1f} us at 2.4 GHz)")
sub = D[:, 7]
a, b_, c = (sub >> 42) & 0x1fffff, (sub >> 21) & 0x1fffff, sub & 0x1fffff
print(f"   (exchange + gate + barrier = MFMA drain + partial-sum writes {a.mean():.0f} | barrier {b_.mean():.0f} | gate {c.mean():.0f} | bias loads + barrier {(D[:, 3] - D[:, 2] - a - b_ - c).mean():.0f})")
for i, n in enumerate(names):
    d = D[:, i + 1] - D[:, i]
    print(f"   {n:34s} {d.mean():8.0f}  ({100 * d.mean() / tot.mean():4.1f} %)")
