"""Per-workgroup stamps of the F(4,3) Winograd kernel (conv_wino4.hip, stamped build): consumer wave 0's cycles in barrier
waits / MFMA streams / epilogues and producer wave 0's slack.   SVOC_WINO_F4=1 python tools/wino4_timeline.py C k d"""
import ctypes, os, sys
import numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
os.environ.setdefault("SVOC_WINO_F4", "1")
from smart_vocoder_amd import _native as N
C, k, d = (int(v) for v in sys.argv[1:4]); B = int(os.environ.get("WB", "16")); L = int(os.environ.get("WL", "0")) or (32768 if C == 128 else 4096)
lib = N.lib()
g = torch.Generator().manual_seed(1)
x = (torch.randn(B, C, L, generator=g) * 0.5).cuda(); wv = (torch.randn(C, C, k, generator=g) * 0.05).cuda()
wg = wv.flatten(1).norm(dim=1).view(C, 1, 1).contiguous(); bias = torch.zeros(C, device="cuda"); y = torch.empty_like(x)
run = lambda: N.check(lib.svoc_conv1d_winograd(N.stream_ptr(), N.ptr(x), N.ptr(wv), N.ptr(wg), N.ptr(bias), N.ptr(x), N.ptr(y), B, C, C, L, k, d, ctypes.c_float(0.1)))
for _ in range(3): run()
buf = torch.zeros(1 << 16, 16, dtype=torch.long, device="cuda"); torch.cuda.synchronize()
# Round 5: the stamped launch is the LAST of a back-to-back burst.  (svoc_conv1d_winograd packs its weights and synchronises at every call, so the
# burst is not gap-free - tools/power_ablate.py loops a ResBlock module for the steady-state clock; a single launch behind an idle gap reads
# ~2.05 GHz, the clock ramp of the power management, which rounds 3-4 took for the sustained clock.)
N.check(lib.svoc_debug_set_stamp_buffer(N.ptr(buf)))
for _ in range(int(os.environ.get("WBURST", "25"))): run()
torch.cuda.synchronize(); N.check(lib.svoc_debug_set_stamp_buffer(None))
D = buf.cpu().numpy(); D = D[D[:, 5] == 4]
G = (k + 1) // 4; ND = G - 1; nm = (C // 32) * 16 * (6 * G + 4 * ND)
if k >= 7 and os.environ.get("SVOC_WINO_F4") != "0": nm = (C // 32) * 16 * 7 * G      # F(4,4)
tiles = D[:, 0]; tot = D[:, 1] / tiles; bar = D[:, 2] / tiles; mf = D[:, 3] / tiles; epi = D[:, 4] / tiles
print(f"C={C} k={k} d={d}: {len(D)} workgroups x {tiles.mean():.1f} tiles; per tile (consumer wave 0), cycles: total {tot.mean():.0f} | barrier waits {bar.mean():.0f} | "
      f"MFMA streams {mf.mean():.0f} ({mf.mean() / nm:.1f} per MFMA, {nm} MFMAs) | epilogue {epi.mean():.0f} | rest {np.mean(tot - bar - mf - epi):.0f}")
if int(os.environ.get("SVOC_DBG_ABL", "0")) & 128: print(f"   drain of the vector-memory queue behind the epilogue (s_waitcnt vmcnt(0)): {np.mean(D[:, 12] / tiles):.0f} cycles per tile")
print(f"   producer wave 0 per tile: total {np.mean(D[:, 8] / tiles):.0f} cycles, of which waiting at stage barriers {np.mean(D[:, 9] / tiles):.0f}")
wall = (D[:, 11] - D[:, 10]) / 100.0     # us (100 MHz constant clock)
print(f"   wall time per workgroup: median {np.median(wall):.1f} us, span first start .. last end {(D[:, 11].max() - D[:, 10].min()) / 100.0:.1f} us; effective shader clock {np.median(D[:, 1] / wall):.0f} MHz (last launch of a burst; steady state: tools/power_ablate.py)")
T = D[:, 1] / 2400.0
print(f"   per workgroup total us: min {T.min():.1f} p10 {np.percentile(T, 10):.1f} median {np.median(T):.1f} p90 {np.percentile(T, 90):.1f} max {T.max():.1f}")
