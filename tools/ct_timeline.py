"""Per-workgroup stamps of the F(4,2) upsampler kernel (convt_wino.hip, stamped build): consumer wave 0's cycles in barrier waits /
MFMA streams / epilogues, producer wave 0's waits, effective shader clock.   python tools/ct_timeline.py Ci Co k s L [B]"""
import ctypes, os, sys
import numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
from smart_vocoder_amd import _native as N
ci, co, k, s, L = (int(v) for v in sys.argv[1:6]); B = int(sys.argv[6]) if len(sys.argv) > 6 else 16
lib = N.lib()
g = torch.Generator().manual_seed(1)
x = (torch.randn(B, ci, L, generator=g) * 0.5).cuda(); wv = (torch.randn(ci, co, k, generator=g) / (ci * k / s) ** 0.5).cuda()
wg = torch.ones(ci, 1, 1).cuda(); bias = torch.zeros(co, device="cuda"); y = torch.empty(B, co, L * s, device="cuda")
run = lambda: N.check(lib.svoc_conv_transpose1d(N.stream_ptr(), N.ptr(x), N.ptr(wv), N.ptr(wg), N.ptr(bias), N.ptr(y), B, ci, co, L, k, s, ctypes.c_float(float(os.environ.get("CT_SLOPE", "0.1")))))
for _ in range(5): run()
buf = torch.zeros(1 << 12, 16, dtype=torch.long, device="cuda"); torch.cuda.synchronize()
N.check(lib.svoc_debug_set_stamp_buffer(N.ptr(buf))); run(); torch.cuda.synchronize(); N.check(lib.svoc_debug_set_stamp_buffer(None))
D = buf.cpu().numpy(); D = D[D[:, 5] == 5]
nm = (ci // 2) * 5                          # MFMAs per tile and consumer
tiles = D[:, 0]; tot = D[:, 1] / tiles; bar = D[:, 2] / tiles; mf = D[:, 3] / tiles; epi = D[:, 4] / tiles
print(f"Ci={ci} Co={co} k={k} s={s} L={L} B={B}: {len(D)} workgroups x {tiles.mean():.1f} tiles; per tile (consumer wave 0), cycles: total {tot.mean():.0f} | "
      f"barrier waits {bar.mean():.0f} | MFMA streams {mf.mean():.0f} ({mf.mean() / nm:.1f} per MFMA, {nm} MFMAs) | epilogue {epi.mean():.0f} | rest {np.mean(tot - bar - mf - epi):.0f}")
print(f"   producer wave 0 per tile: total {np.mean(D[:, 8] / tiles):.0f} cycles, of which waiting at stage barriers {np.mean(D[:, 9] / tiles):.0f}, "
      f"for the global loads {np.mean(D[:, 12] / tiles):.0f}, staging (leaky relu -> raw tile) {np.mean(D[:, 13] / tiles):.0f}, next loads + transform {np.mean(D[:, 14] / tiles):.0f}")
rows = co * s
cps = 2 if (rows % 256 == 0) else 1         # consumer waves per SIMD: the 256-row layout (NRT = 8) runs eight consumers on four SIMDs
print(f"   matrix pipe of a SIMD: {cps} consumer wave(s) x {nm} MFMAs x 64 cycles = {cps * nm * 64} of the tile's {tot.mean():.0f} cycles ({100.0 * cps * nm * 64 / tot.mean():.0f} %)"
      + ("   (two consumers per SIMD: wave 0's barrier waits are mostly its sibling's stream, and the producers only advance between the streams - neither is a stall of the pipe)" if cps == 2 else ""))
wall = (D[:, 11] - D[:, 10]) / 100.0
print(f"   wall time per workgroup: median {np.median(wall):.1f} us, span first start .. last end {(D[:, 11].max() - D[:, 10].min()) / 100.0:.1f} us; effective shader clock {np.median(D[:, 1] / wall):.0f} MHz (ONE launch behind an idle gap: the clock ramp, not the sustained clock - see tools/power_ablate.py)")
