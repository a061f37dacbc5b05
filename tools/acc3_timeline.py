"""Per-workgroup stamps of the merged accumulate kernel (csrc/conv_wino4_acc.hip, stamped instantiation): consumer wave 0's barrier waits / MFMA streams of the three members
(k = 3, 7, 11 into one set of accumulators) and the epilogue (three residuals, divide, store) per tile.  The kernel runs inside a one-stage decoder.
    python tools/acc3_timeline.py [C=128] [B=16] [L=32768]"""
import os, sys
import numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import cases
from cases import sw
from smart_vocoder_amd import models, _native as N
C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
L = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
lib = N.lib()
c = dict(initial_channel=32, resblock="1", rks=[3, 7, 11], rds=[[1, 3, 5]] * 3, ur=[2], uic=2 * C, uks=[4], gin=0)
sd = sw.fill_state_dict(cases.generator_shapes(c), 7762, 1.0)
m = models.Generator(c["initial_channel"], c["resblock"], c["rks"], c["rds"], c["ur"], c["uic"], c["uks"], gin_channels=0)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m = m.cuda().eval()
x = torch.randn(B, 32, L // 2, device="cuda")
for _ in range(3): m(x)
torch.cuda.synchronize()
N.profile_enable(True); m(x); torch.cuda.synchronize()
print("\n".join(l for l in N.profile_report().splitlines() if "wino4A" in l)); N.profile_enable(False)
buf = torch.zeros(1 << 15, 16, dtype=torch.long, device="cuda"); torch.cuda.synchronize()
N.check(lib.svoc_debug_set_stamp_buffer(N.ptr(buf)))
for _ in range(3): m(x)
torch.cuda.synchronize(); N.check(lib.svoc_debug_set_stamp_buffer(None))
S = buf.cpu().numpy()[20000:20000 + 1024]; S = S[S[:, 0] != 0]
if not len(S): print("(no stamps: the merged accumulate launch was not taken)"); sys.exit(0)
t = S[:, 0].astype(float); per = lambda col: float((S[:, col] / t).mean())
tot = per(1); clk = float((S[:, 1] / ((S[:, 11] - S[:, 10]) * 10e-9)).mean()) / 1e6
nm = lambda k: 16 * 7 * ((k + 1) // 4) * (C // 32)
print(f"C={C}: {len(S)} workgroups x {t.mean():.1f} tiles; per tile (consumer wave 0), cycles: total {tot:.0f} | k=3: barrier waits {per(2):.0f}, MFMA streams {per(3):.0f} ({per(3) / nm(3):.1f} per MFMA) | "
      f"k=7: {per(4):.0f}, {per(5):.0f} ({per(5) / nm(7):.1f}) | k=11: {per(6):.0f}, {per(7):.0f} ({per(7) / nm(11):.1f}) | epilogue {per(8):.0f} | rest {tot - sum(per(i) for i in range(2, 9)):.0f}; effective shader clock {clk:.0f} MHz")
