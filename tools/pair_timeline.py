"""Per-workgroup stamps of the fused c1 -> c2 pair kernel (csrc/conv_wino4_pair.hip, stamped instantiation): consumer wave 0's barrier waits / MFMA streams / epilogues of
both convolutions per tile, producer wave 0's barrier waits, effective shader clock.  The kernel runs inside a one-stage decoder (conv_pre -> x2 upsampler -> MRF at C channels).
    python tools/pair_timeline.py [C=32] [B=16] [L=131072]        (C = 32: the d = 1 and d = 3 pairs; C = 64: the d = 1 pair)"""
import os, sys
import numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import cases
from cases import sw
from smart_vocoder_amd import models, _native as N
C = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
L = int(sys.argv[3]) if len(sys.argv) > 3 else 131072
lib = N.lib()
c = dict(initial_channel=32, resblock="1", rks=[3, 7, 11], rds=[[1, 3, 5]] * 3, ur=[2], uic=2 * C, uks=[4], gin=0)
sd = sw.fill_state_dict(cases.generator_shapes(c), 7761, 1.0)
m = models.Generator(c["initial_channel"], c["resblock"], c["rks"], c["rds"], c["ur"], c["uic"], c["uks"], gin_channels=0)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m = m.cuda().eval()
x = torch.randn(B, 32, L // 2, device="cuda")
for _ in range(3): m(x)
torch.cuda.synchronize()
N.profile_enable(True); m(x); torch.cuda.synchronize()
print("\n".join(l for l in N.profile_report().splitlines() if "wino4P" in l)); N.profile_enable(False)
buf = torch.zeros(1 << 15, 16, dtype=torch.long, device="cuda"); torch.cuda.synchronize()
N.check(lib.svoc_debug_set_stamp_buffer(N.ptr(buf)))
for _ in range(3): m(x)
torch.cuda.synchronize(); N.check(lib.svoc_debug_set_stamp_buffer(None))
D = buf.cpu().numpy()
for di, d1 in enumerate((1, 3, 5)):
    for mi, k in enumerate((11, 7, 3)):
        S = D[8192 + (di * 3 + mi) * 1024: 8192 + (di * 3 + mi + 1) * 1024]; S = S[S[:, 0] != 0]
        if not len(S): continue
        t = S[:, 0].astype(float)
        per = lambda col: float((S[:, col] / t).mean())
        tot = per(1); clk = float((S[:, 1] / ((S[:, 11] - S[:, 10]) * 10e-9)).mean()) / 1e6
        G = (k + 1) // 4; nm = 16 * (7 * G if k >= 7 else 6 * G + 4 * (G - 1)) * (C // 32) * (2 if C == 64 else 1)
        print(f"C={C} k={k} c1 d={d1}: {len(S)} workgroups x {t.mean():.1f} tiles; per tile (consumer wave 0), cycles: total {tot:.0f} | c1: barrier waits {per(2):.0f}, MFMA streams {per(3):.0f}, "
              f"epilogue -> LDS tile {per(4):.0f} | c2: barrier waits {per(5):.0f}, MFMA streams {per(6):.0f}, epilogue + x -> global {per(7):.0f} | rest {tot - sum(per(i) for i in range(2, 8)):.0f}")
        print(f"      producer wave 0 per tile: total {per(8):.0f}, waiting at barriers {per(9):.0f}; effective shader clock {clk:.0f} MHz")
