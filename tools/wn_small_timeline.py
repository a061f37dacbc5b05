"""Phase stamps of the short-input WN layer kernel (csrc/wn_small.hip): thread 0 of every workgroup of the LAST layer launch of a burst, 100 MHz wall clock.
    python tools/wn_small_timeline.py [B=4] [T=512] [layers=8]"""
import os, sys
import numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from cases import sw
from smart_vocoder_amd import modules, _native as N
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
NL = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lib = N.lib()
m = modules.WN(192, 5, 1, NL)
m.load_state_dict({n: torch.from_numpy(v) for n, v in sw.fill_state_dict({n: tuple(p.shape) for n, p in m.state_dict().items()}, 7, 0.5).items()})
m = m.cuda().eval()
x = torch.randn(B, 192, T, device="cuda") * 0.5; mask = torch.ones(B, 1, T, device="cuda")
for _ in range(20): m(x, mask)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): m(x, mask)
e1.record(); torch.cuda.synchronize()
print(f"WN(192, k5, {NL} layers) B={B} T={T}: {e0.elapsed_time(e1) / 100 * 1e3:.1f} us per call = {e0.elapsed_time(e1) / 100 / NL * 1e3:.2f} us per layer (events, no stamp buffer set)")
buf = torch.zeros(1 << 12, 16, dtype=torch.long, device="cuda"); torch.cuda.synchronize()
N.check(lib.svoc_debug_set_stamp_buffer(N.ptr(buf)))
for _ in range(20): m(x, mask)
torch.cuda.synchronize(); N.check(lib.svoc_debug_set_stamp_buffer(None))
D = buf.cpu().numpy(); D = D[D[:, 6] != 0]
if not len(D): print("(no stamps: the short-input chain was not taken at this shape)"); sys.exit(0)
two = bool((D[:, 10] != 0).all())
names = ["weight requests of the previous 1 x 1, staging x / acts / mask -> LDS, barrier", "previous layer's 1 x 1 (skip rows of the pair(s), residual rows of all), x update, barrier",
         "pair 1: input transform -> planes, barrier", "pair 1: this wave's 96 MFMAs (16x16x4) of the F(2,5) in_layer, weights three slots ahead",
         "pair 1: barrier, output transform of the partial sums -> LDS, barrier", "pair 1: reduction over the twelve K parts, bias, gate, store"]
if two:
    names += ["pair 2: barrier, input transform, barrier", "pair 2: in_layer MFMAs", "pair 2: barrier, output transform, barrier", "pair 2: reduction, gate, store"]
last = 10 if two else 6
tot = (D[:, last] - D[:, 0]) * 10.0
span = (D[:, last].max() - D[:, 0].min()) * 10.0
print(f"{len(D)} workgroups of the last layer's launch; thread 0, ns: in-kernel span {tot.mean():.0f} = {tot.mean() / 1e3:.1f} us; first start .. last end of the launch {span / 1e3:.1f} us")
for i, n in enumerate(names):
    d = (D[:, i + 1] - D[:, i]) * 10.0
    print(f"   {n:110s} mean {d.mean():7.0f}   p10 {np.percentile(d, 10):7.0f}   p90 {np.percentile(d, 90):7.0f}   ({100 * d.mean() / tot.mean():4.1f} %)")
