"""Phase stamps of the persistent WN stack launch (csrc/wn_stack.hip): thread 0 (wave 0: K half 0, row pair 0) of every workgroup, layer n_layers / 2, 100 MHz wall clock.
    python tools/wn_stack_timeline.py [B=16] [T=512] [layers=16]"""
import os, sys
import numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from cases import sw
from smart_vocoder_amd import modules, _native as N
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
NL = int(sys.argv[3]) if len(sys.argv) > 3 else 16
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
for _ in range(20): m(x, mask)           # (the stamps of the LAST of a burst survive)
torch.cuda.synchronize(); N.check(lib.svoc_debug_set_stamp_buffer(None))
D = buf.cpu().numpy(); D = D[D[:, 11] != 0]
if not len(D): print("(no stamps: the stack launch was not taken at this shape)"); sys.exit(0)
names = ["wait for the left neighbour's layer counter (thread 0)", "barrier, edge loads (768 device-scope words) -> the tile, barrier", "input transform -> planes, barrier",
         "phase A: this wave's 576 MFMAs (16x16x4) of the F(2,5) in_layer", "output transform of the partial sums, hand-over writes, barrier (= the SIMD's other two waves' streams)",
         "gate (tanh * sigmoid) -> acts tile", "barrier (acts tile complete)", "phase B: this wave's res_skip MFMAs (32x32x2)", "exchange writes, barrier",
         "x update in the tile, edge stores, vmcnt(0)", "barrier (publish)"]
tot = (D[:, 11] - D[:, 0]) * 10.0
print(f"{len(D)} workgroups; layer {NL // 2}, thread 0, ns: stamped span {tot.mean():.0f} = {tot.mean() / 1e3:.1f} us")
for i, n in enumerate(names):
    d = (D[:, i + 1] - D[:, i]) * 10.0
    print(f"   {n:120s} mean {d.mean():7.0f}   p10 {np.percentile(d, 10):7.0f}   p90 {np.percentile(d, 90):7.0f}   ({100 * d.mean() / tot.mean():4.1f} %)")
