#!/usr/bin/env python3
"""Two (or N) processes run the persistent WN stack launch on ONE GPU at the same time; each compares every output with its first one.
    python tools/wn_stack_shared_gpu.py [nproc=2] [iters=150] [B=16] [T=512]      (1 200: the short-input launch, wn_mesh.hip)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = f"""
import sys; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {ROOT!r} + '/tests')
import torch
from cases import sw
from smart_vocoder_amd import modules
m = modules.WN(192, 5, 1, 16)
m.load_state_dict({{n: torch.from_numpy(v) for n, v in sw.fill_state_dict({{n: tuple(p.shape) for n, p in m.state_dict().items()}}, 7, 0.5).items()}})
m = m.cuda().eval()
g = torch.Generator().manual_seed(int(sys.argv[1]))
# consecutive launches work on DIFFERENT inputs: a stale halo word (the slot's content from the launch before) must not pass for the right one
B, T = int(sys.argv[3]), int(sys.argv[4])
xs = [(torch.randn(B, 192, T, generator=g) * 0.5).cuda() for _ in range(3)]; mask = torch.ones(B, 1, T, device='cuda')
import os
os.environ_backup = os.environ.get("SVOC_WN_STACK")
refs = []
from smart_vocoder_amd import _native
for x in xs:
    refs.append(m(x, mask).clone()); torch.cuda.synchronize()
import time
t0 = time.time(); bad = 0; worst = 0.0; loud = 0
for it in range(int(sys.argv[2])):
    x, ref = xs[it % 3], refs[it % 3]
    try:
        y = m(x, mask)
    except RuntimeError as e:      # the call AFTER a launch that gave up a wait fails (svoc_check_async_error)
        loud += 1; print('rank', sys.argv[1], 'call', it, 'failed:', str(e)[:150], flush=True); continue
    if not torch.equal(y, ref): bad += 1; worst = max(worst, float((y - ref).abs().max()))
torch.cuda.synchronize()
try:
    _native.check_async_error()
except RuntimeError as e:
    loud += 1
print('rank', sys.argv[1], 'mismatching outputs', bad, 'of', sys.argv[2], 'worst abs diff', worst, 'reported failures', loud, 'seconds', round(time.time() - t0, 2), flush=True)
"""
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = sys.argv[2] if len(sys.argv) > 2 else "150"
shape = [sys.argv[3] if len(sys.argv) > 3 else "16", sys.argv[4] if len(sys.argv) > 4 else "512"]
ps = [subprocess.Popen([sys.executable, "-c", CHILD, str(r), iters] + shape, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for r in range(n)]
for p in ps:
    o, _ = p.communicate(timeout=900)
    print(o.strip())
