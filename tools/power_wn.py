#!/usr/bin/env python3
"""Board power while a 16-layer WN stack (16 x 512) loops back to back: persistent stack launch, and one launch per layer (SVOC_WN_STACK=0)."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = f"""
import sys, time; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {ROOT!r} + '/tests')
import torch
from cases import sw
from smart_vocoder_amd import modules
m = modules.WN(192, 5, 1, 16)
m.load_state_dict({{n: torch.from_numpy(v) for n, v in sw.fill_state_dict({{n: tuple(p.shape) for n, p in m.state_dict().items()}}, 7, 0.5).items()}})
m = m.cuda().eval()
x = torch.randn(16, 192, 512, device='cuda') * 0.5; mask = torch.ones(16, 1, 512, device='cuda')
t0 = time.time(); us = []
while time.time() - t0 < 6:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): m(x, mask)
    e1.record(); torch.cuda.synchronize(); us.append(e0.elapsed_time(e1) * 1e3 / 800)
print('CHILD', sorted(us)[len(us) // 2])
"""
def power():
    r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=20)
    c = next(iter(json.loads(r.stdout).values()))
    return {k: v for k, v in c.items() if "ower" in k or "sclk clock speed" in k}
for name, env in (("persistent stack", {}), ("one launch per layer", {"SVOC_WN_STACK": "0"})):
    p = subprocess.Popen([sys.executable, "-c", CHILD], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    time.sleep(3.5)
    w = [power() for _ in range(3)]
    o, _ = p.communicate(timeout=60)
    print(name, [l for l in o.splitlines() if l.startswith("CHILD")], "us per layer;", w, flush=True)
