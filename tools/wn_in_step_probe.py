#!/usr/bin/env python3
"""Why does a WN layer of the persistent stack launch take 49.9 us inside the 16 x 512 step and 42.9 us looped alone?
Times the stack launch (HIP events around each call) in four situations: one stack looped; the model's five stacks (one of 16 layers, four of 8:
99 MB of weight images - more than the L2s hold, less than the Infinity Cache) in turn; the same with 1.2 GB of unrelated traffic between
the calls (the decoder's 40 GB per step evict the weight images from the Infinity Cache); and with an idle gap instead.  Round 6."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import torch
from cases import sw
from smart_vocoder_amd import modules

def wn(nl, seed):
    m = modules.WN(192, 5, 1, nl)
    m.load_state_dict({n: torch.from_numpy(v) for n, v in sw.fill_state_dict({n: tuple(p.shape) for n, p in m.state_dict().items()}, seed, 0.5).items()})
    return m.cuda().eval()

B, T = 16, 512
stacks = [wn(16, 7)] + [wn(8, 8 + i) for i in range(4)]
x = torch.randn(B, 192, T, device="cuda") * 0.5; mask = torch.ones(B, 1, T, device="cuda")
junk = torch.zeros(300 * 1024 * 1024 // 4, device="cuda")

def run(name, pick, between, reps=60):
    per = []
    for i in range(reps + 10):
        m = pick(i)
        between()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); m(x, mask); e1.record(); torch.cuda.synchronize()
        if i >= 10: per.append(e0.elapsed_time(e1) * 1e3 / len(m.in_layers))
    per.sort()
    print(f"{name:72s} us per layer: median {per[len(per) // 2]:6.2f}  p10 {per[len(per) // 10]:6.2f}  p90 {per[9 * len(per) // 10]:6.2f}", flush=True)

def traffic():
    junk.add_(1.0); junk.mul_(0.5)                        # 4 x 300 MB through the memory system
def idle():
    torch.cuda.synchronize(); time.sleep(0.003)
def back_to_back(m, n=50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5): m(x, mask)
    e0.record()
    for _ in range(n): m(x, mask)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n / len(m.in_layers)

print(f"16-layer stack, 50 calls back to back (no synchronisation between): {back_to_back(stacks[0]):.2f} us per layer")
print(f" 8-layer stack, 50 calls back to back:                              {back_to_back(stacks[1]):.2f} us per layer")
run("one 16-layer stack, a synchronisation between the calls", lambda i: stacks[0], lambda: None)
run("the five stacks in turn", lambda i: stacks[i % 5], lambda: None)
run("the five stacks in turn, 1.2 GB of other traffic before each call", lambda i: stacks[i % 5], traffic)
run("one 16-layer stack, 1.2 GB of other traffic before each call", lambda i: stacks[0], traffic)
run("the five stacks in turn, 3 ms idle before each call", lambda i: stacks[i % 5], idle)
