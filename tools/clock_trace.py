#!/usr/bin/env python3
"""Shader clock while the hot path runs: a one-wave monitor kernel (tools/clock_monitor.hip) samples the shader-clock
counter against the 100 MHz wall clock on its own stream while `infer` runs on the default stream.
    python tools/clock_trace.py [B] [T] [steps]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from cases import sw
from smart_vocoder_amd import models

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
mon = ctypes.CDLL(os.path.join(ROOT, "tools", "libclockmon.so"))
net = models.SynthesizerTrn(513, 32, n_speakers=109, **cases.IITP_MODEL)
net.load_state_dict({k: torch.from_numpy(v) for k, v in cases.full_model_weights().items()}, strict=False)
net = net.cuda().eval()
mel = torch.from_numpy(sw.synthetic_mel(1001, B, T)).cuda(); eps = torch.from_numpy(sw.synthetic_eps(1001, B, T)).cuda()
lengths = torch.full((B,), T, dtype=torch.int64).cuda()
with torch.no_grad():
    net.infer(mel, lengths, noise_scale=0.667, eps=eps)
torch.cuda.synchronize()

interval = 5000                                   # 50 us per sample
nsamples = 40000                                  # up to 2 s
buf = torch.zeros(nsamples * 3, dtype=torch.long, device="cuda")
marks = torch.zeros(2 * steps + 2, dtype=torch.long, device="cuda")
stop = torch.zeros(1, dtype=torch.int32, device="cuda")
work = torch.cuda.Stream()
torch.cuda.synchronize()
# warm the mark kernel (code-object load) before the monitor starts
mon.clockmon_mark(ctypes.c_void_p(work.cuda_stream), ctypes.c_void_p(marks.data_ptr() + 8 * (2 * steps)))
torch.cuda.synchronize()
main = work
mon.clockmon_launch(None, ctypes.c_void_p(buf.data_ptr()), nsamples, interval, ctypes.c_void_p(stop.data_ptr()))
import time
time.sleep(0.05)                                  # idle baseline before the workload
with torch.no_grad(), torch.cuda.stream(work):
    for s in range(steps):
        mon.clockmon_mark(ctypes.c_void_p(main.cuda_stream), ctypes.c_void_p(marks.data_ptr() + 16 * s))
        net.infer(mel, lengths, noise_scale=0.667, eps=eps)
        mon.clockmon_mark(ctypes.c_void_p(main.cuda_stream), ctypes.c_void_p(marks.data_ptr() + 16 * s + 8))
main.synchronize()
time.sleep(0.02)
stop.fill_(1)
torch.cuda.synchronize()
d = buf.cpu().numpy().reshape(-1, 3)
d = d[d[:, 1] > 0]
mk = marks.cpu().numpy()
mhz = 100.0 * d[:, 2] / d[:, 1]
t0 = d[0, 0]
print(f"# {len(d)} samples of {interval / 100:.0f} us; B={B} T={T}")
idle = mhz[d[:, 0] < mk[0] - 1000]
print(f"idle before the workload: median {np.median(idle):.0f} MHz" if len(idle) else "no idle samples")
for s in range(steps):
    a, b = mk[2 * s], mk[2 * s + 1]
    sel = (d[:, 0] >= a) & (d[:, 0] + d[:, 1] <= b)
    m = mhz[sel]
    if not len(m):
        print(f"step {s}: {(b - a) / 100e3:.2f} ms, no samples"); continue
    print(f"step {s}: {(b - a) / 100e3:.2f} ms, {sel.sum()} samples, shader clock min {m.min():.0f} p10 {np.percentile(m, 10):.0f} "
          f"median {np.median(m):.0f} mean {m.mean():.0f} p90 {np.percentile(m, 90):.0f} max {m.max():.0f} MHz")
a, b = mk[2 * (steps - 1)], mk[2 * steps - 1]
sel = (d[:, 0] >= a) & (d[:, 0] <= b)
tt = (d[sel, 0] - a) / 100e3
mm = mhz[sel]
print("# last step, 1 ms bins (ms: mean MHz)")
for k in range(int(tt.max()) + 1):
    q = mm[(tt >= k) & (tt < k + 1)]
    if len(q):
        print(f"  {k:3d}: {q.mean():6.0f}")
