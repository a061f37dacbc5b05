#!/usr/bin/env python3
"""Board power and clocks (rocm-smi) while a workload loops for a few seconds: is the chip at its power cap under the Winograd kernels?
    python tools/power_watch.py            (workloads: the clock/power probe's streams, the C=128 k=11 Winograd convolution, the 16x512 step)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def smi():
    out = {}
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=20)
        j = json.loads(r.stdout)
        c = j.get("card0", next(iter(j.values())))
        for k, v in c.items():
            kl = k.lower()
            if "power" in kl or "sclk" in kl or "mclk" in kl or "fclk" in kl or ("temperature" in kl and ("junction" in kl or "hotspot" in kl or "edge" in kl)):
                out[k] = v
    except Exception as e:   # noqa: BLE001
        out["error"] = repr(e)[:100]
    return out


WORK = {
    "idle": None,
    "probe_mode6_streams (MFMA + LDS + L2 weights + VALU producers, 2.38 GHz inside)": ["bash", "-c", f"for i in 1 2 3 4; do ITERS=400000 ONLY=6 {ROOT}/tools/clock_power_probe > /dev/null; done"],
    "winograd C=128 k=11 d=1 (stamps: ~2.05 GHz inside)": [sys.executable, "-c", f"""
import sys, ctypes, time; sys.path.insert(0, {ROOT!r})
import torch
from smart_vocoder_amd import _native as N
lib = N.lib(); B, C, L, k = 16, 128, 32768, 11
x = torch.randn(B, C, L, device='cuda') * 0.5; y = torch.empty_like(x)
v = torch.randn(C, C, k, device='cuda') / (C * k) ** 0.5; g = torch.rand(C, 1, 1, device='cuda') + 0.5; b = torch.randn(C, device='cuda') * 0.1
t0 = time.time()
while time.time() - t0 < 7:
    for _ in range(50):
        N.check(lib.svoc_conv1d_winograd(N.stream_ptr(), N.ptr(x), N.ptr(v), N.ptr(g), N.ptr(b), N.ptr(x), N.ptr(y), B, C, C, L, k, 1, ctypes.c_float(0.1)))
    torch.cuda.synchronize()
"""],
    "16x512 infer step (bench workload, back to back)": [sys.executable, "-c", f"""
import sys, time; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {ROOT!r} + '/tests')
import torch, cases
from cases import sw
from smart_vocoder_amd import models
net = models.SynthesizerTrn(513, 32, n_speakers=109, **cases.IITP_MODEL)
net.load_state_dict({{k: torch.from_numpy(v) for k, v in cases.full_model_weights().items()}}, strict=False)
net = net.cuda().eval()
mel = torch.from_numpy(sw.synthetic_mel(1001, 16, 512)).cuda(); eps = torch.from_numpy(sw.synthetic_eps(1001, 16, 512)).cuda()
ln = torch.full((16,), 512, dtype=torch.int64).cuda()
t0 = time.time()
with torch.no_grad():
    while time.time() - t0 < 12:
        for _ in range(20): net.infer(mel, ln, noise_scale=0.667, eps=eps)
        torch.cuda.synchronize()
"""],
}
for name, cmd in WORK.items():
    p = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) if cmd else None
    time.sleep((8.0 if "infer" in name else 2.5) if cmd else 0.2)          # process start + warm-up
    print("==", name, flush=True)
    for _ in range(5):
        if p is not None and p.poll() is not None:
            print("   (workload ended)"); break
        print("  ", json.dumps(smi()), flush=True)
        time.sleep(0.3)
    if p is not None:
        try:
            p.wait(timeout=60)
        except subprocess.TimeoutExpired:
            p.kill()
