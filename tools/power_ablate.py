#!/usr/bin/env python3
"""Steady-state power, launch time and in-kernel clock of the stamped C=128 k=11 Winograd kernel with work removed (SVOC_DBG_ABL), each
variant looped back to back for a few seconds (a single launch after an idle gap says nothing about the power cap):
    python tools/power_ablate.py [abl values ...]        (child mode: python tools/power_ablate.py --child K D C)"""
import ctypes, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(k, d, C, stamped):
    """ResBlock1(C, k, dilations (d, d, d)) forward = six launches of the single-convolution kernel (three with dilation d, three undilated), enqueued
    back to back (the module packs its weights once; svoc_conv1d_winograd would pack and synchronise at every call)."""
    import numpy as np, torch
    from smart_vocoder_amd import _native as N, modules
    lib = N.lib(); B, L = 16, (32768 if C >= 128 else 65536 * 128 // C // 2)
    m = modules.ResBlock1(C, k, (d, d, d))
    g = torch.Generator().manual_seed(3)
    for n_, p_ in m.named_parameters():
        p_.data.copy_(torch.randn(p_.shape, generator=g) * (0.3 if "weight_g" in n_ else (1.0 / (C * k) ** 0.5 if "weight_v" in n_ else 0.05)) + (0.8 if "weight_g" in n_ else 0.0))
    m = m.cuda().eval()
    x = torch.randn(B, C, L, device="cuda") * 0.3
    buf = torch.zeros(1 << 16, 16, dtype=torch.long, device="cuda")
    for _ in range(3): m(x)
    torch.cuda.synchronize()
    if stamped: N.check(lib.svoc_debug_set_stamp_buffer(N.ptr(buf)))
    t0 = time.time(); us = []
    with torch.no_grad():
        while time.time() - t0 < 5.0:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): m(x)
            e1.record(); torch.cuda.synchronize()
            us.append(e0.elapsed_time(e1) * 1e3 / 120.0)
    out = {"us_per_launch": float(np.median(us[len(us) // 2:]))}
    if stamped:
        D = buf.cpu().numpy(); D = D[D[:, 5] == 4]
        wall = (D[:, 11] - D[:, 10]) / 100.0
        out["clock_mhz"] = float(np.median(D[:, 1] / wall)); out["cycles_per_tile"] = float(np.mean(D[:, 1] / D[:, 0])); out["stream_cycles_per_tile"] = float(np.mean(D[:, 3] / D[:, 0]))
    print("CHILD " + json.dumps(out), flush=True)


def power():
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=20)
        c = next(iter(json.loads(r.stdout).values()))
        return float(next(v for k, v in c.items() if "ower" in k))
    except Exception:   # noqa: BLE001
        return float("nan")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5] == "1")
        sys.exit(0)
    K, Dl, C = int(os.environ.get("PK", "11")), int(os.environ.get("PD", "1")), int(os.environ.get("PC", "128"))
    variants = [("production build", None)] + [(f"stamped, SVOC_DBG_ABL={a}", a) for a in (sys.argv[1:] or ["0", "1", "2", "4", "8", "5", "13", "15"])]
    print(f"# ResBlock1({C}, {K}, dilations {Dl}) forward looped back to back, 16 x {32768 if C >= 128 else 65536 * 128 // C // 2} columns, time per convolution launch; ABL bits: 1 no global loads, 2 no publish/transform, 4 no epilogue, 8 weights from the L1")
    for name, a in variants:
        env = dict(os.environ)
        if a is not None: env["SVOC_DBG_ABL"] = str(a)
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(K), str(Dl), str(C), "0" if a is None else "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        time.sleep(3.2)
        w = [power() for _ in range(3)]
        o, _ = p.communicate(timeout=120)
        line = [l for l in o.splitlines() if l.startswith("CHILD ")]
        j = json.loads(line[-1][6:]) if line else {}
        print(f"{name:32s} power {min(w):6.0f} .. {max(w):6.0f} W   {j.get('us_per_launch', float('nan')):8.1f} us / launch   clock {j.get('clock_mhz', float('nan')):6.0f} MHz   "
              f"cycles / tile {j.get('cycles_per_tile', float('nan')):8.0f} (streams {j.get('stream_cycles_per_tile', float('nan')):8.0f})", flush=True)
