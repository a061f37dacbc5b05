# A/B of the C=32 stage: grouped F(4,3) (NRT=1) vs fused direct kernels; run on the GPU box
import os, sys, time, subprocess, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, numpy as np
import cases
from cases import sw
from smart_vocoder_amd import models, _native as N
def run(B=16, T=512, steps=10):
    net = models.SynthesizerTrn(513, 32, n_speakers=109, **cases.IITP_MODEL)
    sd = cases.full_model_weights(skip_enc_q=True)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    net = net.cuda().eval()
    mel = torch.from_numpy(sw.synthetic_mel(1001, B, T)).cuda(); eps = torch.from_numpy(sw.synthetic_eps(1001, B, T)).cuda()
    ln = torch.full((B,), T, dtype=torch.int64).cuda()
    with torch.no_grad():
        for _ in range(3): o = net.infer(mel, ln, noise_scale=0.667, eps=eps)[0]
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(steps): o = net.infer(mel, ln, noise_scale=0.667, eps=eps)[0]
        torch.cuda.synchronize(); dt = (time.time() - t0) / steps * 1e3
    return dt, o
if __name__ == "__main__":
    dt, o = run()
    print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("SVOC_")}, "ms": dt, "rms": float(o.pow(2).mean().sqrt())}))
    torch.save(o.cpu(), os.environ.get("OUT", "/tmp/o.pt"))
