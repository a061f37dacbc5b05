import os, sys, time, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import cases
from cases import sw
from smart_vocoder_amd import models
net = models.SynthesizerTrn(513, 32, n_speakers=109, **cases.IITP_MODEL)
net.load_state_dict({k: torch.from_numpy(v) for k, v in cases.full_model_weights().items()}, strict=False)
net = net.cuda().eval()
shapes = [(16, 512), (1, 200), (4, 512), (2, 333), (1, 512), (16, 512), (3, 77), (8, 1024)]
ins = {}
for (B, T) in shapes:
    ins[(B, T)] = (torch.from_numpy(sw.synthetic_mel(1, B, T)).cuda(), torch.full((B,), T, dtype=torch.int64).cuda(), torch.from_numpy(sw.synthetic_eps(1, B, T)).cuda())
ref = {}
free0 = None
t0 = time.time()
for it in range(60):
    for (B, T) in shapes:
        mel, ln, eps = ins[(B, T)]
        o = net.infer(mel, ln, noise_scale=0.667, eps=eps)[0]
        if (B, T) not in ref:
            ref[(B, T)] = o.clone()
        elif it % 10 == 0:
            assert torch.equal(o, ref[(B, T)]), f"non-deterministic at {(B, T)} iteration {it}"
    torch.cuda.synchronize()
    if it == 5:
        free0 = torch.cuda.mem_get_info()[0]
free1 = torch.cuda.mem_get_info()[0]
print(f"soak: 60 rounds x {len(shapes)} shapes in {time.time() - t0:.1f} s, outputs bit-stable, free memory after warm-up {free0 / 2**30:.2f} GiB -> end {free1 / 2**30:.2f} GiB")
