"""Two model handles, each on a stream of its own, in ONE process (round 6): `infer` is enqueued on both streams before anything is synchronised, so the
persistent WN launches of the two compete for the CUs.  Counts silent mismatches against the same calls made alone (must be 0) and failures reported through
`svoc_check_async_error`.   python tools/two_streams_one_process.py [B T iters [B T iters ...]]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from cases import sw
from smart_vocoder_amd import models, _native as N
def mk():
    n = models.SynthesizerTrn(513, 32, n_speakers=109, **cases.IITP_MODEL)
    n.load_state_dict({k: torch.from_numpy(v) for k, v in cases.full_model_weights().items()}, strict=False)
    return n.cuda().eval()
nets = [mk(), mk()]
args = [int(v) for v in sys.argv[1:]] or [16, 512, 30]
if len(args) == 2: args.append(30)
st = [torch.cuda.Stream(), torch.cuda.Stream()]
for B, T, iters in zip(args[0::3], args[1::3], args[2::3]):
    ins = [(torch.from_numpy(sw.synthetic_mel(5 + i, B, T)).cuda(), torch.from_numpy(sw.synthetic_eps(5 + i, B, T)).cuda()) for i in range(2)]
    ln = torch.full((B,), T, dtype=torch.int64).cuda()
    refs = [nets[i].infer(ins[i][0], ln, noise_scale=0.667, eps=ins[i][1])[0].clone() for i in range(2)]
    torch.cuda.synchronize()
    bad = 0; t0 = time.time(); errs = 0
    for it in range(iters):
        outs = [None, None]
        try:
            for i in range(2):
                with torch.cuda.stream(st[i]):
                    outs[i] = nets[i].infer(ins[i][0], ln, noise_scale=0.667, eps=ins[i][1])[0]
            torch.cuda.synchronize()
            N.check_async_error()
        except RuntimeError as e:
            errs += 1; print("reported:", str(e)[:100]); N.debug_persist_control(reenable=True); continue
        for i in range(2):
            if not torch.equal(outs[i], refs[i]):
                bad += 1; print("MISMATCH it", it, i, "finite", bool(torch.isfinite(outs[i]).all()))
    print(f"B={B} T={T}: silent mismatches {bad}, reported failures {errs}, {time.time()-t0:.1f} s, persist state {N.persist_state()}", flush=True)
