#!/usr/bin/env python3
"""Generate the golden fixtures by running the REFERENCE itself (build container only).

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Imports the reference from /root/reference (read-only), loads the repo's
deterministic synthetic weights into the reference modules, runs them on the
deterministic inputs of tests/cases.py and stores only the reference OUTPUTS
(plus the state_dict layout and weight checksums) under tests/golden/.  Nothing
from the reference's source is written out.  The GPU box never runs this.
"""
import json
import os
import sys
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import cases  # noqa: E402
from cases import sw  # noqa: E402
import models as ref_models  # noqa: E402  (reference)
import modules as ref_modules  # noqa: E402  (reference)
import transforms as ref_transforms  # noqa: E402  (reference)

torch.set_grad_enabled(False)
torch.manual_seed(0)


def load_synth(module, seed, gain=None):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = sw.fill_state_dict(shapes, seed=seed, gain_override=gain)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return module.eval()


ONLY = sys.argv[1:]      # optional name filters: only fixtures whose name contains one of them are (re)written


def save(name, **arrs):
    if ONLY and not any(f in name for f in ONLY):
        return
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.0f} KB  " + " ".join(f"{k}{tuple(np.shape(v))}" for k, v in arrs.items()))


def T(a):
    return torch.from_numpy(np.asarray(a))


# ---------------------------------------------------------------- full model
net = ref_models.SynthesizerTrn(513, 32, n_speakers=109, **cases.IITP_MODEL)
layout = [[k, list(v.shape)] for k, v in net.state_dict().items()]
json.dump(layout, open(os.path.join(HERE, "state_dict_layout.json"), "w"))
shapes = {k: tuple(s) for k, s in layout}
sd_np = sw.fill_state_dict(shapes, seed=cases.WEIGHT_SEED)
json.dump({k: sw.checksum(v) for k, v in sd_np.items()}, open(os.path.join(HERE, "weights_checksums.json"), "w"))
net.load_state_dict({k: T(v) for k, v in sd_np.items()})
net.eval()

for name, c in cases.INFER_CASES.items():
    mel, ln, eps = cases.infer_inputs(name)
    # the reference draws eps with randn_like (models.py:336); substitute ours for that one call
    orig = torch.randn_like
    torch.randn_like = lambda t, *a, **k: T(eps).to(t.dtype)
    try:
        o, mask, (z, z_p, m_p, logs_p) = net.infer(T(mel), T(ln), noise_scale=c["noise_scale"], max_len=c["max_len"])
    finally:
        torch.randn_like = orig
    print(name, "o rms", float(o.pow(2).mean().sqrt()), "max", float(o.abs().max()), "z rms", float(z.pow(2).mean().sqrt()))
    save("infer_" + name, o=o.numpy(), mask=mask.numpy(), z=z.numpy(), z_p=z_p.numpy(), m_p=m_p.numpy(),
         logs_p=logs_p.numpy())

# full-size runs of the reference (the shapes that select the throughput kernels): digests only, see cases.REF_LARGE_CASES
for name, c in cases.REF_LARGE_CASES.items():
    if ONLY and not any(f in "ref_" + name for f in ONLY):
        continue
    mel, ln, eps = cases.large_inputs(name)
    orig = torch.randn_like
    torch.randn_like = lambda t, *a, **k: T(eps).to(t.dtype)
    try:
        o, mask, (z, z_p, m_p, logs_p) = net.infer(T(mel), T(ln), noise_scale=c["noise_scale"])
    finally:
        torch.randn_like = orig
    assert bool(mask.all())
    print(name, "o rms", float(o.pow(2).mean().sqrt()), "max", float(o.abs().max()), "z rms", float(z.pow(2).mean().sqrt()))
    save("ref_" + name, **cases.large_digest(c, o.numpy(), z.numpy(), z_p.numpy(), m_p.numpy(), logs_p.numpy()))

# decoder stage activations of the C1 case (stage rms sanity for DESIGN.md; not stored)

# ---------------------------------------------------------------- module-level
for name, c in cases.RESBLOCK1_CASES.items():
    m = load_synth(ref_modules.ResBlock1(c["C"], c["k"], c["d"]), c["seed"], gain=1.0)
    x = cases.rnd(c["seed"], "x", (c["B"], c["C"], c["L"]), 0.5)
    mask = T(cases.lengths_mask(c["mask_lengths"], c["L"])) if "mask_lengths" in c else None
    save(name, y=m(T(x), mask).numpy())

for name, c in cases.RESBLOCK2_CASES.items():
    m = load_synth(ref_modules.ResBlock2(c["C"], c["k"], c["d"]), c["seed"], gain=1.0)
    x = cases.rnd(c["seed"], "x", (c["B"], c["C"], c["L"]), 0.5)
    save(name, y=m(T(x)).numpy())

for name, c in cases.UPS_CASES.items():
    ct = torch.nn.utils.weight_norm(torch.nn.ConvTranspose1d(c["Ci"], c["Co"], c["k"], c["s"], padding=(c["k"] - c["s"]) // 2))
    load_synth(ct, c["seed"], gain=1.0)
    x = cases.rnd(c["seed"], "x", (c["B"], c["Ci"], c["L"]), 0.5)
    save(name, y=ct(torch.nn.functional.leaky_relu(T(x), 0.1)).numpy())

for name, c in cases.WN_CASES.items():
    m = load_synth(ref_modules.WN(c["H"], c["k"], c["dr"], c["n"], gin_channels=c["gin"]), c["seed"])
    x = cases.rnd(c["seed"], "x", (c["B"], c["H"], c["T"]), 1.0)
    mask = T(cases.lengths_mask(c["lengths"], c["T"]))
    g = T(cases.rnd(c["seed"], "g", (c["B"], c["gin"], 1), 1.0)) if c["gin"] else None
    save(name, y=m(T(x) * mask, mask, g=g).numpy())

for name, c in cases.POSTERIOR_CASES.items():
    m = load_synth(ref_models.PosteriorEncoder(c["Cin"], c["Cout"], c["H"], c["k"], c["dr"], c["n"], gin_channels=c["gin"]), c["seed"])
    x = cases.rnd(c["seed"], "x", (c["B"], c["Cin"], c["T"]), 1.0)
    g = T(cases.rnd(c["seed"], "g", (c["B"], c["gin"], 1), 1.0)) if c["gin"] else None
    lengths = torch.tensor(c["lengths"], dtype=torch.int64)
    torch.manual_seed(c["seed"])
    z, mq, logs, mask = m(T(x), lengths, g=g)
    torch.manual_seed(c["seed"])
    eps = torch.randn_like(mq)          # the draw forward() just took (its only RNG call, models.py:111)
    assert torch.equal(z, (mq + eps * torch.exp(logs)) * mask)
    save(name, z=z.numpy(), m=mq.numpy(), logs=logs.numpy(), mask=mask.numpy(), eps=eps.numpy())

for name, c in cases.COUPLING_CASES.items():
    m = ref_modules.ResidualCouplingLayer(c["C"], c["H"], c["k"], c["dr"], c["n"], gin_channels=c["gin"],
                                          mean_only=c["mean_only"])
    load_synth(m, c["seed"])
    x = cases.rnd(c["seed"], "x", (c["B"], c["C"], c["T"]), 1.0)
    mask = T(cases.lengths_mask(c["lengths"], c["T"]))
    g = T(cases.rnd(c["seed"], "g", (c["B"], c["gin"], 1), 1.0)) if c["gin"] else None
    r = m(T(x), mask, g=g, reverse=c["reverse"])
    if c["reverse"]:
        save(name, y=r.numpy())
    else:
        save(name, y=r[0].numpy(), logdet=r[1].numpy())

for name, c in cases.FLOWBLOCK_CASES.items():
    m = load_synth(ref_models.ResidualCouplingBlock(192, 192, 5, 1, c["n"], gin_channels=0), c["seed"])
    x = cases.rnd(c["seed"], "x", (c["B"], 192, c["T"]), 1.0)
    mask = T(cases.lengths_mask(c["lengths"], c["T"]))
    save(name, y=m(T(x), mask, reverse=c["reverse"]).numpy())

for name, c in cases.GENERATOR_CASES.items():
    m = ref_models.Generator(c["initial_channel"], c["resblock"], c["rks"], c["rds"], c["ur"], c["uic"], c["uks"],
                             gin_channels=c["gin"])
    load_synth(m, c["seed"], gain=1.0)
    x = cases.rnd(c["seed"], "x", (c["B"], c["initial_channel"], c["T"]), 1.0)
    g = T(cases.rnd(c["seed"], "g", (c["B"], c["gin"], 1), 1.0)) if c["gin"] else None
    save(name, y=m(T(x), g=g).numpy())

for name, c in cases.DDS_CASES.items():
    m = load_synth(ref_modules.DDSConv(c["C"], c["k"], c["n"]), c["seed"], gain=1.0)
    x = cases.rnd(c["seed"], "x", (c["B"], c["C"], c["T"]), 1.0)
    mask = T(cases.lengths_mask(c["lengths"], c["T"]))
    g = T(cases.rnd(c["seed"], "g", (c["B"], c["C"], c["T"]), 0.5)) if c["with_g"] else None
    save(name, y=m(T(x), mask, g=g).numpy())

for name, c in cases.CONVFLOW_CASES.items():
    m = load_synth(ref_modules.ConvFlow(c["Cin"], c["F"], c["k"], c["n"]), c["seed"], gain=2.0)
    x = cases.rnd(c["seed"], "x", (c["B"], c["Cin"], c["T"]), 2.5)
    mask = T(cases.lengths_mask(c["lengths"], c["T"]))
    r = m(T(x), mask, reverse=c["reverse"])
    if c["reverse"]:
        save(name, y=r.numpy())
    else:
        save(name, y=r[0].numpy(), logdet=r[1].numpy())

for name, c in cases.SPLINE_CASES.items():
    x, uw, uh, ud = cases.spline_inputs(name)
    y, lad = ref_transforms.piecewise_rational_quadratic_transform(T(x), T(uw), T(uh), T(ud), inverse=c["inverse"],
                                                                   tails="linear", tail_bound=5.0)
    save(name, y=y.numpy(), logabsdet=lad.numpy())

for name, c in cases.CONVFLOW_G_CASES.items():
    m = load_synth(ref_modules.ConvFlow(c["Cin"], c["F"], c["k"], c["n"]), c["seed"], gain=2.0)
    x = cases.rnd(c["seed"], "x", (c["B"], c["Cin"], c["T"]), 2.5)
    mask = T(cases.lengths_mask(c["lengths"], c["T"]))
    g = T(cases.rnd(c["seed"], "g", (c["B"], c["F"], c["gT"]), 0.7))
    r = m(T(x), mask, g=g, reverse=c["reverse"])
    if c["reverse"]:
        save(name, y=r.numpy())
    else:
        save(name, y=r[0].numpy(), logdet=r[1].numpy())

for name, c in cases.LAYERNORM_CASES.items():
    m = ref_modules.LayerNorm(c["C"])
    m.gamma.data.copy_(T(1.0 + cases.rnd(c["seed"], "gamma", (c["C"],), 0.3)))
    m.beta.data.copy_(T(cases.rnd(c["seed"], "beta", (c["C"],), 0.2)))
    x = cases.rnd(c["seed"], "x", c["shape"], 1.5)
    save(name, y=m(T(x)).numpy())

for name, c in cases.MELENC_CASES.items():
    m = load_synth(ref_models.MelEncoder(c["Cout"], c["H"], 768, c["n"], c["k"], c["dr"], c["gin"]), c["seed"])
    x = cases.rnd(c["seed"], "x", (c["B"], 80, c["T"]), 1.0)
    xo, mm, logs, mask = m(T(x), torch.tensor(c["lengths"], dtype=torch.int64))
    save(name, x=xo.numpy(), m=mm.numpy(), logs=logs.numpy(), mask=mask.numpy())

for name, c in cases.SPLINE_EXTRA_CASES.items():
    x, uw, uh, ud = cases.spline_extra_inputs(name)
    kw = dict(min_bin_width=c["mins"][0], min_bin_height=c["mins"][1], min_derivative=c["mins"][2])
    if c["tails"] is None:
        if c["inverse"]:      # inverse inputs must lie in [0,1] too: feed the forward outputs' range
            pass
        y, lad = ref_transforms.piecewise_rational_quadratic_transform(T(x), T(uw), T(uh), T(ud), inverse=c["inverse"], **kw)
    else:
        y, lad = ref_transforms.piecewise_rational_quadratic_transform(T(x), T(uw), T(uh), T(ud), inverse=c["inverse"],
                                                                       tails="linear", tail_bound=5.0, **kw)
    save(name, y=y.numpy(), logabsdet=lad.numpy())

# ---------------------------------------------------------------- a checkpoint written by the reference itself
if not ONLY or any(f in "G_7.pth" for f in ONLY):
    import utils as ref_utils  # noqa: E402  (reference)
    rc = cases.REF_CHECKPOINT
    c = cases.GENERATOR_CASES[rc["case"]]
    m = ref_models.Generator(c["initial_channel"], c["resblock"], c["rks"], c["rds"], c["ur"], c["uic"], c["uks"], gin_channels=c["gin"])
    load_synth(m, c["seed"], gain=1.0)
    opt = torch.optim.AdamW(m.parameters(), rc["learning_rate"])
    import logging
    logging.disable(logging.CRITICAL)
    ref_utils.save_checkpoint(m, opt, rc["learning_rate"], rc["iteration"], os.path.join(HERE, rc["file"]))
    logging.disable(logging.NOTSET)
    print(rc["file"], os.path.getsize(os.path.join(HERE, rc["file"])) // 1024, "KB")

print("done")
