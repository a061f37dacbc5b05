"""Shared definitions of the parity cases: shapes, seeds and input construction.

Used by tests/golden/make_golden.py (build container, imports the reference) and
by the tests (which never touch /root/reference).  Inputs and weights are
regenerated from smart-vocoder_amd/synth_weights.py; only reference OUTPUTS are
committed under tests/golden/.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from smart_vocoder_amd import synth_weights as sw  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
WEIGHT_SEED = 1234

# iitp_base.json "model" section (reference configs/iitp_base.json:35-52)
IITP_MODEL = dict(inter_channels=192, hidden_channels=192, filter_channels=768, n_heads=2, n_layers=6,
                  kernel_size=3, p_dropout=0.1, resblock="1", resblock_kernel_sizes=[3, 7, 11],
                  resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], upsample_rates=[8, 8, 2, 2],
                  upsample_initial_channel=512, upsample_kernel_sizes=[16, 16, 4, 4], n_layers_q=3,
                  use_spectral_norm=False, gin_channels=256)

# full-path cases: name -> (B, T, lengths or None, max_len, noise_scale, data seed)
INFER_CASES = {
    "c1": dict(B=1, T=200, lengths=None, max_len=None, noise_scale=0.667, seed=1001),
    "ragged": dict(B=3, T=64, lengths=[64, 50, 17], max_len=None, noise_scale=0.667, seed=1002),
    "maxlen": dict(B=2, T=48, lengths=[48, 30], max_len=40, noise_scale=1.0, seed=1003),
    "odd": dict(B=2, T=37, lengths=[37, 5], max_len=None, noise_scale=0.0, seed=1004),
}


# Full-size runs of the REFERENCE itself (tests/golden/make_golden.py): the bench workload C2 (16 x 512, seed 1001 - exactly
# bench.py's tensors) and one long utterance (1 x 4096).  Stored: every `stride`-th waveform sample, float64 sum / sum of
# squares of the waveform per mel frame (256 samples), per-utterance float64 sum / sum of squares of z, z_p, m_p, logs_p, and
# z of `z_rows` at every `z_stride`-th frame.  These shapes reach the throughput kernels (grouped / pair / accumulate Winograd
# launches, F(4,2) upsamplers, fused F(2,5) WN layers) that the small full-path cases above never select.
REF_LARGE_CASES = {
    "c2_16x512": dict(B=16, T=512, seed=1001, noise_scale=0.667, stride=64, z_rows=(0, 15), z_stride=4),
    "long_1x4096": dict(B=1, T=4096, seed=1005, noise_scale=0.667, stride=64, z_rows=(0,), z_stride=16),
}


def large_inputs(case):
    c = REF_LARGE_CASES[case]
    mel = sw.synthetic_mel(c["seed"], c["B"], c["T"])
    eps = sw.synthetic_eps(c["seed"], c["B"], c["T"])
    ln = np.full((c["B"],), c["T"], dtype=np.int64)
    return mel, ln, eps


def large_digest(c, o, z, z_p, m_p, logs_p, rows=None):
    """The quantities a REF_LARGE fixture holds, computed from full outputs (numpy, float32 in; sums in float64).
    `rows`: the utterances `o` ... hold (default: all of the case's batch)."""
    o = np.asarray(o)[:, 0, :]
    d = {"o_sub": np.ascontiguousarray(o[:, ::c["stride"]]).astype(np.float32)}
    o64 = o.astype(np.float64).reshape(o.shape[0], -1, 256)
    d["o_frame_sum"] = o64.sum(-1)
    d["o_frame_sumsq"] = (o64 * o64).sum(-1)
    for nm, a in (("z", z), ("z_p", z_p), ("m_p", m_p), ("logs_p", logs_p)):
        a64 = np.asarray(a).astype(np.float64)
        d[nm + "_sum"] = a64.sum((1, 2))
        d[nm + "_sumsq"] = (a64 * a64).sum((1, 2))
    rows = list(range(o.shape[0])) if rows is None else list(rows)
    zr = [rows.index(r) for r in c["z_rows"] if r in rows]
    d["z_rows"] = np.ascontiguousarray(np.asarray(z)[zr][:, :, ::c["z_stride"]]).astype(np.float32)
    return d


def infer_inputs(case):
    c = INFER_CASES[case]
    mel = sw.synthetic_mel(c["seed"], c["B"], c["T"])
    eps = sw.synthetic_eps(c["seed"], c["B"], c["T"])
    ln = np.array(c["lengths"] if c["lengths"] is not None else [c["T"]] * c["B"], dtype=np.int64)
    return mel, ln, eps


def rnd(seed, name, shape, scale=1.0):
    return (sw.normal(seed, name, shape) * scale).astype(np.float32)


def lengths_mask(lengths, T):
    return (np.arange(T)[None, :] < np.asarray(lengths)[:, None]).astype(np.float32)[:, None, :]


# module-level cases ---------------------------------------------------------
# ResBlock1: every (C, k) of the iitp_base decoder (reference models.py:129-133) at short L.
RESBLOCK1_CASES = {f"rb1_c{C}_k{k}": dict(C=C, k=k, d=(1, 3, 5), B=2, L=L, seed=2000 + C + k)
                   for C, L in ((256, 40), (128, 72), (64, 136), (32, 264)) for k in (3, 7, 11)}
RESBLOCK1_CASES["rb1_c32_k3_mask"] = dict(C=32, k=3, d=(1, 3, 5), B=2, L=100, seed=2999, mask_lengths=[100, 61])
RESBLOCK2_CASES = {
    "rb2_c64_k3": dict(C=64, k=3, d=(1, 3), B=2, L=100, seed=3001),
    "rb2_c32_k7": dict(C=32, k=7, d=(1, 3), B=1, L=77, seed=3002),
}
# ConvTranspose1d geometries of the decoder (reference models.py:123-127), preceded by lrelu(0.1)
UPS_CASES = {
    "up_512_256_k16_s8": dict(Ci=512, Co=256, k=16, s=8, B=2, L=19, seed=4001),
    "up_256_128_k16_s8": dict(Ci=256, Co=128, k=16, s=8, B=2, L=33, seed=4002),
    "up_128_64_k4_s2": dict(Ci=128, Co=64, k=4, s=2, B=2, L=70, seed=4003),
    "up_64_32_k4_s2": dict(Ci=64, Co=32, k=4, s=2, B=1, L=129, seed=4004),
}
WN_CASES = {
    "wn_h192_k5_n3": dict(H=192, k=5, dr=1, n=3, gin=0, B=2, T=50, lengths=[50, 33], seed=5001),
    "wn_h192_k5_n3_g": dict(H=192, k=5, dr=1, n=3, gin=256, B=2, T=50, lengths=[50, 33], seed=5002),
    "wn_h64_k3_dr2_n4_g": dict(H=64, k=3, dr=2, n=4, gin=32, B=2, T=61, lengths=[61, 40], seed=5003),
}
POSTERIOR_CASES = {   # PosteriorEncoder(in, out, hidden, k, dr, n, gin) (models.py:83-112)
    "posterior_small_g": dict(Cin=33, Cout=16, H=32, k=5, dr=1, n=3, gin=8, B=2, T=37, lengths=[37, 20], seed=5601),
    "posterior_513_192": dict(Cin=513, Cout=192, H=192, k=5, dr=1, n=2, gin=0, B=2, T=24, lengths=[24, 9], seed=5602),
}
COUPLING_CASES = {
    "rcl_mean_rev": dict(C=192, H=192, k=5, dr=1, n=2, gin=0, mean_only=True, reverse=True, B=2, T=40,
                         lengths=[40, 23], seed=6001),
    "rcl_mean_fwd": dict(C=192, H=192, k=5, dr=1, n=2, gin=0, mean_only=True, reverse=False, B=2, T=40,
                         lengths=[40, 23], seed=6002),
    "rcl_full_rev_g": dict(C=64, H=96, k=5, dr=1, n=2, gin=32, mean_only=False, reverse=True, B=2, T=36,
                           lengths=[36, 20], seed=6003),
    "rcl_full_fwd_g": dict(C=64, H=96, k=5, dr=1, n=2, gin=32, mean_only=False, reverse=False, B=2, T=36,
                           lengths=[36, 20], seed=6004),
}
FLOWBLOCK_CASES = {
    "flow_rev": dict(B=2, T=30, lengths=[30, 19], n=2, reverse=True, seed=6101),
    "flow_fwd": dict(B=2, T=30, lengths=[30, 19], n=2, reverse=False, seed=6102),
}
GENERATOR_CASES = {
    # reduced Generator with speaker conditioning g (reference models.py:143-144)
    "gen_small_g": dict(initial_channel=48, resblock="1", rks=[3, 7], rds=[[1, 3, 5], [1, 3, 5]], ur=[4, 2],
                        uic=64, uks=[8, 4], gin=24, B=2, T=21, seed=7001),
    "gen_small_rb2": dict(initial_channel=32, resblock="2", rks=[3, 5], rds=[[1, 3], [1, 3]], ur=[2, 2], uic=64,
                          uks=[4, 4], gin=0, B=1, T=33, seed=7002),
}
DDS_CASES = {
    "dds_c192_k3_n3": dict(C=192, k=3, n=3, B=2, T=45, lengths=[45, 30], with_g=True, seed=8001),
    "dds_c64_k5_n2": dict(C=64, k=5, n=2, B=1, T=70, lengths=[70], with_g=False, seed=8002),
}
CONVFLOW_CASES = {
    "cf_rev": dict(Cin=2, F=192, k=3, n=3, B=2, T=40, lengths=[40, 25], reverse=True, seed=9001),
    "cf_fwd": dict(Cin=2, F=192, k=3, n=3, B=2, T=40, lengths=[40, 25], reverse=False, seed=9002),
    "cf_c4_fwd": dict(Cin=4, F=64, k=3, n=2, B=1, T=33, lengths=[33], reverse=False, seed=9003),
}
CONVFLOW_G_CASES = {   # g is handed to DDSConv (reference modules.py:366): [B, filter_channels, 1] or [B, filter_channels, T]
    "cf_g1_fwd": dict(Cin=2, F=64, k=3, n=2, B=2, T=33, lengths=[33, 20], reverse=False, gT=1, seed=9011),
    "cf_gT_rev": dict(Cin=4, F=64, k=3, n=2, B=2, T=29, lengths=[29, 11], reverse=True, gT=29, seed=9012),
}
LAYERNORM_CASES = {    # standalone modules.LayerNorm.forward (reference modules.py:28-32)
    "ln_c192": dict(C=192, shape=(2, 192, 45), seed=9201),
    "ln_c7_4d": dict(C=7, shape=(3, 7, 5, 6), seed=9202),
}
MELENC_CASES = {       # standalone MelEncoder.forward (reference models.py:35-47): (out, hidden, n_layers, k, dr, gin)
    "melenc_small": dict(Cout=16, H=32, n=3, k=5, dr=1, gin=0, B=2, T=41, lengths=[41, 18], seed=9301),
    "melenc_h192": dict(Cout=192, H=192, n=2, k=5, dr=1, gin=8, B=1, T=30, lengths=[30], seed=9302),
}
SPLINE_CASES = {
    "spline_fwd": dict(N=4096, inverse=False, seed=9101),
    "spline_inv": dict(N=4096, inverse=True, seed=9102),
}
# tails=None (domain [0,1], transforms.py:23-25 -> rational_quadratic_spline) and non-default minimums (transforms.py:20-22)
SPLINE_EXTRA_CASES = {
    "spline_unc_fwd": dict(N=2048, inverse=False, tails=None, mins=(1e-3, 1e-3, 1e-3), seed=9111),
    "spline_unc_inv": dict(N=2048, inverse=True, tails=None, mins=(1e-3, 1e-3, 1e-3), seed=9112),
    "spline_min_fwd": dict(N=2048, inverse=False, tails="linear", mins=(1e-2, 5e-3, 2e-2), seed=9113),
    "spline_min_inv": dict(N=2048, inverse=True, tails="linear", mins=(1e-2, 5e-3, 2e-2), seed=9114),
    "spline_unc_min_fwd": dict(N=1024, inverse=False, tails=None, mins=(2e-2, 1e-2, 5e-2), seed=9115),
}
# a checkpoint written by the REFERENCE's utils.save_checkpoint (reference utils.py:46-56) for the gen_small_rb2 Generator
REF_CHECKPOINT = dict(file="G_7.pth", case="gen_small_rb2", iteration=7, learning_rate=2e-4)


def spline_extra_inputs(case):
    c = SPLINE_EXTRA_CASES[case]
    N = c["N"]
    nb = 10
    if c["tails"] is None:
        x = sw.uniform01(c["seed"], "x", N).astype(np.float32)
        x[:4] = np.array([0.0, 1.0, 0.5, 0.999999], dtype=np.float32)
        nd = nb + 1
    else:
        x = rnd(c["seed"], "x", (N,), 3.0)
        x[:6] = np.array([-5.0, 5.0, -7.5, 9.0, 0.0, 4.999999], dtype=np.float32)
        nd = nb - 1
    uw = rnd(c["seed"], "uw", (N, nb), 1.0)
    uh = rnd(c["seed"], "uh", (N, nb), 1.0)
    ud = rnd(c["seed"], "ud", (N, nd), 1.0)
    return x, uw, uh, ud


def spline_inputs(case):
    c = SPLINE_CASES[case]
    N = c["N"]
    x = rnd(c["seed"], "x", (N,), 3.0)
    x[:8] = np.array([-5.0, 5.0, -7.5, 9.0, 0.0, -4.999999, 4.999999, 5.0000005], dtype=np.float32)
    uw = rnd(c["seed"], "uw", (N, 10), 1.0)
    uh = rnd(c["seed"], "uh", (N, 10), 1.0)
    ud = rnd(c["seed"], "ud", (N, 9), 1.0)
    # a block with flat parameters: knots land exactly on multiples of 1.0 in [-5,5]
    uw[8:40] = 0.0
    uh[8:40] = 0.0
    x[8:19] = np.arange(-5, 6, dtype=np.float32)
    return x, uw, uh, ud


# ---------------------------------------------------------------- state-dict shape builders
# (names and shapes as the reference modules register them; verified against
#  tests/golden/state_dict_layout.json and by make_golden.py's load_state_dict)
def _wnconv(d, p, co, ci, k):
    d[p + ".bias"] = (co,)
    d[p + ".weight_g"] = (co, 1, 1)
    d[p + ".weight_v"] = (co, ci, k)


def _conv(d, p, co, ci, k, bias=True):
    d[p + ".weight"] = (co, ci, k)
    if bias:
        d[p + ".bias"] = (co,)


def wn_shapes(H, k, n, gin, prefix=""):
    d = {}
    for i in range(n):
        _wnconv(d, f"{prefix}in_layers.{i}", 2 * H, H, k)
    for i in range(n):
        _wnconv(d, f"{prefix}res_skip_layers.{i}", 2 * H if i < n - 1 else H, H, 1)
    if gin:
        _wnconv(d, f"{prefix}cond_layer", 2 * H * n, gin, 1)
    return d


def resblock1_shapes(C, k, prefix=""):
    d = {}
    for i in range(3):
        _wnconv(d, f"{prefix}convs1.{i}", C, C, k)
    for i in range(3):
        _wnconv(d, f"{prefix}convs2.{i}", C, C, k)
    return d


def resblock2_shapes(C, k, prefix=""):
    d = {}
    for i in range(2):
        _wnconv(d, f"{prefix}convs.{i}", C, C, k)
    return d


def ups_shapes(Ci, Co, k, prefix=""):
    return {prefix + "bias": (Co,), prefix + "weight_g": (Ci, 1, 1), prefix + "weight_v": (Ci, Co, k)}


def posterior_shapes(Cin, Cout, H, k, n, gin, prefix=""):
    d = {}
    _conv(d, prefix + "pre", H, Cin, 1)
    d.update(wn_shapes(H, k, n, gin, prefix + "enc."))
    _conv(d, prefix + "proj", 2 * Cout, H, 1)
    return d


def coupling_shapes(C, H, k, n, gin, mean_only, prefix=""):
    d = {}
    _conv(d, prefix + "pre", H, C // 2, 1)
    d.update(wn_shapes(H, k, n, gin, prefix + "enc."))
    _conv(d, prefix + "post", (C // 2) * (1 if mean_only else 2), H, 1)
    return d


def flowblock_shapes(n, prefix=""):
    d = {}
    for i in range(4):
        d.update(coupling_shapes(192, 192, 5, n, 0, True, f"{prefix}flows.{2 * i}."))
    return d


def generator_shapes(c, prefix=""):
    d = {}
    _conv(d, prefix + "conv_pre", c["uic"], c["initial_channel"], 7)
    for i, (u, k) in enumerate(zip(c["ur"], c["uks"])):
        d.update(ups_shapes(c["uic"] >> i, c["uic"] >> (i + 1), k, f"{prefix}ups.{i}."))
    for i in range(len(c["ur"])):
        ch = c["uic"] >> (i + 1)
        for j, k in enumerate(c["rks"]):
            f = resblock1_shapes if c["resblock"] == "1" else resblock2_shapes
            d.update(f(ch, k, f"{prefix}resblocks.{i * len(c['rks']) + j}."))
    _conv(d, prefix + "conv_post", 1, ch, 7, bias=False)
    if c["gin"]:
        _conv(d, prefix + "cond", c["uic"], c["gin"], 1)
    return d


def dds_shapes(C, k, n, prefix=""):
    d = {}
    for i in range(n):
        _conv(d, f"{prefix}convs_sep.{i}", C, 1, k)
        _conv(d, f"{prefix}convs_1x1.{i}", C, C, 1)
        for nm in ("norms_1", "norms_2"):
            d[f"{prefix}{nm}.{i}.gamma"] = (C,)
            d[f"{prefix}{nm}.{i}.beta"] = (C,)
    return d


def melenc_shapes(Cout, H, k, n, gin, prefix=""):
    d = {}
    _conv(d, prefix + "pre_enc", H, 80, 1)
    d.update(wn_shapes(H, k, n, gin, prefix + "encoder."))
    _conv(d, prefix + "proj", 2 * Cout, H, 1)
    return d


def convflow_shapes(Cin, Fc, k, n, num_bins=10, prefix=""):
    d = {}
    _conv(d, prefix + "pre", Fc, Cin // 2, 1)
    d.update(dds_shapes(Fc, k, n, prefix + "convs."))
    _conv(d, prefix + "proj", (Cin // 2) * (3 * num_bins - 1), Fc, 1)
    return d


def full_model_shapes():
    import json
    return {k: tuple(s) for k, s in json.load(open(os.path.join(GOLDEN_DIR, "state_dict_layout.json")))}


_FULL_SD = {}


def full_model_weights(seed=WEIGHT_SEED, skip_enc_q=True):
    """Synthetic iitp_base weights as {name: float32 ndarray}; enc_q (unused by infer) skipped by default."""
    key = (seed, skip_enc_q)
    if key not in _FULL_SD:
        shapes = full_model_shapes()
        if skip_enc_q:
            shapes = {k: v for k, v in shapes.items() if not k.startswith("enc_q.")}
        # The variant suite starts ~25 processes that all need these 142 MB: generated once per box, then read back from /tmp
        # (file name = digest of the generator's source, the seed and the shape list, so a change of any of them regenerates)
        import hashlib, tempfile
        h = hashlib.sha1()
        h.update(open(sw.__file__, "rb").read())
        h.update(repr((seed, sorted(shapes.items()))).encode())
        path = os.path.join(tempfile.gettempdir(), f"svoc_test_weights_{h.hexdigest()[:16]}.npz")
        sd = None
        if os.path.exists(path):
            try:
                with np.load(path) as z:
                    sd = {k: z[k] for k in z.files}
                if set(sd) != set(shapes) or any(tuple(sd[k].shape) != tuple(shapes[k]) for k in shapes):
                    sd = None
            except Exception:   # noqa: BLE001 - a torn or foreign file: regenerate
                sd = None
        if sd is None:
            sd = sw.fill_state_dict(shapes, seed=seed)
            try:
                tmp = f"{path}.{os.getpid()}.tmp.npz"
                np.savez(tmp, **sd)
                os.replace(tmp, path)
            except Exception:   # noqa: BLE001 - read-only /tmp: every process generates its own
                pass
        _FULL_SD[key] = sd
    return _FULL_SD[key]


def golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
