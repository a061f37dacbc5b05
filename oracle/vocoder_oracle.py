"""CPU oracle for the SMART-Vocoder inference path.  TEST INFRASTRUCTURE ONLY.

This is a from-the-maths restatement of the reference's ``SynthesizerTrn.infer``
call graph in plain fp32 ``torch.nn.functional`` ops over a flat
``{name: tensor}`` state dict (no nn.Module, no weight-norm hooks).  It is the
checker for the HIP path, and the ``cpu_baseline`` leg of bench.py; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py`` may import it.  The
product (``smart-vocoder_amd/``) never does.

Pinning: the reference owns no tests or golden vectors for this path
(SURVEY.md §4), so the oracle is pinned against outputs of the reference
itself, imported in the build container by ``tests/golden/make_golden.py``;
``tests/test_oracle_golden.py`` checks every committed fixture to <=2e-6 abs.

Reference lines followed (paths relative to the reference repo):
  fold_weight_norm ....... torch weight_norm as used at modules.py:128,135,145,191-206; models.py:125
  sequence_mask .......... commons.py:121-125 (+ .to(dtype) at models.py:40)
  gate ................... commons.py:100-107
  wn ..................... modules.py:148-176
  mel_encoder ............ models.py:35-47
  coupling ............... modules.py:324-343
  flow ................... models.py:73-80, Flip modules.py:270-277
  resblock1/resblock2 .... modules.py:210-223 / 244-252
  generator .............. models.py:141-160
  infer .................. models.py:331-339
  layer_norm_c/dds_conv .. modules.py:28-32 / 96-108
  conv_flow .............. modules.py:363-390
  rq_spline* ............. transforms.py:47-193
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1


# --------------------------------------------------------------------------- weights
def fold_weight_norm(v, g):
    """w = g * v / ||v||, norm over every dim but 0 (Conv1d: out channel; ConvTranspose1d: in channel)."""
    nrm = v.reshape(v.shape[0], -1).norm(dim=1).reshape([-1] + [1] * (v.dim() - 1))
    return v * (g / nrm)


def conv_weight(sd, prefix):
    """Folded weight of a (possibly weight-normed) conv at `prefix`."""
    if prefix + ".weight" in sd:
        return sd[prefix + ".weight"]
    return fold_weight_norm(sd[prefix + ".weight_v"], sd[prefix + ".weight_g"])


def _bias(sd, prefix):
    return sd.get(prefix + ".bias", None)


def _conv(sd, prefix, x, dilation=1, padding=0, groups=1):
    return F.conv1d(x, conv_weight(sd, prefix), _bias(sd, prefix), dilation=dilation, padding=padding, groups=groups)


def same_padding(k, d=1):
    return (k * d - d) // 2


# --------------------------------------------------------------------------- helpers
def sequence_mask(lengths, T, dtype=torch.float32):
    t = torch.arange(T, dtype=lengths.dtype, device=lengths.device)
    return (t[None, :] < lengths[:, None]).to(dtype)[:, None, :]


def gate(a, g, H):
    s = a + g
    return torch.tanh(s[:, :H]) * torch.sigmoid(s[:, H:])


# --------------------------------------------------------------------------- WN
def wn(sd, prefix, x, mask, g, *, hidden, kernel_size, dilation_rate, n_layers):
    out = torch.zeros_like(x)
    gc = _conv(sd, prefix + "cond_layer", g) if g is not None else None
    for i in range(n_layers):
        d = dilation_rate ** i
        a = _conv(sd, f"{prefix}in_layers.{i}", x, dilation=d, padding=same_padding(kernel_size, d))
        if gc is not None:
            gl = gc[:, 2 * hidden * i: 2 * hidden * (i + 1)]
        else:
            gl = torch.zeros_like(a)
        acts = gate(a, gl, hidden)
        rs = _conv(sd, f"{prefix}res_skip_layers.{i}", acts)
        if i < n_layers - 1:
            x = (x + rs[:, :hidden]) * mask
            out = out + rs[:, hidden:]
        else:
            out = out + rs
    return out * mask


# --------------------------------------------------------------------------- encoder / flow
def mel_encoder(sd, mel, lengths, prefix="enc_p.", *, hidden=192, kernel_size=5, dilation_rate=1, n_layers=16):
    """reference models.py:35-47; g is overwritten with None there (models.py:36), so cond_layer is never applied."""
    x = _conv(sd, prefix + "pre_enc", mel)
    mask = sequence_mask(lengths, x.shape[2], x.dtype)
    x = wn(sd, prefix + "encoder.", x * mask, mask, None, hidden=hidden, kernel_size=kernel_size,
           dilation_rate=dilation_rate, n_layers=n_layers)
    stats = _conv(sd, prefix + "proj", x) * mask
    C = stats.shape[1] // 2
    return x, stats[:, :C], stats[:, C:], mask


def posterior_encoder(sd, prefix, x, lengths, g, eps, *, hidden, kernel_size, dilation_rate, n_layers):
    """reference models.py:105-112 (PosteriorEncoder.forward) with the randn_like draw replaced by eps."""
    mask = sequence_mask(lengths, x.shape[2], x.dtype)
    h = _conv(sd, prefix + "pre", x) * mask
    h = wn(sd, prefix + "enc.", h, mask, g, hidden=hidden, kernel_size=kernel_size, dilation_rate=dilation_rate, n_layers=n_layers)
    stats = _conv(sd, prefix + "proj", h) * mask
    C = stats.shape[1] // 2
    m, logs = stats[:, :C], stats[:, C:]
    z = (m + eps * torch.exp(logs)) * mask
    return z, m, logs, mask


def coupling(sd, prefix, x, mask, g=None, *, reverse, hidden=192, kernel_size=5, dilation_rate=1, n_layers=8,
             mean_only=True):
    half = x.shape[1] // 2
    x0, x1 = x[:, :half], x[:, half:]
    h = _conv(sd, prefix + "pre", x0) * mask
    h = wn(sd, prefix + "enc.", h, mask, g, hidden=hidden, kernel_size=kernel_size, dilation_rate=dilation_rate,
           n_layers=n_layers)
    stats = _conv(sd, prefix + "post", h) * mask
    if mean_only:
        m, logs = stats, torch.zeros_like(stats)
    else:
        m, logs = stats[:, :half], stats[:, half:]
    if reverse:
        x1 = (x1 - m) * torch.exp(-logs) * mask
        return torch.cat([x0, x1], 1)
    x1 = m + x1 * torch.exp(logs) * mask
    return torch.cat([x0, x1], 1), logs.sum(dim=(1, 2))


def flow(sd, x, mask, g=None, *, reverse, prefix="flow.", n_flows=4, **kw):
    if reverse:
        for i in reversed(range(n_flows)):
            x = torch.flip(x, [1])
            x = coupling(sd, f"{prefix}flows.{2 * i}.", x, mask, g, reverse=True, **kw)
        return x
    for i in range(n_flows):
        x, _ = coupling(sd, f"{prefix}flows.{2 * i}.", x, mask, g, reverse=False, **kw)
        x = torch.flip(x, [1])
    return x


# --------------------------------------------------------------------------- decoder
def resblock1(sd, prefix, x, k, dilations=(1, 3, 5), mask=None):
    for i, d in enumerate(dilations):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        if mask is not None:
            xt = xt * mask
        xt = _conv(sd, f"{prefix}convs1.{i}", xt, dilation=d, padding=same_padding(k, d))
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        if mask is not None:
            xt = xt * mask
        xt = _conv(sd, f"{prefix}convs2.{i}", xt, padding=same_padding(k, 1))
        x = xt + x
    return x * mask if mask is not None else x


def resblock2(sd, prefix, x, k, dilations=(1, 3), mask=None):
    for i, d in enumerate(dilations):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        if mask is not None:
            xt = xt * mask
        xt = _conv(sd, f"{prefix}convs.{i}", xt, dilation=d, padding=same_padding(k, d))
        x = xt + x
    return x * mask if mask is not None else x


def generator(sd, x, g=None, *, prefix="dec.", resblock="1", resblock_kernel_sizes=(3, 7, 11),
              resblock_dilation_sizes=((1, 3, 5),) * 3, upsample_rates=(8, 8, 2, 2),
              upsample_kernel_sizes=(16, 16, 4, 4), return_stages=False):
    x = _conv(sd, prefix + "conv_pre", x, padding=3)
    if g is not None:
        x = x + _conv(sd, prefix + "cond", g)
    nk = len(resblock_kernel_sizes)
    stages = []
    for i, (u, ku) in enumerate(zip(upsample_rates, upsample_kernel_sizes)):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, conv_weight(sd, f"{prefix}ups.{i}"), _bias(sd, f"{prefix}ups.{i}"), stride=u,
                               padding=(ku - u) // 2)
        xs = None
        for j, (k, d) in enumerate(zip(resblock_kernel_sizes, resblock_dilation_sizes)):
            rb = resblock1 if resblock == "1" else resblock2
            y = rb(sd, f"{prefix}resblocks.{i * nk + j}.", x, k, tuple(d))
            xs = y if xs is None else xs + y
        x = xs / nk
        stages.append(x)
    x = F.leaky_relu(x)  # default slope 0.01 (reference models.py:156)
    x = F.conv1d(x, sd[prefix + "conv_post.weight"], None, padding=3)
    x = torch.tanh(x)
    return (x, stages) if return_stages else x


# --------------------------------------------------------------------------- full path
def infer(sd, mel, lengths, eps, noise_scale=1.0, max_len=None, **gen_kw):
    """Reference SynthesizerTrn.infer with the randn_like draw replaced by the given eps."""
    x, m_p, logs_p, mask = mel_encoder(sd, mel, lengths)
    z_p = m_p + eps * torch.exp(logs_p) * noise_scale
    z = flow(sd, z_p, mask, None, reverse=True)
    o = generator(sd, (z * mask)[:, :, :max_len], None, **gen_kw)
    return o, mask, (z, z_p, m_p, logs_p)


# --------------------------------------------------------------------------- DDSConv / ConvFlow / spline
def layer_norm_c(x, gamma, beta, eps=1e-5):
    return F.layer_norm(x.transpose(1, -1), (x.shape[1],), gamma, beta, eps).transpose(1, -1)


def dds_conv(sd, prefix, x, mask, g=None, *, kernel_size, n_layers):
    if g is not None:
        x = x + g
    C = x.shape[1]
    for i in range(n_layers):
        d = kernel_size ** i
        y = _conv(sd, f"{prefix}convs_sep.{i}", x * mask, dilation=d, padding=same_padding(kernel_size, d), groups=C)
        y = F.gelu(layer_norm_c(y, sd[f"{prefix}norms_1.{i}.gamma"], sd[f"{prefix}norms_1.{i}.beta"]))
        y = _conv(sd, f"{prefix}convs_1x1.{i}", y)
        y = F.gelu(layer_norm_c(y, sd[f"{prefix}norms_2.{i}.gamma"], sd[f"{prefix}norms_2.{i}.beta"]))
        x = x + y
    return x * mask


def _spline_knots(unnorm, lo, hi, min_size):
    nb = unnorm.shape[-1]
    w = min_size + (1.0 - min_size * nb) * F.softmax(unnorm, dim=-1)
    cum = F.pad(torch.cumsum(w, dim=-1), (1, 0))
    cum = (hi - lo) * cum + lo
    cum = cum.clone()
    cum[..., 0] = lo
    cum[..., -1] = hi
    return cum, cum[..., 1:] - cum[..., :-1]


def rq_spline(x, uw, uh, ud, inverse=False, left=0.0, right=1.0, bottom=0.0, top=1.0, min_bin=1e-3, min_deriv=1e-3,
              min_bin_height=None):
    """Rational-quadratic spline on [left,right]->[bottom,top]; ud has num_bins+1 entries (transforms.py:96-193).
    min_bin = min_bin_width; min_bin_height defaults to the same value."""
    if x.numel() and (float(x.min()) < left or float(x.max()) > right):
        raise ValueError('Input to a transform is not within its domain')     # transforms.py:105-106
    cw, w = _spline_knots(uw, left, right, min_bin)
    ch, h = _spline_knots(uh, bottom, top, min_bin if min_bin_height is None else min_bin_height)
    dv = min_deriv + F.softplus(ud)
    knots = (ch if inverse else cw).clone()
    knots[..., -1] += 1e-6
    idx = ((x[..., None] >= knots).sum(-1) - 1)[..., None]
    pick = lambda a: a.gather(-1, idx)[..., 0]
    x_k, w_k, y_k, h_k = pick(cw), pick(w), pick(ch), pick(h)
    s_k = pick(h / w)
    d0, d1 = pick(dv), pick(dv[..., 1:])
    if inverse:
        dy = x - y_k
        t = dy * (d0 + d1 - 2 * s_k)
        a = t + h_k * (s_k - d0)
        b = h_k * d0 - t
        c = -s_k * dy
        root = (2 * c) / (-b - torch.sqrt(b * b - 4 * a * c))
        out = root * w_k + x_k
        th = root
    else:
        th = (x - x_k) / w_k
    tt = th * (1 - th)
    den = s_k + (d0 + d1 - 2 * s_k) * tt
    lad = torch.log(s_k * s_k * (d1 * th * th + 2 * s_k * tt + d0 * (1 - th) * (1 - th))) - 2 * torch.log(den)
    if inverse:
        return out, -lad
    return y_k + h_k * (s_k * th * th + d0 * tt) / den, lad


def rq_spline_linear_tails(x, uw, uh, ud, inverse=False, tail_bound=5.0, min_bin=1e-3, min_deriv=1e-3,
                           min_bin_height=None):
    """transforms.py:55-94: identity outside [-B,B]; boundary derivatives fixed so that softplus+min_derivative == 1."""
    inside = (x >= -tail_bound) & (x <= tail_bound)
    c = float(np.log(np.exp(1 - min_deriv) - 1))
    ud = F.pad(ud, (1, 1), value=c)
    out = x.clone()
    lad = torch.zeros_like(x)
    if inside.any():
        o, l = rq_spline(x[inside], uw[inside], uh[inside], ud[inside], inverse, -tail_bound, tail_bound,
                         -tail_bound, tail_bound, min_bin, min_deriv, min_bin_height)
        out[inside] = o
        lad[inside] = l
    return out, lad


def conv_flow(sd, prefix, x, mask, g=None, *, reverse, filter_channels, kernel_size, n_layers, num_bins=10,
              tail_bound=5.0):
    half = x.shape[1] // 2
    x0, x1 = x[:, :half], x[:, half:]
    h = _conv(sd, prefix + "pre", x0)
    h = dds_conv(sd, prefix + "convs.", h, mask, g, kernel_size=kernel_size, n_layers=n_layers)
    h = _conv(sd, prefix + "proj", h) * mask
    b, c, t = x0.shape
    h = h.reshape(b, c, -1, t).permute(0, 1, 3, 2)
    s = math.sqrt(filter_channels)
    y1, lad = rq_spline_linear_tails(x1, h[..., :num_bins] / s, h[..., num_bins:2 * num_bins] / s,
                                     h[..., 2 * num_bins:], inverse=reverse, tail_bound=tail_bound)
    y = torch.cat([x0, y1], 1) * mask
    if reverse:
        return y
    return y, (lad * mask).sum(dim=(1, 2))
