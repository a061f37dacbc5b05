"""Model classes with the reference's names, constructor signatures, attribute names and state_dict
layout (reference models.py), running SynthesizerTrn.infer on MI355X through libsvoc_hip.so.

``SynthesizerTrn(...).cuda().eval()``, ``utils.load_checkpoint(path, net_g, None)`` and
``net_g.infer(mel, lengths, noise_scale=.667)`` behave as in the reference's inference.ipynb.
The whole of ``infer`` is ONE call into the C ABI (svoc_synth_infer): mel encoder (16-layer WN),
reparameterisation, 4 reverse coupling layers with folded Flips, HiFi-GAN decoder.

Training-side classes (DiscriminatorP/S, MultiPeriodDiscriminator) and ``SynthesizerTrn.forward``
are out of scope; PosteriorEncoder is kept as a parameter container so the
state_dict has the reference's 659 keys.
"""
import torch
from torch import nn

try:
    from . import _native as N
    from . import commons, modules
    from .commons import get_padding, init_weights
    from .modules import (_Conv1dParams, _WNConvTranspose1dParams, _HipModule, _fold_in_place, _g_args, _mask_arg,
                          _check_channels, _same_device)
except ImportError:
    import _native as N
    import commons
    import modules
    from commons import get_padding, init_weights
    from modules import (_Conv1dParams, _WNConvTranspose1dParams, _HipModule, _fold_in_place, _g_args, _mask_arg,
                         _check_channels, _same_device)


class MelEncoder(_HipModule):
    """reference models.py:15-47.  Inside SynthesizerTrn.infer it is part of the single native call; called on its own
    (``net_g.enc_p(mel, lengths)``) it runs the same kernels behind svoc_mel_encoder_forward."""
    _destroy = "svoc_mel_encoder_destroy"

    def __init__(self, out_channels, hidden_channels, filter_channels, n_layers, kernel_size, dilation_rate, gin_channels):
        super().__init__()
        self.out_channels = out_channels
        self.hidden_channels = hidden_channels
        self.filter_channels = filter_channels
        self.n_layers = n_layers
        self.kernel_size = kernel_size
        self.dilation_rate = dilation_rate
        self.encoder = modules.WN(hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=gin_channels)
        self.gin_channels = gin_channels
        self.pre_enc = _Conv1dParams(80, hidden_channels, 1)
        self.proj = _Conv1dParams(hidden_channels, out_channels * 2, 1)

    def _create(self, h, tab):
        N.check(N.lib().svoc_mel_encoder_create(h.out(), 80, self.out_channels, self.hidden_channels, self.kernel_size,
                                                self.dilation_rate, self.n_layers, self.gin_channels, tab.arr, tab.n, b""))

    def forward(self, x, x_lengths, g=None):
        """-> (x, m, logs, x_mask) like the reference; `g` is discarded there too (models.py:36)."""
        x = N.f32(x)
        if x.dim() != 3:
            raise ValueError(f"expected mel [B, 80, T], got {tuple(x.shape)}")
        B, Cc, T = x.shape
        _check_channels("MelEncoder", Cc, 80)
        ln = x_lengths.to(device=x.device, dtype=torch.int64).contiguous()
        dev = x.device
        xo = torch.empty(B, self.hidden_channels, T, dtype=torch.float32, device=dev)
        m, logs = (torch.empty(B, self.out_channels, T, dtype=torch.float32, device=dev) for _ in range(2))
        x_mask = torch.empty(B, 1, T, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            N.check(N.lib().svoc_mel_encoder_forward(self._native(), N.stream_ptr(dev), N.ptr(x), N.ptr(ln), N.ptr(xo), N.ptr(m),
                                                     N.ptr(logs), N.ptr(x_mask), B, T))
        return xo, m, logs, x_mask


class ResidualCouplingBlock(_HipModule):
    """reference models.py:50-80: n_flows x (ResidualCouplingLayer(mean_only=True), Flip)."""
    _destroy = "svoc_flow_destroy"

    def __init__(self, channels, hidden_channels, kernel_size, dilation_rate, n_layers, n_flows=4, gin_channels=0):
        super().__init__()
        self.channels = channels
        self.hidden_channels = hidden_channels
        self.kernel_size = kernel_size
        self.dilation_rate = dilation_rate
        self.n_layers = n_layers
        self.n_flows = n_flows
        self.gin_channels = gin_channels
        self.flows = nn.ModuleList()
        for i in range(n_flows):
            self.flows.append(modules.ResidualCouplingLayer(channels, hidden_channels, kernel_size, dilation_rate, n_layers,
                                                            gin_channels=gin_channels, mean_only=True))
            self.flows.append(modules.Flip())

    def _create(self, h, tab):
        N.check(N.lib().svoc_flow_create(h.out(), self.channels, self.hidden_channels, self.kernel_size, self.dilation_rate,
                                         self.n_layers, self.n_flows, self.gin_channels, tab.arr, tab.n, b""))

    def forward(self, x, x_mask, g=None, reverse=False):
        x = N.f32(x)
        B, Cc, T = x.shape
        _check_channels("ResidualCouplingBlock", Cc, self.channels)
        m = _mask_arg(x_mask, B, T, x.device)
        g, gT = _g_args(g, T, B, self.gin_channels)
        y = torch.empty_like(x)
        _same_device(x, m, g)
        with torch.cuda.device(x.device):
            N.check(N.lib().svoc_flow_forward(self._native(), N.stream_ptr(x.device), N.ptr(x), N.ptr(m), N.ptr(g), gT,
                                              1 if reverse else 0, N.ptr(y), B, T))
        return y


class PosteriorEncoder(_HipModule):
    """reference models.py:83-112 (enc_q of training / voice conversion; not part of ``infer``).

    ``forward`` accepts one extra keyword, ``eps=``, to inject the N(0,1) draw the reference takes with
    ``torch.randn_like(m)`` (models.py:111)."""
    _destroy = "svoc_posterior_destroy"

    def __init__(self, in_channels, out_channels, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=0):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.hidden_channels = hidden_channels
        self.kernel_size = kernel_size
        self.dilation_rate = dilation_rate
        self.n_layers = n_layers
        self.gin_channels = gin_channels
        self.pre = _Conv1dParams(in_channels, hidden_channels, 1)
        self.enc = modules.WN(hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=gin_channels)
        self.proj = _Conv1dParams(hidden_channels, out_channels * 2, 1)

    def _create(self, h, tab):
        N.check(N.lib().svoc_posterior_create(h.out(), self.in_channels, self.out_channels, self.hidden_channels, self.kernel_size,
                                              self.dilation_rate, self.n_layers, self.gin_channels, tab.arr, tab.n, b""))

    def forward(self, x, x_lengths, g=None, eps=None):
        x = N.f32(x)
        B, Cc, T = x.shape
        if Cc != self.in_channels:
            raise ValueError(f"expected {self.in_channels} input channels, got {Cc}")
        lengths = x_lengths.to(device=x.device, dtype=torch.int64).contiguous()
        g, gT = modules._g_args(g, T, B, self.gin_channels)
        if eps is None:
            eps = torch.randn(B, self.out_channels, T, device=x.device, dtype=torch.float32)
        eps = N.f32(eps)
        if tuple(eps.shape) != (B, self.out_channels, T):
            raise ValueError(f"eps must be [{B}, {self.out_channels}, {T}], got {tuple(eps.shape)}")
        z, m, logs = (torch.empty(B, self.out_channels, T, device=x.device, dtype=torch.float32) for _ in range(3))
        x_mask = torch.empty(B, 1, T, device=x.device, dtype=torch.float32)
        _same_device(x, g, eps)
        with torch.cuda.device(x.device):
            N.check(N.lib().svoc_posterior_forward(self._native(), N.stream_ptr(x.device), N.ptr(x), N.ptr(lengths), N.ptr(g), gT,
                                                   N.ptr(eps), N.ptr(z), N.ptr(m), N.ptr(logs), N.ptr(x_mask), B, T))
        return z, m, logs, x_mask


def _gen_config(initial_channel, resblock, rks, rds, ur, uic, uks, gin):
    c = N.svoc_generator_config()
    if len(rks) > 8 or len(ur) > 8 or any(len(d) > 8 for d in rds):
        raise ValueError("at most 8 resblock kernels / dilations / upsamples are supported")
    c.initial_channel = initial_channel
    c.resblock_kind = 1 if resblock == '1' else 2
    c.n_kernels = len(rks)
    for j, (k, d) in enumerate(zip(rks, rds)):
        c.resblock_kernel_sizes[j] = k
        c.n_dilations[j] = len(d)
        for q, dd in enumerate(d):
            c.resblock_dilation_sizes[j][q] = dd
    c.n_upsamples = len(ur)
    for i, (u, k) in enumerate(zip(ur, uks)):
        c.upsample_rates[i] = u
        c.upsample_kernel_sizes[i] = k
    c.upsample_initial_channel = uic
    c.gin_channels = gin
    return c


class Generator(_HipModule):
    """HiFi-GAN V1 style decoder (reference models.py:115-167)."""
    _destroy = "svoc_generator_destroy"

    def __init__(self, initial_channel, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                 upsample_initial_channel, upsample_kernel_sizes, gin_channels=0):
        super(Generator, self).__init__()
        self.num_kernels = len(resblock_kernel_sizes)
        self.num_upsamples = len(upsample_rates)
        self._cfg_args = (initial_channel, resblock, list(resblock_kernel_sizes), [list(d) for d in resblock_dilation_sizes],
                          list(upsample_rates), upsample_initial_channel, list(upsample_kernel_sizes), gin_channels)
        self.gin_channels = gin_channels
        self.conv_pre = _Conv1dParams(initial_channel, upsample_initial_channel, 7)
        rb = modules.ResBlock1 if resblock == '1' else modules.ResBlock2
        self.ups = nn.ModuleList()
        for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)):
            self.ups.append(_WNConvTranspose1dParams(upsample_initial_channel // (2 ** i),
                                                     upsample_initial_channel // (2 ** (i + 1)), k))
        self.resblocks = nn.ModuleList()
        for i in range(len(self.ups)):
            ch = upsample_initial_channel // (2 ** (i + 1))
            for j, (k, d) in enumerate(zip(resblock_kernel_sizes, resblock_dilation_sizes)):
                self.resblocks.append(rb(ch, k, d))
        self.conv_post = _Conv1dParams(ch, 1, 7, bias=False)
        if gin_channels != 0:
            self.cond = _Conv1dParams(gin_channels, upsample_initial_channel, 1)
        self.hop = 1
        for u in upsample_rates:
            self.hop *= u

    def _config(self):
        return _gen_config(*self._cfg_args)

    def _create(self, h, tab):
        cfg = self._config()
        N.check(N.lib().svoc_generator_create(h.out(), cfg, tab.arr, tab.n, b""))

    def forward(self, x, g=None):
        x = N.f32(x)
        B, Cc, T = x.shape
        _check_channels("Generator", Cc, self._cfg_args[0])
        if g is not None:
            if g.dim() != 3 or g.shape[2] != 1:
                raise ValueError("g must be [B, gin_channels, 1]")
            g, _ = _g_args(g, T, B, self.gin_channels)
        out = torch.empty(B, 1, T * self.hop, dtype=torch.float32, device=x.device)
        _same_device(x, g)
        with torch.cuda.device(x.device):
            N.check(N.lib().svoc_generator_forward(self._native(), N.stream_ptr(x.device), N.ptr(x), T, Cc * T, None, 0,
                                                   N.ptr(g), N.ptr(out), B, T))
        return out

    def remove_weight_norm(self):
        print('Removing weight norm...')
        for l in self.ups:
            _fold_in_place(l)
        for l in self.resblocks:
            l.remove_weight_norm()


def decoder_receptive_frames(resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates, upsample_kernel_sizes):
    """Half-width of the Generator's receptive field in mel frames (reference models.py:115-160): conv_pre (k 7) + per stage the
    transposed convolution's taps and the widest ResBlock chain (ResBlock1: every dilated conv is followed by an undilated one,
    modules.py:190-207; ResBlock2: dilated convs only, modules.py:235-240), each divided by the stage's rate, + conv_post (k 7);
    rounded up, plus two frames of margin.  Frames past an utterance's end closer than this still reach samples inside it, so
    length-bucketed shards keep that many (parallel.infer_sharded)."""
    import math
    frames, rate = 3.0, 1
    for u, ku in zip(upsample_rates, upsample_kernel_sizes):
        frames += math.ceil(ku / (2.0 * u)) / rate          # input samples either side that one output sample of the upsampler sees
        rate *= u
        widest = 0
        for k, dils in zip(resblock_kernel_sizes, resblock_dilation_sizes):
            w = sum((k - 1) * d // 2 + ((k - 1) // 2 if str(resblock) == "1" else 0) for d in dils)
            widest = max(widest, w)
        frames += widest / rate
    frames += 3.0 / rate
    return int(math.ceil(frames)) + 2


class SynthesizerTrn(_HipModule):
    """reference models.py:261-349.  Only the inference entry point is implemented."""
    _destroy = "svoc_synth_destroy"

    def __init__(self, spec_channels, segment_size, inter_channels, hidden_channels, filter_channels, n_heads, n_layers,
                 kernel_size, p_dropout, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                 upsample_initial_channel, upsample_kernel_sizes, n_speakers=0, gin_channels=0, **kwargs):
        super().__init__()
        self.spec_channels = spec_channels
        self.inter_channels = inter_channels
        self.hidden_channels = hidden_channels
        self.filter_channels = filter_channels
        self.n_heads = n_heads
        self.n_layers = n_layers
        self.kernel_size = kernel_size
        self.p_dropout = p_dropout
        self.resblock = resblock
        self.resblock_kernel_sizes = resblock_kernel_sizes
        self.resblock_dilation_sizes = resblock_dilation_sizes
        self.upsample_rates = upsample_rates
        self.upsample_initial_channel = upsample_initial_channel
        self.upsample_kernel_sizes = upsample_kernel_sizes
        self.segment_size = segment_size
        self.n_speakers = n_speakers
        self.gin_channels = gin_channels
        # sizes hard-coded by the reference (models.py:305-314), not taken from the config
        self.enc_p = MelEncoder(inter_channels, hidden_channels, filter_channels, n_layers=16, kernel_size=5,
                                dilation_rate=1, gin_channels=gin_channels)
        self.dec = Generator(inter_channels, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                             upsample_initial_channel, upsample_kernel_sizes, gin_channels=gin_channels)
        self.enc_q = PosteriorEncoder(spec_channels, inter_channels, hidden_channels, 5, 1, 16, gin_channels=gin_channels)
        self.flow = ResidualCouplingBlock(inter_channels, hidden_channels, 5, 1, 8, gin_channels=gin_channels)
        self.DECODER_RECEPTIVE_FRAMES = decoder_receptive_frames(resblock, resblock_kernel_sizes, resblock_dilation_sizes,
                                                                 upsample_rates, upsample_kernel_sizes)

    # enc_q is not part of infer: keep it out of the native table and of the change signature
    def _state(self):
        return {k: v for k, v in self.state_dict(keep_vars=True).items() if not k.startswith("enc_q.")}

    def _dev(self):
        p = self.dec.conv_pre.weight
        N.require_gpu(p)
        return p.device

    def _create(self, h, tab):
        c = N.svoc_synth_config()
        c.n_mel = 80
        c.inter_channels = self.inter_channels
        c.hidden_channels = self.hidden_channels
        c.enc_n_layers, c.enc_kernel_size, c.enc_dilation_rate = 16, 5, 1
        c.flow_n_layers, c.flow_kernel_size, c.flow_dilation_rate, c.flow_n_flows = 8, 5, 1, 4
        c.gin_channels = self.gin_channels
        c.dec = self.dec._config()
        N.check(N.lib().svoc_synth_create(h.out(), c, tab.arr, tab.n))

    def forward(self, x, x_lengths, y, y_lengths, sid=None):
        raise NotImplementedError("training forward is outside the MI355X inference path; use infer()")

    def infer(self, x, x_lengths, sid=None, noise_scale=1, length_scale=1, noise_scale_w=1., max_len=None, eps=None):
        """mel [B,80,T], lengths [B] -> (o [B,1,min(T,max_len)*hop], x_mask [B,1,T], (z, z_p, m_p, logs_p)).

        As in the reference, speaker conditioning is off (g=None, models.py:332) and sid/length_scale/noise_scale_w
        are ignored.  `eps` (extension, keyword only in spirit) injects the N(0,1) draw the reference takes with
        randn_like (models.py:336); by default it is drawn exactly as there - ``torch.randn_like`` of the [B, 192, T] fp32
        ``m_p`` on the device - so after ``torch.manual_seed(s)`` the call consumes the generator stream the reference's
        call would (tests/test_gpu_parity.py::test_default_noise_is_the_reference_draw).

        Blocking behaviour: the call only enqueues on the current stream, except that the first call at a new (B, T) sizes every
        workspace for it first (``svoc_synth_reserve``: a no-op unless one has to grow; then one device synchronisation +
        reallocation, BEFORE anything of this call is enqueued, never in the middle of it).  Call ``reserve(batch, frames)``
        once with the largest shape to keep every later ``infer`` free of allocations and synchronisations.
        """
        x = N.f32(x)
        if x.dim() != 3 or x.shape[1] != 80:
            raise ValueError(f"expected mel [B, 80, T], got {tuple(x.shape)}")
        B, _, T = x.shape
        ln = x_lengths.to(device=x.device, dtype=torch.int64).contiguous()
        Td = T if max_len is None else max(0, min(T, int(max_len) if max_len >= 0 else T + int(max_len)))
        dev = x.device
        IC = self.inter_channels
        z, z_p, m_p, logs_p = (torch.empty(B, IC, T, dtype=torch.float32, device=dev) for _ in range(4))
        if eps is None:
            eps = torch.randn_like(m_p)                      # the reference's draw (models.py:336)
        eps = N.f32(eps)
        if tuple(eps.shape) != (B, self.inter_channels, T):
            raise ValueError("eps must be [B, inter_channels, T]")
        o = torch.empty(B, 1, Td * self.dec.hop, dtype=torch.float32, device=dev)
        x_mask = torch.empty(B, 1, T, dtype=torch.float32, device=dev)
        if Td == 0:
            raise ValueError("max_len leaves no frames to decode")
        _same_device(x, eps)
        with torch.cuda.device(dev):
            h = self._native()
            seen = self._nh.__dict__.setdefault("reserved_shapes", set())   # (on the Handle object: a rebuilt handle starts from nothing)
            if (B, T) not in seen:
                # first sight of this shape: size every workspace for it now, ahead of the first launch of this call
                # (a no-op unless some workspace has to grow; then: one device synchronisation + reallocation)
                N.check(N.lib().svoc_synth_reserve(h, B, T))
                if len(seen) >= 4096:
                    seen.clear()
                seen.add((B, T))
            N.check(N.lib().svoc_synth_infer(h, N.stream_ptr(dev), N.ptr(x), N.ptr(ln), N.ptr(eps),
                                             float(noise_scale), Td, N.ptr(o), N.ptr(x_mask), N.ptr(z), N.ptr(z_p),
                                             N.ptr(m_p), N.ptr(logs_p), B, T))
        return o, x_mask, (z, z_p, m_p, logs_p)

    def reserve(self, batch, frames):
        """Extension: size the library's workspaces for batches up to [batch, 80, frames] now, so that later ``infer``
        calls neither allocate nor synchronise (svoc_synth_reserve)."""
        with torch.cuda.device(self._dev()):
            h = self._native()
            N.check(N.lib().svoc_synth_reserve(h, int(batch), int(frames)))
            self._nh.__dict__.setdefault("reserved_shapes", set()).add((int(batch), int(frames)))
        return self

    # receptive half-width of the whole path in mel frames: encoder WN 16*2 + flow 4*8*2 + decoder (conv_pre 3,
    # and <= 60/8 + 60/64 + 60/128 + 60/256 + upsampler taps) -- 128 covers it with margin
    RECEPTIVE_FRAMES = 128
    # decoder alone (frames past an utterance's end that still influence samples inside it; encoder and flows are masked
    # layer by layer, so nothing else reaches across the end): computed from the configuration in __init__
    # (decoder_receptive_frames; iitp_base: 3 + 1 + 60/8 + 60/64 + 60/128 + 60/256 + taps -> 16)
    DECODER_RECEPTIVE_FRAMES = 16

    def infer_chunked(self, x, x_lengths, chunk_frames=1024, noise_scale=1, eps=None, halo_frames=None):
        """Long-form inference by time tiling: the mel is cut into chunks of `chunk_frames`, each is
        run with `halo_frames` of real context on both sides, and only the interior samples are kept.  With a halo
        of at least RECEPTIVE_FRAMES every output sample sees exactly the inputs it sees in one-shot `infer`, so the
        result equals it up to fp32 summation order (the library picks tile shapes / K splits from the batch and
        length, e.g. the K-split WN kernel for short chunks, so the two are not guaranteed bit-identical: the test
        bounds the difference at 1e-5 absolute), while the activation workspace stays bounded by the chunk size.
        Returns the waveform [B, 1, T*hop] only."""
        x = N.f32(x)
        B, _, T = x.shape
        halo = self.RECEPTIVE_FRAMES if halo_frames is None else int(halo_frames)
        hop = self.dec.hop
        if eps is None:
            eps = torch.randn(B, self.inter_channels, T, dtype=torch.float32, device=x.device)
        eps = N.f32(eps)
        ln = x_lengths.to(device=x.device, dtype=torch.int64)
        out = torch.empty(B, 1, T * hop, dtype=torch.float32, device=x.device)
        for s in range(0, T, chunk_frames):
            e = min(T, s + chunk_frames)
            a, b = max(0, s - halo), min(T, e + halo)
            o = self.infer(x[:, :, a:b].contiguous(), torch.clamp(ln - a, min=0, max=b - a), noise_scale=noise_scale,
                           eps=eps[:, :, a:b].contiguous())[0]
            out[:, :, s * hop:e * hop] = o[:, :, (s - a) * hop:(e - a) * hop]
        return out

    def stream(self, noise_scale=1, halo_frames=None):
        """Incremental long-form inference: ``s = net.stream(noise_scale=.667)``, then ``wave = s.push(mel_chunk)``
        as mel frames arrive and ``wave = s.flush()`` at the end; the concatenated pieces equal one ``infer`` over the whole
        utterance (to fp32 summation order, as ``infer_chunked``).  See :class:`InferStream`."""
        return InferStream(self, noise_scale=noise_scale, halo_frames=halo_frames)

    def voice_conversion(self, y, y_lengths, sid_src, sid_tgt, eps=None):
        """reference models.py:341-349.  The reference never creates ``emb_g`` (models.py:305-314), so as there this
        raises AttributeError unless the caller attaches one (``net.emb_g = nn.Embedding(n_speakers, gin_channels)``)."""
        assert self.n_speakers > 0, "n_speakers have to be larger than 0."
        if not hasattr(self, "emb_g"):
            raise AttributeError("'SynthesizerTrn' object has no attribute 'emb_g'")
        g_src = self.emb_g(sid_src).unsqueeze(-1)
        g_tgt = self.emb_g(sid_tgt).unsqueeze(-1)
        z, m_q, logs_q, y_mask = self.enc_q(y, y_lengths, g=g_src, eps=eps)
        z_p = self.flow(z, y_mask, g=g_src)
        z_hat = self.flow(z_p, y_mask, g=g_tgt, reverse=True)
        o_hat = self.dec(z_hat * y_mask, g=g_tgt)
        return o_hat, y_mask, (z, z_p, z_hat)


class InferStream:
    """Feed-as-it-arrives front of ``SynthesizerTrn.infer`` (extension; the reference's only hook for partial decoding is
    ``max_len``, models.py:338).  The path is not causal - every output sample depends on ``RECEPTIVE_FRAMES`` mel frames on either
    side - so the stream keeps a halo of past frames, holds back the newest ``halo`` frames until their future context has
    arrived, and runs ``infer`` on [halo | ready frames | halo]; only the samples of the ready frames are returned.  Nothing
    but the halo is carried between calls (the library's workspaces are sized by the largest window seen).  ``push`` may return an
    empty waveform while the look-ahead fills; ``flush`` emits what is left (the true end is zero-padded by the convolutions
    exactly as in a one-shot call).  Every utterance of the batch advances in lock step (all frames of a chunk are valid)."""

    def __init__(self, net, noise_scale=1, halo_frames=None):
        self.net = net
        self.noise_scale = noise_scale
        self.halo = int(net.RECEPTIVE_FRAMES if halo_frames is None else halo_frames)
        self.mel = None          # frames [start, end) of the utterance: left halo + everything not yet emitted
        self.eps = None
        self.start = 0           # utterance index of self.mel[:, :, 0]
        self.emitted = 0         # frames whose samples have been returned

    @property
    def end(self):
        return self.start + (0 if self.mel is None else self.mel.shape[2])

    def _run(self, upto):
        """samples of frames [emitted, upto) from one infer over [start, end)"""
        hop = self.net.dec.hop
        B, _, W = self.mel.shape
        ln = torch.full((B,), W, dtype=torch.int64, device=self.mel.device)
        o = self.net.infer(self.mel, ln, noise_scale=self.noise_scale, eps=self.eps)[0]
        out = o[:, :, (self.emitted - self.start) * hop:(upto - self.start) * hop].clone()
        self.emitted = upto
        keep_from = max(self.start, self.emitted - self.halo)          # drop what can no longer influence future samples
        if keep_from > self.start:
            self.mel = self.mel[:, :, keep_from - self.start:].contiguous()
            self.eps = self.eps[:, :, keep_from - self.start:].contiguous()
            self.start = keep_from
        return out

    def push(self, mel, eps=None):
        """mel [B, 80, t] (t >= 1 new frames), eps [B, inter_channels, t] or None (drawn) -> waveform [B, 1, n * hop], n >= 0"""
        mel = N.f32(mel)
        if eps is None:
            eps = torch.randn(mel.shape[0], self.net.inter_channels, mel.shape[2], dtype=torch.float32, device=mel.device)
        eps = N.f32(eps)
        if self.mel is None:
            self.mel, self.eps = mel, eps
        else:
            if mel.shape[0] != self.mel.shape[0]:
                raise ValueError("the batch size of a stream is fixed by its first chunk")
            self.mel = torch.cat([self.mel, mel], 2)
            self.eps = torch.cat([self.eps, eps], 2)
        upto = self.end - self.halo                                   # frames whose right context is complete
        if upto <= self.emitted:
            return torch.empty(mel.shape[0], 1, 0, dtype=torch.float32, device=mel.device)
        return self._run(upto)

    def flush(self):
        """the samples of the frames still held back; the stream is empty afterwards"""
        if self.mel is None or self.end <= self.emitted:
            dev = self.net._dev()
            return torch.empty(0 if self.mel is None else self.mel.shape[0], 1, 0, dtype=torch.float32, device=dev)
        out = self._run(self.end)
        self.mel = self.eps = None
        self.start = self.emitted
        return out
