"""Building blocks of the vocoder with the reference's class names, constructor signatures and
state_dict layouts (reference modules.py), executed by HIP kernels through the C ABI.

Each class is a parameter container (so ``state_dict``/``load_state_dict``/``.cuda()`` behave like
the reference's modules, including the ``weight_g``/``weight_v`` pairs of torch weight-norm) plus a
``forward`` that hands the parameters to libsvoc_hip.so once (weights are folded and repacked on
the device at that point and again whenever a parameter changes) and then only enqueues kernels.
There is no PyTorch compute path; without the built library or a GPU, ``forward`` raises.

Not provided: ConvReluNorm, Log, ElementwiseAffine — never instantiated by the reference
(SURVEY.md §2 row 8).
"""
import ctypes as C
import math

import torch
from torch import nn

try:
    from . import _native as N
    from . import commons
    from .commons import get_padding, init_weights
except ImportError:
    import _native as N
    import commons
    from commons import get_padding, init_weights

LRELU_SLOPE = 0.1


# ----------------------------------------------------------------------------- parameter containers
def _conv_default_init_(w):
    nn.init.kaiming_uniform_(w, a=math.sqrt(5))


class _Conv1dParams(nn.Module):
    """state_dict keys of a plain nn.Conv1d: weight [Co,Ci,k], bias [Co]."""

    def __init__(self, cin, cout, k, bias=True, groups=1):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin // groups, k))
        _conv_default_init_(self.weight)
        if bias:
            bound = 1.0 / math.sqrt((cin // groups) * k)
            self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)


class _WNConv1dParams(nn.Module):
    """state_dict keys of weight_norm(nn.Conv1d): bias, weight_g [Co,1,1], weight_v [Co,Ci,k]."""

    def __init__(self, cin, cout, k):
        super().__init__()
        v = torch.empty(cout, cin, k)
        _conv_default_init_(v)
        bound = 1.0 / math.sqrt(cin * k)
        self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))
        self.weight_g = nn.Parameter(v.reshape(cout, -1).norm(dim=1).reshape(cout, 1, 1))
        self.weight_v = nn.Parameter(v)


class _WNConvTranspose1dParams(nn.Module):
    """weight_norm(nn.ConvTranspose1d): bias [Co], weight_g [Ci,1,1], weight_v [Ci,Co,k] (norm per INPUT channel)."""

    def __init__(self, cin, cout, k):
        super().__init__()
        v = torch.empty(cin, cout, k)
        _conv_default_init_(v)
        bound = 1.0 / math.sqrt(cout * k)
        self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))
        self.weight_g = nn.Parameter(v.reshape(cin, -1).norm(dim=1).reshape(cin, 1, 1))
        self.weight_v = nn.Parameter(v)


_STRUCT_EPOCH = [0]


def _fold_tensor(weight_v, weight_g):
    """w = g * v / ||v|| over all dims but 0 (torch.nn.utils.weight_norm, dim=0), on the device (HIP kernel)."""
    v = N.f32(weight_v.detach())
    g = N.f32(weight_g.detach())
    w = torch.empty_like(v)
    with torch.cuda.device(v.device):
        N.check(N.lib().svoc_fold_weight_norm(N.stream_ptr(v.device), N.ptr(v), N.ptr(g), N.ptr(w), v.shape[0],
                                              v[0].numel()))
    return w


def _fold_in_place(m):
    """remove_weight_norm for one container: replace (weight_g, weight_v) by the folded `weight` (HIP kernel)."""
    if not hasattr(m, "weight_v"):
        return
    w = _fold_tensor(m.weight_v, m.weight_g)
    bias = m.bias
    del m.weight_g, m.weight_v, m.bias
    m.weight = nn.Parameter(w)     # key order after removal in the reference: bias, weight
    m.bias = bias
    _STRUCT_EPOCH[0] += 1          # every cached tensor list is rebuilt on next use


def _check_channels(what, got, want):
    """The reference raises a conv shape error on a wrong channel count; the library derives strides from its own
    configuration, so a mismatch must be refused here (it would read or write out of bounds on the device)."""
    if got != want:
        raise ValueError(f"{what}: expected {want} channels, got {got}")


class _HipModule(nn.Module):
    """Caches the library handle built from this module's parameters.

    Change detection: the handle is rebuilt when a parameter is replaced (``.cuda()``, ``load_state_dict``,
    ``remove_weight_norm``) or modified in place through autograd-visible ops (``Parameter._version``).  Writes that
    bypass the version counter (``p.data.copy_()``, ``p.data = ...`` after the first forward) are NOT seen: call
    ``invalidate()`` afterwards.  Handles are per process and per device; ``copy.deepcopy`` / pickling drop them.
    """

    _destroy = None

    def invalidate(self):
        """Drop every cached library handle below this module (weights are re-folded and re-packed on next use)."""
        for m in self.modules():
            if isinstance(m, _HipModule):
                m._invalidate()

    def __getstate__(self):
        d = dict(self.__dict__)
        for k in ("_nh", "_nh_sig", "_nh_tensors", "_nh_epoch"):
            d.pop(k, None)
        return d

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__getstate__().items():
            object.__setattr__(new, k, copy.deepcopy(v, memo))
        return new

    def _state(self):
        """{state_dict key: tensor} handed to the library (parameters only; subclasses may filter)."""
        return dict(self.state_dict(keep_vars=True))

    def _sig(self):
        # cheap per-call change detection: the cached tensor list is rebuilt only when the module structure
        # changes (_apply / remove_weight_norm / load_state_dict(assign=True) all go through here via versions)
        st = self.__dict__.get("_nh_tensors")
        if st is None or self.__dict__.get("_nh_epoch") != _STRUCT_EPOCH[0]:
            st = list(self._state().values())
            object.__setattr__(self, "_nh_tensors", st)
            object.__setattr__(self, "_nh_epoch", _STRUCT_EPOCH[0])
        return (_STRUCT_EPOCH[0], len(st), sum(t._version for t in st), st[0].data_ptr() if st else 0,
                st[-1].data_ptr() if st else 0)

    def _table(self):
        return N.TensorTable(self._state())

    def _invalidate(self):
        object.__setattr__(self, "_nh_tensors", None)
        if self.__dict__.get("_nh") is not None:
            self._nh.close()
            object.__setattr__(self, "_nh", None)

    def _apply(self, fn, *a, **k):          # .cuda() / .to() / .float(): tensors are replaced
        r = super()._apply(fn, *a, **k)
        for m in self.modules():
            if isinstance(m, _HipModule):
                m._invalidate()
        return r

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        for m in self.modules():
            if isinstance(m, _HipModule):
                m._invalidate()
        return r

    def _dev(self):
        p = next(self.parameters())
        N.require_gpu(p)
        return p.device

    def _native(self):
        """Library handle (created on first use).  Call inside ``with torch.cuda.device(self._dev())``: the library
        allocates and launches on the CURRENT device."""
        N.require_gpu(next(self.parameters()))
        sig = self._sig()
        if self.__dict__.get("_nh") is None or self.__dict__.get("_nh_sig") != sig:
            if self.__dict__.get("_nh") is not None:
                self._nh.close()
            object.__setattr__(self, "_nh_tensors", None)
            sig = self._sig()
            h = N.Handle(self._destroy)
            tab = self._table()
            with torch.cuda.device(self._dev()):
                self._create(h, tab)
                torch.cuda.current_stream().synchronize()
            object.__setattr__(self, "_nh", h)
            object.__setattr__(self, "_nh_sig", sig)
        return self._nh.h

    def _create(self, handle, tab):
        raise NotImplementedError

    def _check_eval(self):
        if self.training and getattr(self, "p_dropout", 0):
            raise NotImplementedError("dropout (training mode) is outside the inference path; call .eval()")


def _g_args(g, T, B=None, gin=None):
    if g is None:
        return None, 0
    g = N.f32(g)
    if g.dim() != 3 or g.shape[2] not in (1, T):
        raise ValueError(f"g must be [B, gin, 1] or [B, gin, {T}], got {tuple(g.shape)}")
    if gin is not None:
        if gin == 0:
            raise N.SvocError("g given but the module was built with gin_channels == 0")
        _check_channels("g", g.shape[1], gin)
    if B is not None and g.shape[0] != B:
        if g.shape[0] != 1:
            raise ValueError(f"g has batch {g.shape[0]}, expected {B}")
        g = g.expand(B, g.shape[1], g.shape[2]).contiguous()
    return g, g.shape[2]


def _same_device(x, *others):
    for t in others:
        if t is not None and t.device != x.device:
            raise N.SvocError(f"tensors on different devices: {x.device} and {t.device}")


def _mask_arg(x_mask, B, T, device):
    m = N.f32(x_mask)
    if m.dim() != 3 or m.shape[1] != 1 or m.shape[2] != T:
        raise ValueError(f"x_mask must be [B, 1, {T}], got {tuple(m.shape)}")
    if m.shape[0] != B:
        m = m.expand(B, 1, T).contiguous()
    return m


# ----------------------------------------------------------------------------- LayerNorm / DDSConv
class LayerNorm(nn.Module):
    """Channel LayerNorm (reference modules.py:20-32).  Inside DDSConv it runs fused with the depthwise conv / GELU
    kernels; the standalone forward is the HIP kernel behind svoc_layer_norm."""

    def __init__(self, channels, eps=1e-5):
        super().__init__()
        self.channels = channels
        self.eps = eps
        self.gamma = nn.Parameter(torch.ones(channels))
        self.beta = nn.Parameter(torch.zeros(channels))

    def forward(self, x):
        # x.transpose(1, -1) -> F.layer_norm over the last dim -> transpose back == normalise dim 1 at every other index
        x = N.f32(x)
        if x.dim() < 2:
            raise ValueError("LayerNorm expects [B, C, ...]")
        _check_channels("LayerNorm", x.shape[1], self.channels)
        B, Cc = x.shape[0], x.shape[1]
        inner = x[0, 0].numel() if x.dim() > 2 else 1
        y = torch.empty_like(x)
        if x.numel() == 0:
            return y
        _same_device(x, self.gamma)
        with torch.cuda.device(x.device):
            N.check(N.lib().svoc_layer_norm(N.stream_ptr(x.device), N.ptr(x), N.ptr(N.f32(self.gamma.detach())),
                                            N.ptr(N.f32(self.beta.detach())), float(self.eps), N.ptr(y), B, Cc, inner))
        return y


class DDSConv(_HipModule):
    """Dilated and depth-separable convolution stack (reference modules.py:70-108)."""
    _destroy = "svoc_dds_destroy"

    def __init__(self, channels, kernel_size, n_layers, p_dropout=0.):
        super().__init__()
        self.channels = channels
        self.kernel_size = kernel_size
        self.n_layers = n_layers
        self.p_dropout = p_dropout
        self.convs_sep = nn.ModuleList()
        self.convs_1x1 = nn.ModuleList()
        self.norms_1 = nn.ModuleList()
        self.norms_2 = nn.ModuleList()
        for i in range(n_layers):
            self.convs_sep.append(_Conv1dParams(channels, channels, kernel_size, groups=channels))
            self.convs_1x1.append(_Conv1dParams(channels, channels, 1))
            self.norms_1.append(LayerNorm(channels))
            self.norms_2.append(LayerNorm(channels))

    def _create(self, h, tab):
        N.check(N.lib().svoc_dds_create(h.out(), self.channels, self.kernel_size, self.n_layers, tab.arr, tab.n, b""))

    def forward(self, x, x_mask, g=None):
        self._check_eval()
        x = N.f32(x)
        B, Cc, T = x.shape
        _check_channels("DDSConv", Cc, self.channels)
        m = _mask_arg(x_mask, B, T, x.device)
        g, gT = _g_args(g, T, B, self.channels)      # `x + g`: g is [B, C, T] or broadcast over time [B, C, 1]
        y = torch.empty_like(x)
        _same_device(x, m, g)
        with torch.cuda.device(x.device):
            N.check(N.lib().svoc_dds_forward(self._native(), N.stream_ptr(x.device), N.ptr(x), N.ptr(m), N.ptr(g), gT,
                                             N.ptr(y), B, T))
        return y


# ----------------------------------------------------------------------------- WN
class WN(_HipModule):
    """Gated (non-causal WaveNet) conv stack (reference modules.py:111-185)."""
    _destroy = "svoc_wn_destroy"

    def __init__(self, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=0, p_dropout=0):
        super().__init__()
        assert (kernel_size % 2 == 1)
        self.hidden_channels = hidden_channels
        self.kernel_size = kernel_size,
        self._k = kernel_size
        self.dilation_rate = dilation_rate
        self.n_layers = n_layers
        self.gin_channels = gin_channels
        self.p_dropout = p_dropout
        self.in_layers = nn.ModuleList()
        self.res_skip_layers = nn.ModuleList()
        if gin_channels != 0:
            self.cond_layer = _WNConv1dParams(gin_channels, 2 * hidden_channels * n_layers, 1)
        for i in range(n_layers):
            self.in_layers.append(_WNConv1dParams(hidden_channels, 2 * hidden_channels, kernel_size))
            rs = 2 * hidden_channels if i < n_layers - 1 else hidden_channels
            self.res_skip_layers.append(_WNConv1dParams(hidden_channels, rs, 1))

    def _create(self, h, tab):
        N.check(N.lib().svoc_wn_create(h.out(), self.hidden_channels, self._k, self.dilation_rate, self.n_layers,
                                       self.gin_channels, tab.arr, tab.n, b""))

    def forward(self, x, x_mask, g=None, **kwargs):
        self._check_eval()
        x = N.f32(x)
        B, H, T = x.shape
        _check_channels("WN", H, self.hidden_channels)
        m = _mask_arg(x_mask, B, T, x.device)
        g, gT = _g_args(g, T, B, self.gin_channels)
        out = torch.empty_like(x)
        _same_device(x, m, g)
        with torch.cuda.device(x.device):
            N.check(N.lib().svoc_wn_forward(self._native(), N.stream_ptr(x.device), N.ptr(x), N.ptr(m), N.ptr(g), gT,
                                            N.ptr(out), B, T))
        return out

    def remove_weight_norm(self):
        if self.gin_channels != 0:
            _fold_in_place(self.cond_layer)
        for l in self.in_layers:
            _fold_in_place(l)
        for l in self.res_skip_layers:
            _fold_in_place(l)


# ----------------------------------------------------------------------------- ResBlocks
class _ResBlockBase(_HipModule):
    _destroy = "svoc_resblock_destroy"
    _kind = 1

    def _create(self, h, tab):
        d = (C.c_int * len(self.dilation))(*self.dilation)
        N.check(N.lib().svoc_resblock_create(h.out(), self._kind, self.channels, self.kernel_size, d, len(self.dilation),
                                             tab.arr, tab.n, b""))

    def forward(self, x, x_mask=None):
        x = N.f32(x)
        B, Cc, L = x.shape
        _check_channels(type(self).__name__, Cc, self.channels)
        m = _mask_arg(x_mask, B, L, x.device) if x_mask is not None else None
        y = torch.empty_like(x)
        _same_device(x, m)
        with torch.cuda.device(x.device):
            N.check(N.lib().svoc_resblock_forward(self._native(), N.stream_ptr(x.device), N.ptr(x), N.ptr(m), N.ptr(y), B, L))
        return y


class ResBlock1(_ResBlockBase):
    """reference modules.py:187-229: 3 x {lrelu, dilated conv, lrelu, conv, + x}."""
    _kind = 1

    def __init__(self, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        self.channels, self.kernel_size, self.dilation = channels, kernel_size, tuple(dilation)
        self.convs1 = nn.ModuleList([_WNConv1dParams(channels, channels, kernel_size) for _ in self.dilation])
        self.convs2 = nn.ModuleList([_WNConv1dParams(channels, channels, kernel_size) for _ in self.dilation])

    def remove_weight_norm(self):
        for l in self.convs1:
            _fold_in_place(l)
        for l in self.convs2:
            _fold_in_place(l)


class ResBlock2(_ResBlockBase):
    """reference modules.py:232-256: 2 x {lrelu, dilated conv, + x}."""
    _kind = 2

    def __init__(self, channels, kernel_size=3, dilation=(1, 3)):
        super().__init__()
        self.channels, self.kernel_size, self.dilation = channels, kernel_size, tuple(dilation)
        self.convs = nn.ModuleList([_WNConv1dParams(channels, channels, kernel_size) for _ in self.dilation])

    def remove_weight_norm(self):
        for l in self.convs:
            _fold_in_place(l)


# ----------------------------------------------------------------------------- Flip / coupling
class Flip(nn.Module):
    """Channel reversal (reference modules.py:270-277).  Inside ResidualCouplingBlock the flips are folded into
    the neighbouring weights; the standalone module runs the HIP copy kernel."""

    def forward(self, x, *args, reverse=False, **kwargs):
        x = N.f32(x)
        B, Cc, T = x.shape
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            N.check(N.lib().svoc_flip_channels(N.stream_ptr(x.device), N.ptr(x), N.ptr(y), B, Cc, T))
        if not reverse:
            logdet = torch.zeros(B, dtype=x.dtype, device=x.device)
            return y, logdet
        return y


class ResidualCouplingLayer(_HipModule):
    """Affine coupling (reference modules.py:298-343)."""
    _destroy = "svoc_coupling_destroy"

    def __init__(self, channels, hidden_channels, kernel_size, dilation_rate, n_layers, p_dropout=0, gin_channels=0,
                 mean_only=False):
        assert channels % 2 == 0, "channels should be divisible by 2"
        super().__init__()
        self.channels = channels
        self.hidden_channels = hidden_channels
        self.kernel_size = kernel_size
        self.dilation_rate = dilation_rate
        self.n_layers = n_layers
        self.half_channels = channels // 2
        self.mean_only = mean_only
        self.gin_channels = gin_channels
        self.p_dropout = p_dropout
        self.pre = _Conv1dParams(self.half_channels, hidden_channels, 1)
        self.enc = WN(hidden_channels, kernel_size, dilation_rate, n_layers, p_dropout=p_dropout, gin_channels=gin_channels)
        self.post = _Conv1dParams(hidden_channels, self.half_channels * (2 - mean_only), 1)
        self.post.weight.data.zero_()
        self.post.bias.data.zero_()

    def _create(self, h, tab):
        N.check(N.lib().svoc_coupling_create(h.out(), self.channels, self.hidden_channels, self.kernel_size,
                                             self.dilation_rate, self.n_layers, self.gin_channels,
                                             1 if self.mean_only else 0, tab.arr, tab.n, b""))

    def forward(self, x, x_mask, g=None, reverse=False):
        self._check_eval()
        x = N.f32(x)
        B, Cc, T = x.shape
        _check_channels("ResidualCouplingLayer", Cc, self.channels)
        m = _mask_arg(x_mask, B, T, x.device)
        g, gT = _g_args(g, T, B, self.gin_channels)
        y = torch.empty_like(x)
        logdet = None if reverse else torch.empty(B, dtype=torch.float32, device=x.device)
        _same_device(x, m, g)
        with torch.cuda.device(x.device):
            N.check(N.lib().svoc_coupling_forward(self._native(), N.stream_ptr(x.device), N.ptr(x), N.ptr(m), N.ptr(g), gT,
                                                  1 if reverse else 0, N.ptr(y), N.ptr(logdet), B, T))
        return y if reverse else (y, logdet)


class ConvFlow(_HipModule):
    """Spline coupling flow (reference modules.py:346-390)."""
    _destroy = "svoc_convflow_destroy"

    def __init__(self, in_channels, filter_channels, kernel_size, n_layers, num_bins=10, tail_bound=5.0):
        super().__init__()
        self.in_channels = in_channels
        self.filter_channels = filter_channels
        self.kernel_size = kernel_size
        self.n_layers = n_layers
        self.num_bins = num_bins
        self.tail_bound = tail_bound
        self.half_channels = in_channels // 2
        self.pre = _Conv1dParams(self.half_channels, filter_channels, 1)
        self.convs = DDSConv(filter_channels, kernel_size, n_layers, p_dropout=0.)
        self.proj = _Conv1dParams(filter_channels, self.half_channels * (num_bins * 3 - 1), 1)
        self.proj.weight.data.zero_()
        self.proj.bias.data.zero_()

    def _create(self, h, tab):
        N.check(N.lib().svoc_convflow_create(h.out(), self.in_channels, self.filter_channels, self.kernel_size,
                                             self.n_layers, self.num_bins, float(self.tail_bound), tab.arr, tab.n, b""))

    def forward(self, x, x_mask, g=None, reverse=False):
        x = N.f32(x)
        B, Cc, T = x.shape
        _check_channels("ConvFlow", Cc, self.in_channels)
        m = _mask_arg(x_mask, B, T, x.device)
        # g goes to DDSConv unchanged (reference modules.py:366), i.e. it is added to pre(x0): [B, filter_channels, T or 1]
        g, gT = _g_args(g, T, B, self.filter_channels)
        y = torch.empty_like(x)
        logdet = None if reverse else torch.empty(B, dtype=torch.float32, device=x.device)
        _same_device(x, m, g)
        with torch.cuda.device(x.device):
            N.check(N.lib().svoc_convflow_forward(self._native(), N.stream_ptr(x.device), N.ptr(x), N.ptr(m), N.ptr(g), gT,
                                                  1 if reverse else 0, N.ptr(y), N.ptr(logdet), B, T))
        return y if reverse else (y, logdet)
