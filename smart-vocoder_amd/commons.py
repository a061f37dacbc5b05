"""Helpers of the inference path with the reference's names (reference commons.py).

Only the functions the hot path uses are provided; the
training/TTS-only helpers of the reference's commons.py (KL, gumbel, slicing,
timing signals, generate_path, clip_grad_value_) are out of scope.
"""
import torch

try:
    from . import _native as N
except ImportError:  # imported as a top-level module (notebook-style: sys.path points at this directory)
    import _native as N


def init_weights(m, mean=0.0, std=0.01):
    """reference commons.py:8-11.  Under weight-norm the reference's call is a silent no-op
    (it writes the derived `.weight`), so this only touches modules owning a plain `weight`."""
    if m.__class__.__name__.find("Conv") != -1 and hasattr(m, "weight") and isinstance(m.weight, torch.nn.Parameter):
        m.weight.data.normal_(mean, std)


def get_padding(kernel_size, dilation=1):
    """reference commons.py:14-15"""
    return int((kernel_size * dilation - dilation) / 2)


def sequence_mask(length, max_length=None):
    """reference commons.py:121-125: bool [B, max_length]; computed by the HIP kernel svoc_sequence_mask."""
    N.require_gpu(length)
    if max_length is None:
        max_length = int(length.max().item())
    ln = length.to(torch.int64).contiguous()
    out = torch.empty(ln.shape[0], int(max_length), dtype=torch.float32, device=ln.device)
    with torch.cuda.device(ln.device):
        N.check(N.lib().svoc_sequence_mask(N.stream_ptr(ln.device), N.ptr(ln), N.ptr(out), ln.shape[0], int(max_length)))
    return out.bool()


def fused_add_tanh_sigmoid_multiply(input_a, input_b, n_channels):
    """reference commons.py:100-107: tanh((a+b)[:, :n]) * sigmoid((a+b)[:, n:]); n_channels is an IntTensor([n])."""
    n = int(n_channels[0])
    a, b = N.f32(input_a), N.f32(input_b)
    if a.shape != b.shape or a.dim() != 3 or a.shape[1] != 2 * n:
        raise ValueError(f"expected two [B, {2 * n}, T] tensors, got {tuple(a.shape)} and {tuple(b.shape)}")
    out = torch.empty(a.shape[0], n, a.shape[2], dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        N.check(N.lib().svoc_fused_add_tanh_sigmoid_multiply(N.stream_ptr(a.device), N.ptr(a), N.ptr(b), N.ptr(out),
                                                             a.shape[0], n, a.shape[2]))
    return out
