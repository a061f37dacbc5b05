"""Piecewise rational-quadratic spline transform with the reference's entry point
(reference transforms.py:12-44), computed by the HIP kernel behind svoc_rq_spline."""
import torch

try:
    from . import _native as N
except ImportError:
    import _native as N

DEFAULT_MIN_BIN_WIDTH = 1e-3
DEFAULT_MIN_BIN_HEIGHT = 1e-3
DEFAULT_MIN_DERIVATIVE = 1e-3


def piecewise_rational_quadratic_transform(inputs, unnormalized_widths, unnormalized_heights,
                                           unnormalized_derivatives, inverse=False, tails=None, tail_bound=1.,
                                           min_bin_width=DEFAULT_MIN_BIN_WIDTH,
                                           min_bin_height=DEFAULT_MIN_BIN_HEIGHT,
                                           min_derivative=DEFAULT_MIN_DERIVATIVE):
    """inputs [...]; widths/heights [..., bins]; derivatives [..., bins-1] (tails='linear') or [..., bins+1].
    Returns (outputs, logabsdet), both shaped like inputs."""
    if tails not in (None, "linear"):
        raise RuntimeError("{} tails are not implemented.".format(tails))
    x = N.f32(inputs)
    nb = unnormalized_widths.shape[-1]
    if min_bin_width * nb > 1.0:           # reference transforms.py:110-113
        raise ValueError('Minimal bin width too large for the number of bins')
    if min_bin_height * nb > 1.0:
        raise ValueError('Minimal bin height too large for the number of bins')
    uw = N.f32(unnormalized_widths).reshape(-1, nb)
    uh = N.f32(unnormalized_heights).reshape(-1, nb)
    nd = nb - 1 if tails == "linear" else nb + 1
    if unnormalized_derivatives.shape[-1] != nd:
        raise ValueError(f"expected {nd} derivatives per element, got {unnormalized_derivatives.shape[-1]}")
    ud = N.f32(unnormalized_derivatives).reshape(-1, nd)
    n = x.numel()
    if tails is None and n and (float(x.min()) < 0.0 or float(x.max()) > 1.0):
        raise ValueError("Input to a transform is not within its domain")
    out = torch.empty_like(x)
    lad = torch.empty_like(x)
    with torch.cuda.device(x.device):
        N.check(N.lib().svoc_rq_spline(N.stream_ptr(x.device), N.ptr(x), N.ptr(uw), N.ptr(uh), N.ptr(ud), n, nb,
                                       1 if inverse else 0, 1 if tails == "linear" else 0, float(tail_bound),
                                       float(min_bin_width), float(min_bin_height), float(min_derivative),
                                       N.ptr(out), N.ptr(lad)))
    return out, lad
