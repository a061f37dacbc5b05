"""CPU checks of the Winograd algebra the HIP kernels implement (csrc/conv_wino4.h): the F(4,3) and F(4,4) transform matrices as
written in the header, the grouping of a k-tap filter into three- / four-tap groups on SHARED transformed planes (window step = group
spacing = 4), the left-over taps of the F(4,3) form, the polyphase treatment of dilation and the window-major row order a dilated
convolution hands to the undilated one behind it - restated in numpy in float64 and compared with the direct convolution
(reference modules.py:190-207: Conv1d(k, dilation d, padding (k - 1) d / 2)).  No GPU: this pins the index algebra and the constants;
the kernels themselves are compared with torch / the oracle in tests/test_gpu_parity.py."""
import numpy as np
import pytest


def direct(x, w, D):
    """y[n] = sum_j w[j] x[n + (j - (k - 1) / 2) D], zero outside [0, L)"""
    C, L = x.shape
    k = w.shape[-1]
    pad = (k - 1) // 2 * D
    xp = np.zeros((C, L + 2 * pad))
    xp[:, pad:pad + L] = x
    y = np.zeros((w.shape[0], L))
    for j in range(k):
        y += w[:, :, j] @ xp[:, j * D:j * D + L]
    return y


# ---- the matrices of conv_wino4.h (input transform V = BT d, weights U = G w, outputs y = AT M)
def bt43(d):
    d0, d1, d2, d3, d4, d5 = d
    return np.stack([4 * d0 - 5 * d2 + d4, -4 * d1 - 4 * d2 + d3 + d4, 4 * d1 - 4 * d2 - d3 + d4, -2 * d1 - d2 + 2 * d3 + d4,
                     2 * d1 - d2 - 2 * d3 + d4, 4 * d1 - 5 * d3 + d5])


def g43(w0, w1, w2):
    return np.stack([w0 / 4, -(w0 + w1 + w2) / 6, -(w0 - w1 + w2) / 6, (w0 + 2 * w1 + 4 * w2) / 24, (w0 - 2 * w1 + 4 * w2) / 24, w2])


def at43(M):
    return np.stack([M[0] + M[1] + M[2] + M[3] + M[4], M[1] - M[2] + 2 * M[3] - 2 * M[4], M[1] + M[2] + 4 * M[3] + 4 * M[4],
                     M[1] - M[2] + 8 * M[3] - 8 * M[4] + M[5]])


def bt44(d):
    d0, d1, d2, d3, d4, d5, d6 = d
    e1, o1 = d2 + d6 - 4.25 * d4, d1 + d5 - 4.25 * d3
    e2, o2 = d6 + 0.25 * d2 - 1.25 * d4, 0.5 * d1 - 2.5 * d3 + 2 * d5
    e3, o3 = d6 + 4 * d2 - 5 * d4, 2 * d1 - 2.5 * d3 + 0.5 * d5
    return np.stack([(d0 - d6) + 5.25 * (d4 - d2), e1 + o1, e1 - o1, e2 + o2, e2 - o2, e3 + o3, e3 - o3])


def g44(w0, w1, w2, w3):
    return np.stack([w0, -((w0 + w2) + (w1 + w3)) * (2 / 9), -((w0 + w2) - (w1 + w3)) * (2 / 9), ((w0 + 4 * w2) + (2 * w1 + 8 * w3)) / 90,
                     ((w0 + 4 * w2) - (2 * w1 + 8 * w3)) / 90, ((w0 + w2 / 4) + (w1 / 2 + w3 / 8)) * (32 / 45),
                     ((w0 + w2 / 4) - (w1 / 2 + w3 / 8)) * (32 / 45)])


def at44(M):
    a = np.array([0, 1, -1, 2, -2, 0.5, -0.5])
    return np.stack([sum((a[p] ** i if (a[p] != 0 or i > 0) else 1.0) * M[p] for p in range(7)) for i in range(4)])


def winograd(x, w, D, f44):
    """The kernels' decomposition.  Window w = D b + ph owns the outputs n = 4 D b + ph + r D, r = 0..3; its group-g inputs are
    d_j = x[n0 + (4 g + j - pad) D], n0 = 4 D b + ph.  Every group reads the SAME planes V_p[c][window + g D]."""
    C, L = x.shape
    Co, _, k = w.shape
    G, pad = (k + 1) // 4, (k - 1) // 2
    nwin = D * ((L + 4 * D - 1) // (4 * D))
    NV = 7 if f44 else 6
    ext = nwin + (G - 1) * D                                 # windows whose planes are needed (the tile's halo)
    lo, span = pad * D, (4 * (nwin // D + G) + NV) * D + L
    xp = np.zeros((C, lo + span + 8 * D))
    xp[:, lo:lo + L] = x
    first = lambda wi: 4 * D * (wi // D) + wi % D            # column of the window's output 0
    d = np.stack([np.stack([xp[:, lo + first(wi) + (j - pad) * D] for wi in range(ext)], 1) for j in range(NV)])     # [NV][C][ext]
    V = (bt44 if f44 else bt43)(d)                           # planes [NV][C][ext]
    M = np.zeros((NV if f44 else 8, Co, nwin))
    for g in range(G):
        taps = [w[:, :, 4 * g + t] if 4 * g + t < k else np.zeros_like(w[:, :, 0]) for t in range(4)]
        U = g44(*taps) if f44 else g43(*taps[:3])            # [NV][Co][C]
        for p in range(NV):
            M[p] += U[p] @ V[p][:, g * D:g * D + nwin]
    if not f44:
        # left-over taps 3, 7: w * x[n + (4 t + 3 - pad) D] for output r of the window = sample d_{1 + (r + 2) % 4} of window + t + (r + 2) // 4
        # (the X planes); r = 0 lands in M0, r = 3 in M5, r = 1 / 2 in accumulators of their own (6, 7)
        X = d[1:5]
        for t in range(G - 1):
            wt = w[:, :, 4 * t + 3]
            for r, acc in ((0, 0), (1, 6), (2, 7), (3, 5)):
                sh = (t + (r + 2) // 4) * D
                M[acc] += wt @ X[(r + 2) % 4][:, sh:sh + nwin] if sh + nwin <= ext else 0
        y4 = at43(M[:6])
        y4[1] += M[6]; y4[2] += M[7]
    else:
        y4 = at44(M)
    y = np.zeros((Co, L))
    for wi in range(nwin):
        for r in range(4):
            n = first(wi) + r * D
            if n < L:
                y[:, n] = y4[r][:, wi]
    return y


@pytest.mark.parametrize("k", [3, 7, 11])
@pytest.mark.parametrize("D", [1, 3, 5])
@pytest.mark.parametrize("L", [4, 37, 128, 257])
def test_f43_and_f44_equal_the_direct_convolution(k, D, L):
    rng = np.random.default_rng(100 * k + 10 * D + L)
    x = rng.standard_normal((5, L)); w = rng.standard_normal((3, 5, k))
    ref = direct(x, w, D)
    assert np.abs(winograd(x, w, D, False) - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())
    assert np.abs(winograd(x, w, D, True) - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())      # k = 3: a zero fourth tap (the merged accumulate launch)


def test_product_counts():
    """products per window (four outputs) and channel pair: what svoc_stats_executed_flops books per form"""
    for k, f43, f44 in ((3, 6, 7), (7, 16, 14), (11, 26, 21)):
        G = (k + 1) // 4
        assert 6 * G + 4 * (G - 1) == f43 and 7 * G == f44
        assert abs(f43 / 4 / k - (1.5 * G + (G - 1)) / k) < 1e-15 and abs(f44 / 4 / k - 1.75 * G / k) < 1e-15
    assert (26 + 16 + 6, 21 + 14 + 6, 21 + 14 + 7) == (48, 41, 42)      # grouped launch F(4,3) / F(4,4); merged accumulate launch F(4,4)


def test_f44_points_and_fp32_error():
    """The seven points 0, +-1, +-2, +-1/2: AT is their Vandermonde matrix, G / BT the Lagrange factors; in float32 one convolution
    (C = 192) stays within 1.5e-6 relative RMS of float64 (F(4,3): 1e-6; the waveform budget of the path is 1e-4)."""
    rng = np.random.default_rng(7)
    d = rng.standard_normal((7, 1000)); w4 = rng.standard_normal((4, 1000))
    ref = np.stack([sum(w4[j] * d[i + j] for j in range(4)) for i in range(4)])
    y = at44(g44(*w4) * bt44(d))
    assert np.abs(y - ref).max() < 1e-12
    C, Co, T = 192, 16, 512
    x = rng.standard_normal((C, T)); w = rng.standard_normal((Co, C, 11)) / np.sqrt(C * 11)
    ref = direct(x, w, 1)
    x32, w32 = x.astype(np.float32), w.astype(np.float32)
    G, pad, nwin = 3, 5, T // 4
    xp = np.zeros((C, T + 32), np.float32); xp[:, pad:pad + T] = x32
    dd = np.stack([np.stack([xp[:, 4 * q + j] for q in range(nwin + 2)], 1) for j in range(7)])
    V = bt44(dd).astype(np.float32)
    M = np.zeros((7, Co, nwin), np.float32)
    for g in range(G):
        taps = [w32[:, :, 4 * g + t] if 4 * g + t < 11 else np.zeros_like(w32[:, :, 0]) for t in range(4)]
        U = g44(*taps).astype(np.float32)
        for p in range(7):
            M[p] += (U[p] @ V[p][:, g:g + nwin]).astype(np.float32)
    y4 = at44(M.astype(np.float32)).astype(np.float32)
    y = np.stack([y4[r] for r in range(4)], -1).reshape(Co, T)
    rel = np.sqrt(np.mean((y - ref) ** 2) / np.mean(ref ** 2))
    assert rel < 1.5e-6, rel


@pytest.mark.parametrize("P", [3, 5])
def test_window_major_row_order(P):
    """A dilation-P convolution writes row[4 w + r] = y[4 P b + ph + r P] (w = P b + ph): one 16-byte store per lane; the undilated
    convolution behind it reads through the same map.  The map is a permutation inside each q block of 4 P columns."""
    L = 4 * P * 7
    n = np.arange(L)
    b, rem = n // (4 * P), n % (4 * P)
    r, ph = rem // P, rem % P
    pos = 4 * (P * b + ph) + r
    assert sorted(pos) == list(range(L))
    assert all(pos // (4 * P) == b)
