#!/usr/bin/env python3
"""fp32 error of candidate Winograd / Cook-Toom forms F(m, r) against the direct form (numpy, CPU): matrices built exactly (fractions)
from the interpolation points, checked for exactness in float64, then run in float32 on random windows.  Design input for the
choice of forms (DESIGN.md "Candidates for round 4"); not part of the product path.   python tools/wino_numerics.py"""
import numpy as np
from fractions import Fraction as F
def cook_toom(m, r, pts):
    # F(m,r): n = m+r-1 points (last = infinity). Returns AT (m x n), G (n x r), BT (n x n) in float64, correlation form.
    n = m + r - 1
    assert len(pts) == n - 1
    # polynomial-evaluation matrices
    def V(rows, cols):  # rows: points, evaluate poly of degree cols-1
        M = [[F(p) ** j for j in range(cols)] for p in pts]
        M.append([F(0)] * (cols - 1) + [F(1)])
        return M
    A = V(n, m)      # n x m
    G = V(n, r)      # n x r
    # scale G rows by 1/prod(p_i - p_j)
    for i, p in enumerate(pts):
        d = F(1)
        for j, q in enumerate(pts):
            if i != j: d *= (F(p) - F(q))
        G[i] = [g / d for g in G[i]]
    # B^T from interpolation: linear-convolution algorithm transposed -> correlation: Y = A^T [(G g) * (B^T d)]
    # B^T = inverse-transpose relation: rows are coefficients of M(x)/(x-p_i) etc.  Build by solving: for all d, g the identity holds.
    # Use the known construction: B^T[i] = coefficients of prod_{j!=i}(x - p_j) (degree n-2... padded), last row = prod_j (x - p_j)
    import numpy.polynomial.polynomial as P
    BT = []
    for i, p in enumerate(pts):
        c = [F(1)]
        for j, q in enumerate(pts):
            if i != j:
                c = [ (c[k-1] if k>0 else F(0)) - F(q) * (c[k] if k < len(c) else F(0)) for k in range(len(c)+1)]
        BT.append(c + [F(0)] * (n - len(c)))
    c = [F(1)]
    for q in pts:
        c = [ (c[k-1] if k>0 else F(0)) - F(q) * (c[k] if k < len(c) else F(0)) for k in range(len(c)+1)]
    BT.append(c)
    f = lambda M: np.array([[float(x) for x in row] for row in M])
    return f(A).T, f(G), f(BT)
def check(m, r, pts, N=200000, seed=0):
    AT, G, BT = cook_toom(m, r, pts)
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((N, m + r - 1)); g = rng.standard_normal((N, r)) / np.sqrt(r)
    ref = np.stack([(d[:, i:i + r] * g).sum(1) for i in range(m)], 1)
    y64 = ((g @ G.T) * (d @ BT.T)) @ AT.T
    e64 = np.abs(y64 - ref).max()
    d32, g32 = d.astype(np.float32), g.astype(np.float32)
    U = (g32 @ G.T.astype(np.float32)); Vv = (d32 @ BT.T.astype(np.float32))
    y32 = ((U * Vv) @ AT.T.astype(np.float32))
    dir32 = np.stack([(d32[:, i:i + r] * g32).sum(1, dtype=np.float32) for i in range(m)], 1)
    rms = lambda a: float(np.sqrt(np.mean(a.astype(np.float64) ** 2)))
    return e64, rms(y32 - ref) / rms(ref), rms(dir32 - ref) / rms(ref)
for name, m, r, pts in (("F(2,3)", 2, 3, [0, 1, -1]), ("F(4,3)", 4, 3, [0, 1, -1, 2, -2]), ("F(2,5)", 2, 5, [0, 1, -1, 2, -2]),
                        ("F(2,5) halves", 2, 5, [0, 1, -1, F(1, 2), F(-1, 2)]), ("F(4,5)", 4, 5, [0, 1, -1, 2, -2, F(1, 2), F(-1, 2)]),
                        ("F(6,3)", 6, 3, [0, 1, -1, 2, -2, F(1, 2), F(-1, 2)]),
                        # round 6: the k = 7 convolutions as ONE group (10 products per four outputs instead of two F(4,4) groups' 14) - the best of five point sets tried
                        ("F(4,4)", 4, 4, [0, 1, -1, 2, -2, F(1, 2)]), ("F(4,7)", 4, 7, [0, 1, -1, F(1, 2), F(-1, 2), 2, -2, F(3, 4), F(-3, 4)])):
    e64, w, dr = check(m, r, pts)
    print(f"{name:14s} exactness (f64 max err) {e64:.1e}   fp32 rel-RMS: winograd {w:.2e}   direct {dr:.2e}   ratio {w / dr:.1f}")


def conv_level(m, r, pts, C=192, Co=64, T=512, seed=1):
    """One convolution Co x C x r over T columns ('same' padding) in F(m, r) form with fp32 transforms, products and channel
    accumulation, against the fp32 direct form; reference = float64 direct."""
    AT, G, BT = cook_toom(m, r, pts)
    n = m + r - 1
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((C, T)); w = rng.standard_normal((Co, C, r)) / np.sqrt(C * r)
    pad = (r - 1) // 2
    Tp = ((T + m - 1) // m) * m
    xp = np.zeros((C, Tp + r - 1)); xp[:, pad:pad + T] = x
    ref = np.zeros((Co, Tp))
    for t in range(r):
        ref += np.einsum("oc,ct->ot", w[:, :, t], xp[:, t:t + Tp])
    x32, w32 = xp.astype(np.float32), w.astype(np.float32)
    d32 = np.zeros((Co, Tp), np.float32)
    for t in range(r):
        d32 += np.einsum("oc,ct->ot", w32[:, :, t], x32[:, t:t + Tp]).astype(np.float32)
    nw = Tp // m
    win = np.stack([x32[:, m * q:m * q + n] for q in range(nw)], 1)            # [C, nw, n]
    V = np.einsum("pj,cqj->pcq", BT.astype(np.float32), win).astype(np.float32)   # [n, C, nw]
    U = np.einsum("pj,ocj->poc", G.astype(np.float32), w32).astype(np.float32)   # [n, Co, C]
    M = np.einsum("poc,pcq->poq", U, V).astype(np.float32)                        # the GEMMs, fp32 accumulation
    y = np.einsum("ip,poq->oqi", AT.astype(np.float32), M).astype(np.float32).reshape(Co, Tp)
    rms = lambda a: float(np.sqrt(np.mean(a.astype(np.float64) ** 2)))
    return rms(y - ref) / rms(ref), rms(d32 - ref) / rms(ref)


print("\none convolution, C = 192 input channels, 64 output rows, 512 columns (fp32 transforms / products / accumulation):")
for name, m, r, pts in (("F(4,3)", 4, 3, [0, 1, -1, 2, -2]), ("F(2,5)", 2, 5, [0, 1, -1, 2, -2]),
                        ("F(4,5)", 4, 5, [0, 1, -1, 2, -2, F(1, 2), F(-1, 2)]), ("F(6,3)", 6, 3, [0, 1, -1, 2, -2, F(1, 2), F(-1, 2)]),
                        ("F(4,4)", 4, 4, [0, 1, -1, 2, -2, F(1, 2)]), ("F(4,7)", 4, 7, [0, 1, -1, F(1, 2), F(-1, 2), 2, -2, F(3, 4), F(-3, 4)])):
    w_, d_ = conv_level(m, r, pts)
    print(f"{name:14s} fp32 rel-RMS: winograd {w_:.2e}   direct {d_:.2e}   ratio {w_ / d_:.1f}")
