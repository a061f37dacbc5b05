#!/usr/bin/env python3
"""Numerics of a Winograd F(4,5) form for the WN in_layer (k = 5, 192 -> 384 channels) against the F(2,5) form the kernels use, in fp32 on the CPU
(numpy): error of one layer's pre-activation against an fp64 direct convolution.  A pricing of DESIGN.md section 5's open item (1), not product code.
    python tools/wino_f45_error.py"""
import numpy as np
from fractions import Fraction as Fr

def cook_toom(m, r, pts):
    """F(m, r) with finite points `pts` (m + r - 2 of them) + infinity: A^T (m x n), G (n x r), B^T (n x n), n = m + r - 1.  A and G are the evaluation
    (Vandermonde) matrices; B^T is SOLVED exactly from the bilinear identity  sum_p A^T[i,p] G[p,j] B^T[p,q] = [q == i + j]."""
    import sympy as sp
    n = m + r - 1
    assert len(pts) == n - 1
    R = lambda v: sp.Rational(Fr(v).numerator, Fr(v).denominator)
    A = sp.Matrix([[R(x) ** k for k in range(m)] for x in pts] + [[0] * (m - 1) + [1]])          # n x m
    G = sp.Matrix([[R(x) ** k for k in range(r)] for x in pts] + [[0] * (r - 1) + [1]])          # n x r
    K = sp.Matrix([[A[p_, i] * G[p_, j] for p_ in range(n)] for i in range(m) for j in range(r)])      # (m r) x n
    BT = sp.zeros(n, n)
    for q in range(n):
        rhs = sp.Matrix([1 if i + j == q else 0 for i in range(m) for j in range(r)])
        sol = sp.linsolve((K, rhs))
        b = list(sol)[0]
        for p_ in range(n): BT[p_, q] = b[p_]
    f = lambda M_: np.array(M_.tolist(), dtype=np.float64)
    return f(A).T, f(G), f(BT)

def check(m, r, pts):
    AT, G, BT = cook_toom(m, r, pts)
    rng = np.random.default_rng(0)
    d = rng.standard_normal(m + r - 1); g = rng.standard_normal(r)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(d[i + j] * g[j] for j in range(r)) for i in range(m)])
    assert np.allclose(y, ref, atol=1e-9), (y, ref)
    return AT, G, BT

def layer_error(AT, G, BT, m, r, C=192, Co=64, T=512, seed=1):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((C, T + r - 1)) * 0.7                    # WN activations: O(1)
    w = rng.standard_normal((Co, C, r)) / np.sqrt(C * r)             # weight-normed rows
    ref = np.zeros((Co, T))
    for j in range(r): ref += w[:, :, j] @ x[:, j:j + T]
    n = m + r - 1
    x32 = x.astype(np.float32); U = np.einsum('pk,ock->ocp', G, w).astype(np.float32)          # filter transform exact in fp64, stored fp32 (as the images are)
    BT32 = BT.astype(np.float32); AT32 = AT.astype(np.float32)
    out = np.zeros((Co, T), dtype=np.float32)
    for t0 in range(0, T, m):
        dwin = x32[:, t0:t0 + n]                                      # [C][n]
        V = (dwin @ BT32.T).astype(np.float32)                        # [C][n]
        Mm = np.einsum('ocp,cp->op', U, V, dtype=np.float32)          # fp32 accumulate over channels
        out[:, t0:t0 + m] = (Mm @ AT32.T).astype(np.float32)[:, :min(m, T - t0)]
    err = out.astype(np.float64) - ref
    return float(np.sqrt((err ** 2).mean())), float(np.abs(err).max()), float(np.sqrt((ref ** 2).mean()))

def direct32(C=192, Co=64, T=512, r=5, seed=1):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((C, T + r - 1)) * 0.7; w = rng.standard_normal((Co, C, r)) / np.sqrt(C * r)
    ref = np.zeros((Co, T)); out = np.zeros((Co, T), dtype=np.float32)
    for j in range(r):
        ref += w[:, :, j] @ x[:, j:j + T]
        out += (w[:, :, j].astype(np.float32) @ x[:, j:j + T].astype(np.float32))
    err = out.astype(np.float64) - ref
    return float(np.sqrt((err ** 2).mean())), float(np.abs(err).max())

if __name__ == "__main__":
    forms = {"F(2,5) points 0, +-1, +-2 (the kernels' form)": (2, 5, [0, 1, -1, 2, -2]),
             "F(4,5) points 0, +-1, +-2, +-1/2": (4, 5, [0, 1, -1, 2, -2, Fr(1, 2), Fr(-1, 2)]),
             "F(4,5) points 0, +-1, +-1/2, +-2 reordered (same set)": (4, 5, [0, 1, -1, Fr(1, 2), Fr(-1, 2), 2, -2]),
             "F(4,5) points 0, +-1, +-1/2, +-3/2": (4, 5, [0, 1, -1, Fr(1, 2), Fr(-1, 2), Fr(3, 2), Fr(-3, 2)])}
    e, mx = direct32()
    print(f"direct form, fp32 accumulate:            rms {e:.3e}  max {mx:.3e}")
    for name, (m, r, pts) in forms.items():
        AT, G, BT = check(m, r, pts)
        e, mx, ref = layer_error(AT, G, BT, m, r)
        print(f"{name:55s} rms {e:.3e}  max {mx:.3e}  (ref rms {ref:.3f}; products per output x tap: {(m + r - 1) / (m * r):.2f})   max |B^T| {np.abs(BT).max():.2f} max |A^T| {np.abs(AT).max():.2f}")
