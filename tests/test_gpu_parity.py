"""GPU parity tests: the HIP path (called through the C ABI via the reference-named modules) against
the CPU oracle on identical seeded inputs, and against the golden fixtures produced by the reference.

Tolerances (fp32 path; the reference computes with oneDNN/ATen fp32 in a different summation order):
  * per-op / per-module: max|err| <= 2e-5 + 1e-4 * max|ref|
  * full infer waveform (north_star): RMS error <= 1e-3, and relative RMS <= 1e-4
  * spline: 1e-3 max-abs (fp32 conditioning of the inverse), bin-edge flips tolerated
"""
import os
import time

import numpy as np
import pytest
import torch

import cases
from cases import sw
from oracle import vocoder_oracle as O

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def sdT(d):
    return {k: T(v) for k, v in d.items()}


def load(module, sd_np):
    module.load_state_dict({k: T(v) for k, v in sd_np.items()})
    return module.cuda().eval()


def check(name, got, ref, atol=2e-5, rtol=1e-4):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    ref = ref.detach().cpu().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert np.isfinite(got).all(), f"{name}: non-finite output"
    err = np.abs(got - ref)
    bound = atol + rtol * float(np.abs(ref).max())
    idx = np.unravel_index(int(err.argmax()), err.shape) if err.size else ()
    assert float(err.max()) <= bound, f"{name}: max|err| {err.max():.3e} > {bound:.3e} at {idx} (got {got[idx]:.6f} ref {ref[idx]:.6f}); rms {np.sqrt((err**2).mean()):.3e}"


@pytest.fixture(scope="module")
def M():
    from smart_vocoder_amd import models, modules, commons, transforms, _native
    assert torch.cuda.is_available()
    _native.lib()
    class NS: pass
    ns = NS()
    ns.models, ns.modules, ns.commons, ns.transforms, ns.native = models, modules, commons, transforms, _native
    return ns


# ----------------------------------------------------------------------------- single ops
def test_sequence_mask_and_gate_and_flip(M):
    ln = torch.tensor([5, 0, 9, 3], dtype=torch.int64).cuda()
    m = M.commons.sequence_mask(ln, 9).cpu()
    assert torch.equal(m, torch.arange(9)[None, :] < ln.cpu()[:, None])
    a = T(cases.rnd(1, "a", (2, 128, 37), 2.0)); b = T(cases.rnd(1, "b", (2, 128, 37), 1.0))
    y = M.commons.fused_add_tanh_sigmoid_multiply(a.cuda(), b.cuda(), torch.IntTensor([64]))
    check("gate", y, O.gate(a, b, 64), 2e-6, 0)
    # the gate uses the hardware exp2 / rcp: saturation, zeros and large arguments must stay finite and within 2e-6
    ext = torch.tensor([-200.0, -90.0, -20.0, -1e-3, 0.0, 1e-3, 20.0, 90.0, 200.0])
    a2 = ext.repeat(128, 1).reshape(1, 128, 9).contiguous(); b2 = torch.flip(a2, [2]).contiguous() * 0.5
    y2 = M.commons.fused_add_tanh_sigmoid_multiply(a2.cuda(), b2.cuda(), torch.IntTensor([64]))
    assert torch.isfinite(y2).all()
    check("gate extremes", y2, O.gate(a2, b2, 64), 2e-6, 0)
    x = T(cases.rnd(2, "x", (2, 6, 11)))
    y, ld = M.modules.Flip()(x.cuda())
    assert torch.equal(y.cpu(), torch.flip(x, [1])) and float(ld.abs().sum()) == 0


@pytest.mark.parametrize("C,k,d,L,B,res", [(32, 3, 1, 700, 2, True), (32, 11, 5, 520, 1, False), (64, 7, 3, 300, 2, True),
                                            (128, 3, 5, 200, 2, True), (256, 11, 1, 150, 1, True), (192, 5, 1, 37, 3, False),
                                            (80, 1, 1, 50, 2, False), (96, 1, 1, 33, 2, False), (48, 7, 1, 1, 1, False)])
def test_conv1d(M, C, k, d, L, B, res):
    """lrelu -> weight-normed dilated Conv1d (+ residual): every (C,k,d) family of the decoder plus odd shapes."""
    import ctypes
    seed = 100 + C + 7 * k + d
    co = C if res else (C // 2 if C % 64 == 0 else C + 16)
    v = T(cases.rnd(seed, "v", (co, C, k), 1.0 / np.sqrt(C * k)))
    g = T((0.5 + sw.uniform01(seed, "g", co)).astype(np.float32)).reshape(co, 1, 1)
    bias = T(cases.rnd(seed, "b", (co,), 0.1))
    x = T(cases.rnd(seed, "x", (B, C, L), 1.0))
    w = O.fold_weight_norm(v, g)
    ref = torch.nn.functional.conv1d(torch.nn.functional.leaky_relu(x, 0.1), w, bias, dilation=d, padding=(k * d - d) // 2)
    if res:
        ref = ref + x
    N = M.native
    xc, vc, gc, bc = x.cuda(), v.cuda(), g.cuda(), bias.cuda()
    y = torch.empty(B, co, L, device="cuda")
    N.check(N.lib().svoc_conv1d(N.stream_ptr(), N.ptr(xc), N.ptr(vc), N.ptr(gc), N.ptr(bc), N.ptr(xc) if res else None,
                                N.ptr(y), B, C, co, L, k, d, ctypes.c_float(0.1)))
    check(f"conv1d C{C} k{k} d{d}", y, ref)


def _wino_f4_expected(C, co, k, d, L):
    """Which Winograd form the library documents for a shape: F(4,3) needs Cout as 64- / 128-row blocks with an even
    number of 32-channel chunks (or the single 32 x 32 block of the last MRF stage); everything else the Winograd entry point takes
    (odd row-block counts such as Cout = 96 / 160, odd chunk counts such as Cin = 192) runs the F(2,3) kernels.  (Both forms need
    rows of a multiple of four floats: other lengths are refused by the entry point and run in direct form inside the library.)"""
    import os
    if os.environ.get("SVOC_WINO_F4") == "0":
        return False
    mt, nch = (co + 31) // 32, (C + 31) // 32
    blocks = (mt % 2 == 0 and nch % 2 == 0) or (mt == 1 and nch == 1 and os.environ.get("SVOC_W4_C32") != "0")
    return blocks and (d != 1 or L % 4 == 0)


def _wino_f44_expected(C, co, k, d, L):
    """F(4,4) (conv_wino4.h, round 4): every shape the F(4,3) kernels take runs k = 7 / 11 as four-tap groups in seven products each,
    no left-over taps; k = 3 stays F(4,3)."""
    import os
    return _wino_f4_expected(C, co, k, d, L) and k >= 7


@pytest.mark.parametrize("C,co,k,d,L,B,res", [(128, 128, 3, 1, 4096, 2, True), (128, 128, 7, 1, 1000, 3, True), (128, 128, 11, 1, 4100, 1, False),
                                               (256, 256, 11, 1, 516, 2, True), (64, 96, 7, 1, 260, 2, False), (256, 256, 3, 1, 128, 1, True),
                                               (128, 128, 11, 1, 12, 1, True), (128, 128, 3, 3, 1000, 2, True), (128, 128, 7, 3, 4096, 1, True),
                                               (128, 128, 11, 3, 756, 2, True), (256, 256, 3, 5, 1204, 1, False), (128, 128, 7, 5, 4100, 2, True),
                                               (128, 128, 11, 5, 2400, 2, True), (64, 96, 11, 5, 40, 1, False), (128, 128, 11, 3, 8, 1, True),
                                               (64, 64, 11, 5, 1000, 2, True), (64, 64, 7, 3, 700, 1, True), (64, 64, 3, 1, 332, 2, True),
                                               (128, 128, 7, 1, 1004, 1, True), (64, 64, 11, 1, 2048, 1, False),
                                               (192, 64, 7, 1, 1000, 2, False), (64, 160, 3, 3, 600, 1, False),
                                               (32, 32, 3, 1, 4096, 2, True), (32, 32, 7, 1, 1000, 3, True), (32, 32, 11, 1, 4100, 1, False),
                                               (32, 32, 3, 3, 1500, 2, True), (32, 32, 7, 3, 4096, 1, True), (32, 32, 11, 3, 756, 2, True),
                                               (32, 32, 3, 5, 1204, 1, False), (32, 32, 7, 5, 4100, 2, True), (32, 32, 11, 5, 2400, 2, True),
                                               (32, 32, 11, 5, 40, 1, True), (32, 32, 11, 1, 12, 1, True), (32, 32, 7, 1, 516, 2, True),
                                               (256, 256, 7, 1, 4, 1, True), (256, 256, 7, 5, 8, 2, True), (128, 256, 11, 1, 132, 2, False),
                                               (256, 128, 7, 3, 1996, 1, False), (128, 128, 7, 5, 44, 3, False)])
def test_conv1d_winograd(M, C, co, k, d, L, B, res):
    """lrelu -> Conv1d(k, dilation d) [+ residual] through the Winograd entry point against torch's direct convolution, AND which
    form ran, read from the executed-multiply-add counter: F(4,3) (conv_wino4.hip, the default: three-tap groups at tap offsets
    0/4/8 on shared transformed planes + left-over taps 3/7 on the de-interleaved planes; dilation through the polyphase view)
    issues (1.5, 4, 6.5)/k of the direct form's multiply-adds for k = 3/7/11; F(4,4) (the default for k = 7/11: four-tap groups in
    seven products, no left-over taps) issues (3.5, 5.25)/k; the documented fall-back F(2,3) (conv_wino.hip: odd row-block or chunk
    counts, SVOC_WINO_F4=0) issues (2, 5, 8)/k.
    Shapes: every (k, d) of the model for the 128- / 64-row blocks and for the single 32-row block of the last MRF stage (C = 32,
    round 4: 1 x 4 consumers), ragged last tiles, inputs shorter than the halo, odd row-block counts (Cout = 96), lanes whose
    later outputs fall beyond the end (dilated tiles)."""
    import ctypes, os
    if os.environ.get("SVOC_WINO") == "0":
        pytest.skip("SVOC_WINO=0: the Winograd entry point refuses by design (variant run of the direct-form fallback)")
    if C == 32 and (os.environ.get("SVOC_WINO_F4") == "0" or os.environ.get("SVOC_W4_C32") == "0"):
        pytest.skip("C = 32 exists in F(4,3) form only")
    seed = 300 + C + 7 * k + L + 1000 * d
    v = T(cases.rnd(seed, "v", (co, C, k), 1.0 / np.sqrt(C * k)))
    g = T((0.5 + sw.uniform01(seed, "g", co)).astype(np.float32)).reshape(co, 1, 1)
    bias = T(cases.rnd(seed, "b", (co,), 0.1))
    x = T(cases.rnd(seed, "x", (B, C, L), 1.0))
    w = O.fold_weight_norm(v, g)
    ref = torch.nn.functional.conv1d(torch.nn.functional.leaky_relu(x, 0.1), w, bias, dilation=d, padding=(k - 1) * d // 2)
    if res:
        ref = ref + x
    N = M.native
    xc, vc, gc, bc = x.cuda(), v.cuda(), g.cuda(), bias.cuda()
    y = torch.full((B, co, L), float("nan"), device="cuda")
    N.stats_reset()
    N.check(N.lib().svoc_conv1d_winograd(N.stream_ptr(), N.ptr(xc), N.ptr(vc), N.ptr(gc), N.ptr(bc), N.ptr(xc) if res else None,
                                         N.ptr(y), B, C, co, L, k, d, ctypes.c_float(0.1)))
    st = N.stats_get()
    G = (k + 1) // 4
    want = ((1.5 if _wino_f4_expected(C, co, k, d, L) else 2.0) * G + (G - 1)) / k
    if _wino_f44_expected(C, co, k, d, L):
        want = 1.75 * G / k
    assert st["conv_flops"] == 2.0 * C * co * k * B * L, st
    assert abs(st["executed_flops"] / st["conv_flops"] - want) < 1e-9, (st, want)
    check(f"winograd C{C} k{k} d{d} L{L}", y, ref)
    rc = N.lib().svoc_conv1d_winograd(N.stream_ptr(), N.ptr(xc), N.ptr(vc), N.ptr(gc), N.ptr(bc), None, N.ptr(y), B, C, co, L, 5, 1, ctypes.c_float(0.1))
    assert rc == -5


def test_error_paths_return_codes(M):
    """Reference behaviour at the boundary is 'raise'; at the C ABI that is a negative svoc_status plus a message, never a
    crash: NULL arguments, a convolution whose dilation halo cannot fit the 160 KB LDS, conditioning passed to a module
    built without gin_channels, a missing tensor at create time."""
    import ctypes
    N = M.native
    lib = N.lib()
    x = torch.zeros(1, 32, 64, device="cuda"); w = torch.zeros(32, 32, 11, device="cuda"); y = torch.zeros(1, 32, 64, device="cuda")
    assert lib.svoc_conv1d(N.stream_ptr(), None, N.ptr(w), None, None, None, N.ptr(y), 1, 32, 32, 64, 11, 1, ctypes.c_float(1.0)) == -1
    rc = lib.svoc_conv1d(N.stream_ptr(), N.ptr(x), N.ptr(w), None, None, None, N.ptr(y), 1, 32, 32, 64, 11, 4000, ctypes.c_float(1.0))
    assert rc == -5 and b"LDS" in lib.svoc_last_error()
    torch.cuda.synchronize()
    sd = sw.fill_state_dict(cases.wn_shapes(64, 3, 2, 0), 1)
    m = load(M.modules.WN(64, 3, 1, 2, gin_channels=0), sd)
    xm = torch.zeros(1, 64, 40, device="cuda"); mask = torch.ones(1, 1, 40, device="cuda")
    with pytest.raises(N.SvocError):
        m(xm, mask, g=torch.zeros(1, 8, 1, device="cuda"))
    h = ctypes.c_void_p()
    empty = N.TensorTable({})
    rc = lib.svoc_wn_create(ctypes.byref(h), 64, 3, 1, 2, 0, empty.arr, empty.n, b"")
    assert rc == -2 and not h.value and b"in_layers" in lib.svoc_last_error()


def test_conv1d_random_shapes(M):
    """Seeded sweep over odd shapes: channel counts that are not multiples of 32, every tile configuration (many / few
    columns), large dilations (wide LDS rows), lengths that leave ragged last tiles and workgroup totals that are not
    multiples of 8 (XCD remap), against torch's CPU conv1d."""
    import ctypes
    rng = np.random.default_rng(20260928)
    N = M.native
    for it in range(28):
        C = int(rng.choice([1, 7, 32, 33, 64, 96, 100, 128, 160, 256]))
        co = int(rng.choice([1, 5, 32, 48, 64, 128, 130, 256]))
        k = int(rng.choice([1, 3, 5, 7, 11, 13]))
        d = int(rng.choice([1, 2, 3, 5, 9]))
        L = int(rng.choice([1, 2, 31, 64, 129, 500, 1023, 4100]))
        B = int(rng.choice([1, 2, 3, 5]))
        slope = float(rng.choice([1.0, 0.1, 0.01]))
        res = bool(rng.integers(0, 2)) and co == C
        seed = 9000 + it
        v = T(cases.rnd(seed, "v", (co, C, k), 1.0 / np.sqrt(C * k)))
        g = T((0.5 + sw.uniform01(seed, "g", co)).astype(np.float32)).reshape(co, 1, 1)
        bias = T(cases.rnd(seed, "b", (co,), 0.1))
        x = T(cases.rnd(seed, "x", (B, C, L), 1.0))
        w = O.fold_weight_norm(v, g)
        ref = torch.nn.functional.conv1d(torch.nn.functional.leaky_relu(x, slope), w, bias, dilation=d, padding=(k * d - d) // 2)
        if res:
            ref = ref + x
        xc, vc, gc, bc = x.cuda(), v.cuda(), g.cuda(), bias.cuda()
        y = torch.full((B, co, L), float("nan"), device="cuda")
        N.check(N.lib().svoc_conv1d(N.stream_ptr(), N.ptr(xc), N.ptr(vc), N.ptr(gc), N.ptr(bc), N.ptr(xc) if res else None,
                                    N.ptr(y), B, C, co, L, k, d, ctypes.c_float(slope)))
        check(f"conv1d#{it} C{C}->{co} k{k} d{d} L{L} B{B} slope{slope} res{res}", y, ref)


@pytest.mark.parametrize("k,s", [(16, 8), (4, 2), (8, 8), (2, 2), (6, 2), (12, 4), (3, 1), (1, 1), (7, 3), (16, 4), (10, 4)])
def test_conv_transpose_geometries(M, k, s):
    """Polyphase ConvTranspose1d beyond the model's k = 2*stride upsamplers (k < 2s, k > 2s, k % s != 0, stride 1), odd
    channel counts, ragged lengths; k - stride odd (output length != L*stride) is refused with an error."""
    import ctypes
    N = M.native
    for (ci, co, L, B) in [(64, 32, 50, 2), (33, 7, 129, 1)]:
        seed = 7000 + 31 * k + s + ci
        v = T(cases.rnd(seed, "v", (ci, co, k), 1.0 / np.sqrt(ci * k)))
        g = T((0.5 + sw.uniform01(seed, "g", ci)).astype(np.float32)).reshape(ci, 1, 1)
        bias = T(cases.rnd(seed, "b", (co,), 0.1))
        x = T(cases.rnd(seed, "x", (B, ci, L), 1.0))
        w = O.fold_weight_norm(v, g)
        ref = torch.nn.functional.conv_transpose1d(torch.nn.functional.leaky_relu(x, 0.1), w, bias, stride=s, padding=(k - s) // 2)
        assert ref.shape[2] == L * s
        xc, vc, gc, bc = x.cuda(), v.cuda(), g.cuda(), bias.cuda()
        y = torch.full((B, co, L * s), float("nan"), device="cuda")
        N.check(N.lib().svoc_conv_transpose1d(N.stream_ptr(), N.ptr(xc), N.ptr(vc), N.ptr(gc), N.ptr(bc), N.ptr(y), B, ci, co, L, k, s,
                                              ctypes.c_float(0.1)))
        check(f"convT k{k} s{s} ci{ci}", y, ref)
    xc = torch.zeros(1, 8, 4, device="cuda"); vc = torch.zeros(8, 8, k + 1, device="cuda"); y = torch.zeros(1, 8, 4 * s, device="cuda")
    rc = N.lib().svoc_conv_transpose1d(N.stream_ptr(), N.ptr(xc), N.ptr(vc), None, None, N.ptr(y), 1, 8, 8, 4, k + 1, s, ctypes.c_float(1.0))
    assert rc != 0       # (k+1) - s has the other parity


@pytest.mark.parametrize("ci,co,k,s,L,B", [(128, 64, 16, 8, 1000, 16), (128, 64, 16, 8, 1020, 17), (64, 32, 16, 8, 4096, 9),
                                          (512, 256, 16, 8, 256, 16), (128, 64, 4, 2, 4000, 16), (64, 64, 4, 2, 8188, 9),
                                          (64, 32, 4, 2, 16384, 9), (64, 8, 16, 8, 8192, 16), (128, 24, 16, 8, 2048, 22), (256, 128, 16, 8, 1024, 16),
                                          (128, 128, 4, 2, 4096, 17)])
def test_conv_transpose_winograd(M, ci, co, k, s, L, B):
    """The upsamplers' F(4,2) kernel (convt_wino.hip): k = 2 s, stride 8 and 2, shapes large enough to pass its
    workgroup-count gate (checked through the executed-flop counter: 5/8 of the algorithmic count); ragged last window tiles, the extra
    column q = L, both edges; 256-row blocks (eight consumer
    waves), 128-row blocks and 64-row blocks (two row tiles x two column halves)."""
    import ctypes, os
    if os.environ.get("SVOC_CT_WINO") == "0":
        pytest.skip("SVOC_CT_WINO=0: the F(4,2) kernel is switched off")
    N = M.native
    seed = 7700 + ci + 3 * L + s
    v = T(cases.rnd(seed, "v", (ci, co, k), 1.0 / np.sqrt(ci * k / s)))
    g = T((0.5 + sw.uniform01(seed, "g", ci)).astype(np.float32)).reshape(ci, 1, 1)
    bias = T(cases.rnd(seed, "b", (co,), 0.1))
    x = T(cases.rnd(seed, "x", (B, ci, L), 1.0))
    w = O.fold_weight_norm(v, g)
    ref = torch.nn.functional.conv_transpose1d(torch.nn.functional.leaky_relu(x, 0.1), w, bias, stride=s, padding=(k - s) // 2)
    xc, vc, gc, bc = x.cuda(), v.cuda(), g.cuda(), bias.cuda()
    y = torch.full((B, co, L * s), float("nan"), device="cuda")
    N.stats_reset()
    N.check(N.lib().svoc_conv_transpose1d(N.stream_ptr(), N.ptr(xc), N.ptr(vc), N.ptr(gc), N.ptr(bc), N.ptr(y), B, ci, co, L, k, s,
                                          ctypes.c_float(0.1)))
    st = N.stats_get()
    assert st["conv_flops"] > 0 and abs(st["executed_flops"] / st["conv_flops"] - 5.0 / 8.0) < 1e-6, st
    check(f"convT F(4,2) k{k} s{s} ci{ci} L{L}", y, ref)
    if L == 1000:   # no activation in front (slope 1) and another slope
        for slope in (1.0, 0.25):
            ref2 = torch.nn.functional.conv_transpose1d(torch.nn.functional.leaky_relu(x, slope), w, bias, stride=s, padding=(k - s) // 2)
            y.fill_(float("nan"))
            N.check(N.lib().svoc_conv_transpose1d(N.stream_ptr(), N.ptr(xc), N.ptr(vc), N.ptr(gc), N.ptr(bc), N.ptr(y), B, ci, co, L, k, s,
                                                  ctypes.c_float(slope)))
            check(f"convT F(4,2) slope {slope}", y, ref2)


@pytest.mark.parametrize("name", list(cases.UPS_CASES))
def test_conv_transpose(M, name):
    import ctypes
    c = cases.UPS_CASES[name]
    sd = sdT(sw.fill_state_dict(cases.ups_shapes(c["Ci"], c["Co"], c["k"]), c["seed"], 1.0))
    x = T(cases.rnd(c["seed"], "x", (c["B"], c["Ci"], c["L"]), 0.5))
    N = M.native
    xc, vc, gc, bc = x.cuda(), sd["weight_v"].cuda(), sd["weight_g"].cuda(), sd["bias"].cuda()   # keep alive across the call
    y = torch.empty(c["B"], c["Co"], c["L"] * c["s"], device="cuda")
    N.check(N.lib().svoc_conv_transpose1d(N.stream_ptr(), N.ptr(xc), N.ptr(vc), N.ptr(gc), N.ptr(bc), N.ptr(y), c["B"], c["Ci"],
                                          c["Co"], c["L"], c["k"], c["s"], ctypes.c_float(0.1)))
    check(name, y, cases.golden(name)["y"])


# ----------------------------------------------------------------------------- modules
@pytest.mark.parametrize("name", list(cases.RESBLOCK1_CASES))
def test_resblock1(M, name):
    c = cases.RESBLOCK1_CASES[name]
    sd = sw.fill_state_dict(cases.resblock1_shapes(c["C"], c["k"]), c["seed"], 1.0)
    m = load(M.modules.ResBlock1(c["C"], c["k"], c["d"]), sd)
    x = T(cases.rnd(c["seed"], "x", (c["B"], c["C"], c["L"]), 0.5))
    mask = T(cases.lengths_mask(c["mask_lengths"], c["L"])) if "mask_lengths" in c else None
    y = m(x.cuda(), mask.cuda() if mask is not None else None)
    check(name + " vs oracle", y, O.resblock1(sdT(sd), "", x, c["k"], c["d"], mask))
    check(name + " vs golden", y, cases.golden(name)["y"])


def test_resblock1_edge_lengths(M):
    """ResBlock1 (fused kernel for C = 32 / 64, two convolutions for C = 128) at lengths around the tile edges: shorter than
    the halo, exactly N2 = N_A - (k-1), one more, a ragged last tile, and many tiles; every kernel size of the model and a
    non-standard one; with the padding mask as well (unfused path)."""
    rng = np.random.default_rng(11)
    for it, (C, k) in enumerate([(32, 3), (32, 11), (64, 3), (64, 7), (64, 11), (32, 5), (128, 7)]):
        NA = 256 if C == 32 else 128
        n2 = NA - (k - 1)
        sd = sw.fill_state_dict(cases.resblock1_shapes(C, k), 8100 + it, 1.0)
        m = load(M.modules.ResBlock1(C, k, (1, 3, 5)), sd)
        for L in (1, 5, n2 - 1, n2, n2 + 1, 2 * n2 + 3, 1000):
            B = int(rng.integers(1, 4))
            x = T(cases.rnd(8200 + it, f"x{L}", (B, C, L), 0.5))
            with torch.no_grad():
                ref = O.resblock1(sdT(sd), "", x, k, (1, 3, 5))
            check(f"resblock1 C{C} k{k} L{L} B{B}", m(x.cuda()), ref.numpy())
        L = 300
        x = T(cases.rnd(8300 + it, "xm", (2, C, L), 0.5))
        mask = T(cases.lengths_mask([L, 111], L))
        with torch.no_grad():
            ref = O.resblock1(sdT(sd), "", x, k, (1, 3, 5), mask=mask)
        check(f"resblock1 masked C{C} k{k}", m(x.cuda(), mask.cuda()), ref.numpy())


@pytest.mark.parametrize("name", list(cases.RESBLOCK2_CASES))
def test_resblock2(M, name):
    c = cases.RESBLOCK2_CASES[name]
    sd = sw.fill_state_dict(cases.resblock2_shapes(c["C"], c["k"]), c["seed"], 1.0)
    m = load(M.modules.ResBlock2(c["C"], c["k"], c["d"]), sd)
    x = T(cases.rnd(c["seed"], "x", (c["B"], c["C"], c["L"]), 0.5))
    check(name, m(x.cuda()), cases.golden(name)["y"])


@pytest.mark.parametrize("name", list(cases.WN_CASES))
def test_wn(M, name):
    c = cases.WN_CASES[name]
    sd = sw.fill_state_dict(cases.wn_shapes(c["H"], c["k"], c["n"], c["gin"]), c["seed"])
    m = load(M.modules.WN(c["H"], c["k"], c["dr"], c["n"], gin_channels=c["gin"]), sd)
    x = T(cases.rnd(c["seed"], "x", (c["B"], c["H"], c["T"]), 1.0))
    mask = T(cases.lengths_mask(c["lengths"], c["T"]))
    g = T(cases.rnd(c["seed"], "g", (c["B"], c["gin"], 1), 1.0)) if c["gin"] else None
    y = m((x * mask).cuda(), mask.cuda(), g=g.cuda() if g is not None else None)
    check(name, y, cases.golden(name)["y"])


def test_wn_edge_shapes(M):
    """WN over hidden sizes with even / odd numbers of 32-channel chunks (K-split 12-wave kernel vs one wave per row pair),
    lengths around the 32-column tile, short and long batches (NR = 1 / 2 tiles), with and without conditioning."""
    rng = np.random.default_rng(12)
    for it, (H, k, dr, n, gin) in enumerate([(192, 5, 1, 2, 0), (64, 3, 2, 3, 16), (96, 5, 1, 2, 0), (160, 3, 1, 2, 8), (256, 5, 1, 2, 0)]):
        sd = sw.fill_state_dict(cases.wn_shapes(H, k, n, gin), 8400 + it)
        m = load(M.modules.WN(H, k, dr, n, gin_channels=gin), sd)
        for Tn, B in ((1, 1), (31, 2), (32, 1), (33, 3), (100, 2), (513, 2), (700, 24)):
            x = T(cases.rnd(8500 + it, f"x{Tn}", (B, H, Tn), 1.0))
            lens = [Tn] + [int(rng.integers(1, Tn + 1)) for _ in range(B - 1)]
            mask = T(cases.lengths_mask(lens, Tn))
            g = T(cases.rnd(8500 + it, f"g{Tn}", (B, gin, 1), 1.0)) if gin else None
            with torch.no_grad():
                ref = O.wn(sdT(sd), "", x * mask, mask, g, hidden=H, kernel_size=k, dilation_rate=dr, n_layers=n)
            y = m((x * mask).cuda(), mask.cuda(), g=g.cuda() if g is not None else None)
            check(f"wn H{H} k{k} dr{dr} T{Tn} B{B}", y, ref.numpy())


@pytest.mark.parametrize("n_layers,B,Tn", [(16, 8, 520), (8, 5, 1000), (3, 16, 512), (2, 255, 32), (8, 6, 512), (4, 32, 512), (3, 4, 4096), (2, 32, 500)])
def test_wn_stack_one_persistent_launch(M, n_layers, B, Tn, tmp_path):
    """A whole WN stack in ONE persistent launch (csrc/wn_stack.hip; reference modules.py:148-176): taken while every 32-column tile has a CU of its
    own (96 .. 256 tiles: from 3/8 of the CUs on), the tiles exchanging two-column edges between the layers.  Ragged lengths (an utterance that ends inside a tile, one
    of a single frame), a last tile of 8 columns (T = 520), 255 one-tile utterances (no neighbours at all).  Against the oracle, and BIT FOR BIT against
    the one-launch-per-layer path (a second process with SVOC_WN_STACK=0: the arithmetic per layer is the same instruction stream)."""
    import subprocess, sys, os
    if any(os.environ.get(k) == "0" for k in ("SVOC_WN_STACK", "SVOC_WN_F25", "SVOC_FUSE_WN")):
        pytest.skip("a variant run that switches the persistent stack launch (or the layer kernel under it) off")
    rng = np.random.default_rng(n_layers * 1000 + B)
    sd = sw.fill_state_dict(cases.wn_shapes(192, 5, n_layers, 0), 8700 + n_layers)
    m = load(M.modules.WN(192, 5, 1, n_layers, gin_channels=0), sd)
    x = T(cases.rnd(8800 + B, "x", (B, 192, Tn), 1.0))
    lens = [Tn, 1] + [int(rng.integers(1, Tn + 1)) for _ in range(B - 2)]
    mask = T(cases.lengths_mask(lens, Tn))
    M.native.stats_reset()
    y = m((x * mask).cuda(), mask.cuda())
    st = M.native.stats_get()
    # ONE GEMM-family launch for the 2 * n_layers convolutions - or, for a batch beyond the launch's capacity (32 x 512, 4 x 4096: 512 tiles), one per group of
    # whole utterances (round 6: csrc/wn_stack.hip wn_stack_groups)
    groups = max(1, -(-B * ((Tn + 31) // 32) // 256))
    assert st["conv_launches"] == groups, st
    assert st["convolutions"] == 2 * n_layers
    with torch.no_grad():
        ref = O.wn(sdT(sd), "", x * mask, mask, None, hidden=192, kernel_size=5, dilation_rate=1, n_layers=n_layers)
    check(f"wn stack n{n_layers} B{B} T{Tn}", y, ref.numpy())
    y2 = m((x * mask).cuda(), mask.cuda())
    assert torch.equal(y, y2)                                  # the neighbour hand-shake leaves no run-to-run difference
    code = f"""
import sys, numpy as np, torch
sys.path.insert(0, {cases.ROOT!r}); sys.path.insert(0, {os.path.join(cases.ROOT, 'tests')!r})
import cases
from cases import sw
from smart_vocoder_amd import modules, _native
m = modules.WN(192, 5, 1, {n_layers}, gin_channels=0)
m.load_state_dict({{k: torch.from_numpy(v) for k, v in sw.fill_state_dict(cases.wn_shapes(192, 5, {n_layers}, 0), {8700 + n_layers}).items()}})
m = m.cuda().eval()
x = torch.from_numpy(cases.rnd({8800 + B}, "x", ({B}, 192, {Tn}), 1.0)); mask = torch.from_numpy(cases.lengths_mask({lens!r}, {Tn}))
_native.stats_reset()
y = m((x * mask).cuda(), mask.cuda())
assert _native.stats_get()["conv_launches"] == {n_layers}, _native.stats_get()
np.save({str(tmp_path / 'y.npy')!r}, y.cpu().numpy())
"""
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SVOC_WN_STACK="0"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    assert np.array_equal(np.load(str(tmp_path / "y.npy")), y.cpu().numpy()), "persistent stack differs from the per-layer launches"


@pytest.mark.parametrize("n_layers,B,Tn", [(16, 1, 200), (8, 1, 200), (8, 2, 100), (2, 10, 7), (3, 1, 320), (4, 2, 33),
                                           (16, 1, 512), (8, 1, 640), (3, 2, 288), (4, 2, 200), (2, 5, 33)])
def test_wn_mesh_short_inputs_one_persistent_launch(M, n_layers, B, Tn):
    """A whole WN stack for SHORT inputs in ONE persistent launch (csrc/wn_mesh.hip; reference modules.py:148-176): twelve workgroups per
    32-column tile that hand acts rows and x rows (with the k = 5 halo of the neighbouring tiles) to each other twice per layer.  BASELINE
    configs[0] (1 x 200: seven tiles, the last of 8 columns), ragged batches, ten one-tile utterances (no neighbours), the largest grid the
    launcher takes with one tile per workgroup group (1 x 320: ten tiles = 120 of 256 CUs), a last tile of ONE column; then TWO tiles per group (11 .. 20
    tiles): 1 x 512, the largest input (1 x 640), odd tile counts per utterance (2 x 288: nine tiles, 2 x 200: seven - the last group holds one tile), five
    two-tile utterances.  Against the oracle; run to run bit-identical."""
    import os
    if any(os.environ.get(k) == "0" for k in ("SVOC_WN_MESH", "SVOC_WN_F25", "SVOC_WN_SMALL", "SVOC_WN_SMALL_F25", "SVOC_FUSE_WN")):
        pytest.skip("a variant run that switches the short-input kernels off")
    rng = np.random.default_rng(n_layers * 1000 + B)
    sd = sw.fill_state_dict(cases.wn_shapes(192, 5, n_layers, 0), 8900 + n_layers)
    m = load(M.modules.WN(192, 5, 1, n_layers, gin_channels=0), sd)
    x = T(cases.rnd(9000 + B, "x", (B, 192, Tn), 1.0))
    lens = [Tn] + [int(rng.integers(1, Tn + 1)) for _ in range(B - 1)]
    mask = T(cases.lengths_mask(lens, Tn))
    M.native.stats_reset()
    y = m((x * mask).cuda(), mask.cuda())
    st = M.native.stats_get()
    assert st["conv_launches"] == 1, st                       # ONE launch for the 2 * n_layers convolutions
    assert st["convolutions"] == 2 * n_layers
    with torch.no_grad():
        ref = O.wn(sdT(sd), "", x * mask, mask, None, hidden=192, kernel_size=5, dilation_rate=1, n_layers=n_layers)
    check(f"wn mesh n{n_layers} B{B} T{Tn}", y, ref.numpy())
    for _ in range(20):                                         # the hand-overs leave no run-to-run difference, the flags are clean for the next launch
        assert torch.equal(y, m((x * mask).cuda(), mask.cuda()))
    xo = T(cases.rnd(9050 + B, "xo", (2, 192, 150), 1.0))    # another (B, T) lays the hand-over rows out differently: nothing of it may be read afterwards
    m(xo.cuda(), torch.ones(2, 1, 150).cuda())
    assert torch.equal(y, m((x * mask).cuda(), mask.cuda()))
    x2 = T(cases.rnd(9100 + B, "x2", (B, 192, Tn), 1.0))     # ... and nothing of the previous input survives in the hand-over rows
    with torch.no_grad():
        ref2 = O.wn(sdT(sd), "", x2 * mask, mask, None, hidden=192, kernel_size=5, dilation_rate=1, n_layers=n_layers)
    check(f"wn mesh n{n_layers} B{B} T{Tn} second input", m((x2 * mask).cuda(), mask.cuda()), ref2.numpy())


@pytest.mark.parametrize("n_layers,B,Tn", [(8, 2, 512), (8, 3, 512), (16, 4, 512), (8, 5, 512), (3, 4, 500)])
def test_wn_mid_size_batches(M, n_layers, B, Tn):
    """21 .. 95 tiles of 32 columns (2 x 512 .. 5 x 512) sit between the short-input persistent launch and the stack launch and run
    the chain of csrc/wn_small.hip - one launch per layer, since round 6 with TWO row pairs per workgroup where that makes one round of workgroups out of
    two (43 .. 85 tiles).  Against the oracle, ragged lengths; n_layers + 1 launches (the last layer's 1 x 1 is a convolution of its own)."""
    sd = sw.fill_state_dict(cases.wn_shapes(192, 5, n_layers, 0), 9500 + n_layers)
    m = load(M.modules.WN(192, 5, 1, n_layers, gin_channels=0), sd)
    rng = np.random.default_rng(B * 1000 + Tn)
    x = T(cases.rnd(9500 + B, "x", (B, 192, Tn), 1.0))
    lens = [Tn] + [int(rng.integers(Tn // 2, Tn + 1)) for _ in range(B - 1)]
    mask = T(cases.lengths_mask(lens, Tn))
    M.native.stats_reset()
    y = m((x * mask).cuda(), mask.cuda())
    st = M.native.stats_get()
    assert st["convolutions"] == 2 * n_layers, st
    if not any(k in os.environ for k in ("SVOC_WN_F25", "SVOC_WN_SMALL_F25", "SVOC_WN_SMALL", "SVOC_KSPLIT", "SVOC_FUSE_WN", "SVOC_WN_SMALL_TILES")):
        assert st["conv_launches"] == n_layers + 1, st        # (the variant runs of tests/test_gpu_variants.py take other launch forms at these shapes)
    with torch.no_grad():
        ref = O.wn(sdT(sd), "", x * mask, mask, None, hidden=192, kernel_size=5, dilation_rate=1, n_layers=n_layers)
    check(f"wn mid-size n{n_layers} B{B} T{Tn}", y, ref.numpy())
    assert torch.equal(y, m((x * mask).cuda(), mask.cuda()))


def test_persistent_launch_failure_is_reported_by_the_next_call(M):
    """The persistent WN launches bound their waits (SVOC_PERSIST_TIMEOUT_MS) and a workgroup that gives up raises a word in pinned host memory
    (include/svoc.h svoc_check_async_error).  Raised from the host here, exactly as the device would: the NEXT WN call fails with the reason, once;
    from then on the process takes one launch per layer (same function) until the diagnostics switch re-enables the persistent launches;
    `check_async_error()` itself reports and clears as well."""
    sd = sw.fill_state_dict(cases.wn_shapes(192, 5, 3, 0), 9300)
    m = load(M.modules.WN(192, 5, 1, 3, gin_channels=0), sd)
    x = T(cases.rnd(9301, "x", (1, 192, 100), 1.0)).cuda(); mask = torch.ones(1, 1, 100).cuda()
    y = m(x, mask)
    M.native.check_async_error()
    assert M.native.persist_state()[0] is False
    try:
        M.native.check(M.native.lib().svoc_debug_raise_async_error())
        with pytest.raises(RuntimeError, match="persistent WN launch"):
            m(x, mask)
        assert M.native.persist_state()[0] is True                 # safe mode: per-layer launches
        M.native.stats_reset()
        y_layers = m(x, mask)
        assert M.native.stats_get()["conv_launches"] > 1
        check("wn per-layer launches after a reported give-up", y_layers, y.cpu().numpy())
        M.native.debug_persist_control(reenable=True)
        assert M.native.persist_state()[0] is False
        assert torch.equal(m(x, mask), y)
        M.native.check(M.native.lib().svoc_debug_raise_async_error())
        with pytest.raises(RuntimeError, match="persistent WN launch"):
            M.native.check_async_error()
        M.native.check_async_error()
    finally:
        M.native.debug_persist_control(reenable=True)


@pytest.mark.parametrize("kind,n_layers,B,Tn", [("stack", 8, 8, 520), ("stack", 16, 16, 512), ("mesh", 16, 1, 200), ("mesh", 8, 1, 512)])
def test_persistent_launch_device_side_give_up_poisons_the_same_call(M, kind, n_layers, B, Tn):
    """a workgroup of a persistent WN launch that gives up a wait ON THE DEVICE must make the call it corrupts visibly wrong.
    The diagnostics switch makes the workgroup of tile 3 withhold the flags its neighbours wait for and bounds the waits by 5 ms: the neighbours give
    up, their rows of THAT call's output are NaN (the reference's WN.forward, modules.py:148-176, cannot return a wrong finite tensor), the next call
    fails with the reason, the process falls back to one launch per layer, and with the switch off again the launch is clean and bit-identical."""
    sd = sw.fill_state_dict(cases.wn_shapes(192, 5, n_layers, 0), 9400 + n_layers)
    m = load(M.modules.WN(192, 5, 1, n_layers, gin_channels=0), sd)
    x = T(cases.rnd(9401 + B, "x", (B, 192, Tn), 1.0)).cuda(); mask = torch.ones(B, 1, Tn).cuda()
    M.native.check_async_error()
    M.native.stats_reset()
    y = m(x, mask)
    torch.cuda.synchronize()
    assert M.native.stats_get()["conv_launches"] == 1                 # the persistent launch is what runs
    assert torch.isfinite(y).all()
    try:
        M.native.debug_persist_control(fault_tile=3, timeout_ms=5)
        assert M.native.persist_state() == (False, 5)
        t0 = time.time()
        yb = m(x, mask)                                              # returns success: nothing is known yet on the host ...
        torch.cuda.synchronize()
        assert time.time() - t0 < 5.0                                # (one bounded wait per workgroup, not one per layer)
        assert not torch.isfinite(yb).all()                          # ... but the result says so
        assert torch.isnan(yb[0, :, 64:96]).all() and torch.isnan(yb[0, :, 128:160]).all()      # tiles 2 and 4 waited for tile 3
        with pytest.raises(RuntimeError, match="persistent WN launch"):
            M.native.check_async_error()                             # right behind the caller's own synchronisation of the affected call
        assert M.native.persist_state()[0] is True
        M.native.stats_reset()
        y_layers = m(x, mask)                                        # fallen back: one launch per layer, the fault switch does not reach it
        assert M.native.stats_get()["conv_launches"] > 1
        check(f"wn {kind} per-layer launches after a give-up", y_layers, y.cpu().numpy())
    finally:
        M.native.debug_persist_control(reenable=True)
    M.native.stats_reset()
    y2 = m(x, mask)
    assert M.native.stats_get()["conv_launches"] == 1
    assert torch.equal(y2, y)                                        # flags and counters were left clean by the launch that gave up
    torch.cuda.synchronize()
    M.native.check_async_error()


@pytest.mark.parametrize("name", list(cases.POSTERIOR_CASES))
def test_posterior_encoder(M, name):
    c = cases.POSTERIOR_CASES[name]
    sd = sw.fill_state_dict(cases.posterior_shapes(c["Cin"], c["Cout"], c["H"], c["k"], c["n"], c["gin"]), c["seed"])
    m = load(M.models.PosteriorEncoder(c["Cin"], c["Cout"], c["H"], c["k"], c["dr"], c["n"], gin_channels=c["gin"]), sd)
    gold = cases.golden(name)
    x = T(cases.rnd(c["seed"], "x", (c["B"], c["Cin"], c["T"]), 1.0))
    g = T(cases.rnd(c["seed"], "g", (c["B"], c["gin"], 1), 1.0)) if c["gin"] else None
    z, mq, logs, mask = m(x.cuda(), torch.tensor(c["lengths"]).cuda(), g=g.cuda() if g is not None else None, eps=T(gold["eps"]).cuda())
    check(name + ".m", mq, gold["m"])
    check(name + ".logs", logs, gold["logs"])
    check(name + ".z", z, gold["z"], atol=5e-5)
    assert np.array_equal(mask.cpu().numpy(), gold["mask"])


def test_voice_conversion_needs_emb_g_like_the_reference(M):
    # reference models.py:341-349 dereferences self.emb_g, which models.py:305-314 never creates
    net = M.models.SynthesizerTrn(513, 32, n_speakers=109, **cases.IITP_MODEL)
    with pytest.raises(AttributeError):
        net.voice_conversion(torch.zeros(1, 513, 8), torch.tensor([8]), torch.tensor([0]), torch.tensor([1]))


@pytest.mark.parametrize("name", list(cases.COUPLING_CASES))
def test_coupling(M, name):
    c = cases.COUPLING_CASES[name]
    sd = sw.fill_state_dict(cases.coupling_shapes(c["C"], c["H"], c["k"], c["n"], c["gin"], c["mean_only"]), c["seed"])
    m = load(M.modules.ResidualCouplingLayer(c["C"], c["H"], c["k"], c["dr"], c["n"], gin_channels=c["gin"], mean_only=c["mean_only"]), sd)
    x = T(cases.rnd(c["seed"], "x", (c["B"], c["C"], c["T"]), 1.0))
    mask = T(cases.lengths_mask(c["lengths"], c["T"]))
    g = T(cases.rnd(c["seed"], "g", (c["B"], c["gin"], 1), 1.0)) if c["gin"] else None
    r = m(x.cuda(), mask.cuda(), g=g.cuda() if g is not None else None, reverse=c["reverse"])
    gold = cases.golden(name)
    if c["reverse"]:
        check(name, r, gold["y"])
    else:
        check(name, r[0], gold["y"])
        check(name + " logdet", r[1], gold["logdet"], 1e-3, 1e-4)


@pytest.mark.parametrize("name", list(cases.FLOWBLOCK_CASES))
def test_flowblock(M, name):
    c = cases.FLOWBLOCK_CASES[name]
    sd = sw.fill_state_dict(cases.flowblock_shapes(c["n"]), c["seed"])
    m = load(M.models.ResidualCouplingBlock(192, 192, 5, 1, c["n"], gin_channels=0), sd)
    x = T(cases.rnd(c["seed"], "x", (c["B"], 192, c["T"]), 1.0))
    mask = T(cases.lengths_mask(c["lengths"], c["T"]))
    check(name, m(x.cuda(), mask.cuda(), reverse=c["reverse"]), cases.golden(name)["y"])


def test_flowblock_roundtrip(M):
    """size-independent property: reverse(forward(x)) == x on the valid frames (both directions share folded flips)."""
    sd = sw.fill_state_dict(cases.flowblock_shapes(2), 6199)
    m = load(M.models.ResidualCouplingBlock(192, 192, 5, 1, 2, gin_channels=0), sd)
    x = T(cases.rnd(6199, "x", (2, 192, 300), 1.0))
    mask = T(cases.lengths_mask([300, 211], 300))
    xm = (x * mask).cuda()
    y = m(xm, mask.cuda(), reverse=False)
    xr = m(y, mask.cuda(), reverse=True)
    check("flow roundtrip", xr, xm.cpu(), 5e-5, 1e-4)


@pytest.mark.parametrize("name", list(cases.GENERATOR_CASES))
def test_generator(M, name):
    c = cases.GENERATOR_CASES[name]
    sd = sw.fill_state_dict(cases.generator_shapes(c), c["seed"], 1.0)
    m = load(M.models.Generator(c["initial_channel"], c["resblock"], c["rks"], c["rds"], c["ur"], c["uic"], c["uks"], gin_channels=c["gin"]), sd)
    x = T(cases.rnd(c["seed"], "x", (c["B"], c["initial_channel"], c["T"]), 1.0))
    g = T(cases.rnd(c["seed"], "g", (c["B"], c["gin"], 1), 1.0)) if c["gin"] else None
    check(name, m(x.cuda(), g=g.cuda() if g is not None else None), cases.golden(name)["y"], 5e-5, 1e-4)


def test_remove_weight_norm_is_identity(M):
    c = cases.GENERATOR_CASES["gen_small_g"]
    sd = sw.fill_state_dict(cases.generator_shapes(c), c["seed"], 1.0)
    m = load(M.models.Generator(c["initial_channel"], c["resblock"], c["rks"], c["rds"], c["ur"], c["uic"], c["uks"], gin_channels=c["gin"]), sd)
    x = T(cases.rnd(c["seed"], "x", (c["B"], c["initial_channel"], c["T"]), 1.0)).cuda()
    y0 = m(x)
    m.remove_weight_norm()
    assert "ups.0.weight" in m.state_dict() and "ups.0.weight_v" not in m.state_dict()
    check("remove_weight_norm", m(x), y0.cpu(), 2e-6, 1e-6)


@pytest.mark.parametrize("name", list(cases.DDS_CASES))
def test_dds(M, name):
    c = cases.DDS_CASES[name]
    sd = sw.fill_state_dict(cases.dds_shapes(c["C"], c["k"], c["n"]), c["seed"], 1.0)
    m = load(M.modules.DDSConv(c["C"], c["k"], c["n"]), sd)
    x = T(cases.rnd(c["seed"], "x", (c["B"], c["C"], c["T"]), 1.0))
    mask = T(cases.lengths_mask(c["lengths"], c["T"]))
    g = T(cases.rnd(c["seed"], "g", (c["B"], c["C"], c["T"]), 0.5)) if c["with_g"] else None
    check(name, m(x.cuda(), mask.cuda(), g=g.cuda() if g is not None else None), cases.golden(name)["y"], 5e-5, 1e-4)


@pytest.mark.parametrize("name", list(cases.CONVFLOW_CASES))
def test_convflow(M, name):
    c = cases.CONVFLOW_CASES[name]
    sd = sw.fill_state_dict(cases.convflow_shapes(c["Cin"], c["F"], c["k"], c["n"]), c["seed"], 2.0)
    m = load(M.modules.ConvFlow(c["Cin"], c["F"], c["k"], c["n"]), sd)
    x = T(cases.rnd(c["seed"], "x", (c["B"], c["Cin"], c["T"]), 2.5))
    mask = T(cases.lengths_mask(c["lengths"], c["T"]))
    r = m(x.cuda(), mask.cuda(), reverse=c["reverse"])
    gold = cases.golden(name)
    if c["reverse"]:
        check(name, r, gold["y"], 1e-3, 0)
    else:
        check(name, r[0], gold["y"], 1e-3, 0)
        check(name + " logdet", r[1], gold["logdet"], 2e-2, 1e-3)


@pytest.mark.parametrize("name", list(cases.SPLINE_CASES))
def test_spline(M, name):
    c = cases.SPLINE_CASES[name]
    x, uw, uh, ud = cases.spline_inputs(name)
    y, lad = M.transforms.piecewise_rational_quadratic_transform(T(x).cuda(), T(uw).cuda(), T(uh).cuda(), T(ud).cuda(),
                                                                 inverse=c["inverse"], tails="linear", tail_bound=5.0)
    gold = cases.golden(name)
    ey = np.abs(y.cpu().numpy() - gold["y"])
    el = np.abs(lad.cpu().numpy() - gold["logabsdet"])
    # an element sitting on a knot may legitimately pick the neighbouring bin: the map is continuous there,
    # its log-derivative is not
    assert ey.max() <= 1e-3, ey.max()
    assert (el > 1e-2).mean() <= 0.01, (el > 1e-2).mean()
    assert np.median(el) <= 1e-5


def test_spline_roundtrip(M):
    x, uw, uh, ud = cases.spline_inputs("spline_fwd")
    f = M.transforms.piecewise_rational_quadratic_transform
    args = (T(uw).cuda(), T(uh).cuda(), T(ud).cuda())
    y, l1 = f(T(x).cuda(), *args, inverse=False, tails="linear", tail_bound=5.0)
    xr, l2 = f(y, *args, inverse=True, tails="linear", tail_bound=5.0)
    err = (xr.cpu() - T(x)).abs()
    assert float(err.quantile(0.999)) <= 1e-3 and float(err.max()) <= 2e-2, (float(err.quantile(0.999)), float(err.max()))


# ----------------------------------------------------------------------------- full path
@pytest.fixture(scope="module")
def net(M):
    n = M.models.SynthesizerTrn(513, 32, n_speakers=109, **cases.IITP_MODEL)
    sd = cases.full_model_weights(skip_enc_q=True)
    missing = n.load_state_dict({k: T(v) for k, v in sd.items()}, strict=False)
    assert all(k.startswith("enc_q.") for k in missing.missing_keys)
    return n.cuda().eval()


@pytest.mark.parametrize("name", list(cases.INFER_CASES))
def test_infer_vs_reference_golden(M, net, name):
    c = cases.INFER_CASES[name]
    mel, ln, eps = cases.infer_inputs(name)
    o, mask, (z, z_p, m_p, logs_p) = net.infer(T(mel).cuda(), T(ln).cuda(), noise_scale=c["noise_scale"], max_len=c["max_len"],
                                               eps=T(eps).cuda())
    g = cases.golden("infer_" + name)
    assert torch.equal(mask.cpu(), T(g["mask"]))
    check(name + " m_p", m_p, g["m_p"]); check(name + " logs_p", logs_p, g["logs_p"])
    check(name + " z_p", z_p, g["z_p"]); check(name + " z", z, g["z"], 5e-5, 1e-4)
    err = o.cpu().numpy() - g["o"]
    rms, ref = float(np.sqrt((err ** 2).mean())), float(np.sqrt((g["o"] ** 2).mean()))
    print(f"{name}: waveform rms err {rms:.3e} (ref rms {ref:.3f}, rel {rms / ref:.2e}), max {np.abs(err).max():.3e}")
    assert rms <= 1e-3 and rms / ref <= 1e-4, (rms, rms / ref)


def test_voice_conversion_with_attached_emb_g(M):
    """models.py:341-349 once an emb_g exists: enc_q -> flow(g_src) -> flow^-1(g_tgt) -> dec(g_tgt), against the oracle."""
    sd = cases.full_model_weights(skip_enc_q=False)
    n = M.models.SynthesizerTrn(513, 32, n_speakers=109, **cases.IITP_MODEL)
    n.load_state_dict({k: T(v) for k, v in sd.items()})
    n.emb_g = torch.nn.Embedding(109, 256)
    n.emb_g.weight.data.copy_(T(cases.rnd(31, "emb_g", (109, 256), 1.0)))
    n = n.cuda().eval()
    Bn, Tn = 2, 24
    y = T(np.abs(cases.rnd(32, "spec", (Bn, 513, Tn), 1.0)))
    ln = torch.tensor([24, 15])
    eps = T(cases.rnd(33, "eps", (Bn, 192, Tn), 1.0))
    src, tgt = torch.tensor([3, 77]), torch.tensor([50, 4])
    o, mask, (z, z_p, z_hat) = n.voice_conversion(y.cuda(), ln.cuda(), src.cuda(), tgt.cuda(), eps=eps.cuda())
    sdt = sdT(sd)
    with torch.no_grad():
        emb = T(cases.rnd(31, "emb_g", (109, 256), 1.0))
        g_src, g_tgt = emb[src].unsqueeze(-1), emb[tgt].unsqueeze(-1)
        rz, _, _, rmask = O.posterior_encoder(sdt, "enc_q.", y, ln, g_src, eps, hidden=192, kernel_size=5, dilation_rate=1, n_layers=16)
        rzp = O.flow(sdt, rz, rmask, g_src, reverse=False)
        rzh = O.flow(sdt, rzp, rmask, g_tgt, reverse=True)
        ro = O.generator(sdt, rzh * rmask, g_tgt)
    check("vc z", z, rz.numpy(), 5e-5, 1e-4); check("vc z_p", z_p, rzp.numpy(), 1e-4, 1e-4); check("vc z_hat", z_hat, rzh.numpy(), 2e-4, 2e-4)
    err = (o.cpu() - ro).numpy()
    rms, ref = float(np.sqrt((err ** 2).mean())), float(ro.pow(2).mean().sqrt())
    assert rms <= 1e-3 and rms / ref <= 1e-3, (rms, rms / ref)


def test_folded_weight_file_round_trip(M, net, tmp_path):
    """export_folded -> load_folded into a fresh model gives the same waveform as the weight-normed model."""
    from smart_vocoder_amd import utils
    mel, ln, eps = cases.infer_inputs("ragged")
    o_ref = net.infer(T(mel).cuda(), T(ln).cuda(), noise_scale=0.667, eps=T(eps).cuda())[0]
    nkeys = len(net.state_dict())
    path = utils.export_folded(net, str(tmp_path / "G_folded.pth"), iteration=7)
    assert len(net.state_dict()) == nkeys and any(k.endswith("weight_g") for k in net.state_dict())      # source untouched
    fresh = M.models.SynthesizerTrn(513, 32, n_speakers=109, **cases.IITP_MODEL).cuda().eval()
    _, it = utils.load_folded(path, fresh)
    assert it == 7 and not any(k.endswith("weight_g") or k.endswith("weight_v") for k in fresh.state_dict())
    o = fresh.infer(T(mel).cuda(), T(ln).cuda(), noise_scale=0.667, eps=T(eps).cuda())[0]
    assert (o - o_ref).abs().max().item() <= 1e-6


BATCH_TOL = 2e-6      # |waveform| <= 1: kernel variants (tile shape, K split) are chosen from the launch size, so an utterance run
                      # alone and inside a batch may be summed in a different order; equal to fp32 rounding, deterministic per shape


def test_infer_batch_independence(M, net):
    """utterances are independent: a batch equals the same utterances run one by one (basis of the multi-GPU shard), to
    fp32 rounding (see BATCH_TOL), and two runs of the same shape are bit-identical."""
    mel, ln, eps = cases.infer_inputs("ragged")
    o, *_ = net.infer(T(mel).cuda(), T(ln).cuda(), noise_scale=0.667, eps=T(eps).cuda())
    o2, *_ = net.infer(T(mel).cuda(), T(ln).cuda(), noise_scale=0.667, eps=T(eps).cuda())
    assert torch.equal(o, o2)
    for b in range(mel.shape[0]):
        ob, *_ = net.infer(T(mel[b:b + 1]).cuda(), T(ln[b:b + 1]).cuda(), noise_scale=0.667, eps=T(eps[b:b + 1]).cuda())
        assert (ob[0] - o[b]).abs().max().item() <= BATCH_TOL, b


def test_full_size_properties(M, net):
    """BASELINE.json configs[1] (16 x 512 frames, the bench workload), where the oracle is too slow: size-independent
    properties instead.  (a) two runs are bit-identical (no atomics / race in the grouped and multi-stream launches);
    (b) every utterance equals its single-utterance run to fp32 rounding (BATCH_TOL); (c) max_len truncates the
    decoder INPUT (models.py:338), so its output equals the full one up to the decoder's receptive field from the cut;
    (d) flow^-1 followed by flow returns z_p; (e) the waveform is finite and inside tanh's range."""
    Bn, Tn = 16, 512
    mel = T(sw.synthetic_mel(1001, Bn, Tn)).cuda(); eps = T(sw.synthetic_eps(1001, Bn, Tn)).cuda()
    ln = torch.full((Bn,), Tn, dtype=torch.int64).cuda()
    ln[3] = 400; ln[9] = 77                                            # two ragged rows
    o1, mask, (z, z_p, m_p, logs_p) = net.infer(mel, ln, noise_scale=0.667, eps=eps)
    o2 = net.infer(mel, ln, noise_scale=0.667, eps=eps)[0]
    assert torch.equal(o1, o2)                                         # (a)
    for b in (0, 3, 9, 15):                                            # (b)
        ob = net.infer(mel[b:b + 1], ln[b:b + 1], noise_scale=0.667, eps=eps[b:b + 1])[0]
        assert (ob[0] - o1[b]).abs().max().item() <= BATCH_TOL, b
    om = net.infer(mel, ln, noise_scale=0.667, eps=eps, max_len=300)[0]
    keep = (300 - M.models.SynthesizerTrn.RECEPTIVE_FRAMES) * 256      # (c)
    assert om.shape[2] == 300 * 256 and (om[:, :, :keep] - o1[:, :, :keep]).abs().max().item() <= BATCH_TOL
    assert not torch.equal(om[:, :, -256:], o1[:, :, 299 * 256:300 * 256])    # the cut really is upstream of the decoder
    back = net.flow(z, mask, reverse=False)                            # (d)
    err = ((back - z_p * mask).abs().max() / z_p.abs().max()).item()
    assert err <= 1e-4, err
    assert torch.isfinite(o1).all() and o1.abs().max().item() <= 1.0   # (e)
    assert o1.shape == (Bn, 1, Tn * 256)



def test_c2_full_size_vs_oracle(M, net):
    """BASELINE.json configs[1] - the bench workload itself, 16 x 512 frames, seed 1001 (exactly bench.py's tensors) - run as ONE batch on the GPU and
    compared with the CPU oracle on four of its utterances (0, 5, 10, 15; the oracle treats utterances independently, and it needs 10-40 s for all
    sixteen - `bench.py` does that whole-batch comparison on every bench run, and `test_full_size_vs_reference_fixture[c2_16x512]` covers every sample
    of the batch against numbers the reference itself produced): waveform RMS <= 1e-3 (north_star) and relative RMS <= 1e-4, latents to the
    per-module tolerance."""
    Bn, Tn = 16, 512
    mel = T(sw.synthetic_mel(1001, Bn, Tn)); eps = T(sw.synthetic_eps(1001, Bn, Tn))
    ln = torch.full((Bn,), Tn, dtype=torch.int64)
    o, mask, (z, z_p, m_p, logs_p) = net.infer(mel.cuda(), ln.cuda(), noise_scale=0.667, eps=eps.cuda())
    assert torch.isfinite(o).all()
    pick = [0, 5, 10, 15]
    with torch.no_grad():
        o_ref, mask_ref, (z_ref, zp_ref, mp_ref, lp_ref) = O.infer(sdT(cases.full_model_weights()), mel[pick], ln[pick], eps[pick], 0.667)
    check("c2 m_p", m_p[pick], mp_ref); check("c2 logs_p", logs_p[pick], lp_ref); check("c2 z_p", z_p[pick], zp_ref); check("c2 z", z[pick], z_ref, 5e-5, 1e-4)
    err = (o[pick].cpu() - o_ref).numpy()
    rms, ref = float(np.sqrt((err ** 2).mean())), float(o_ref.pow(2).mean().sqrt())
    print(f"c2 16x512: waveform rms err {rms:.3e} (ref rms {ref:.3f}, rel {rms / ref:.2e}), max {np.abs(err).max():.3e}")
    assert rms <= 1e-3 and rms / ref <= 1e-4, (rms, rms / ref)


def test_default_noise_is_the_reference_draw(M, net):
    """Without `eps=` the reference draws `torch.randn_like(m_p)` (models.py:336: one N(0,1) draw of a [B, 192, T] fp32
    tensor on the model's device).  Our `infer` must consume the generator the same way: after the same seed, the default
    path equals the injected-eps path given that very draw, bit for bit, and leaves the generator in the same state."""
    mel, ln, _ = cases.infer_inputs("ragged")
    mel, ln = T(mel).cuda(), T(ln).cuda()
    B, _, Tn = mel.shape
    torch.manual_seed(4242)
    o1, _, (z1, zp1, mp1, _) = net.infer(mel, ln, noise_scale=0.667)
    after_default = torch.cuda.get_rng_state()
    torch.manual_seed(4242)
    eps = torch.randn_like(torch.empty(B, 192, Tn, dtype=torch.float32, device="cuda"))
    after_draw = torch.cuda.get_rng_state()
    o2, _, (z2, zp2, mp2, _) = net.infer(mel, ln, noise_scale=0.667, eps=eps)
    assert torch.equal(after_default, after_draw)
    assert torch.equal(zp1, zp2) and torch.equal(z1, z2) and torch.equal(o1, o2)
    lp = net.infer(mel, ln, noise_scale=0.667, eps=eps)[2][3]
    rec = ((zp1 - mp1) / (torch.exp(lp) * 0.667))                      # the draw, recovered as SURVEY.md 8c describes
    m = (torch.arange(Tn, device="cuda")[None, :] < ln[:, None])[:, None, :].expand_as(rec)
    assert (rec - eps)[m].abs().max().item() <= 1e-4
    torch.manual_seed(4243)
    assert not torch.equal(net.infer(mel, ln, noise_scale=0.667)[0], o1)


@pytest.mark.parametrize("name", list(cases.REF_LARGE_CASES))
def test_full_size_vs_reference_fixture(M, net, name):
    """The throughput kernels against numbers the REFERENCE produced: `tests/golden/make_golden.py` ran
    /root/reference's SynthesizerTrn.infer (models.py:331-339) on the bench batch itself (16 x 512, seed 1001) and on a
    1 x 4096 utterance; the fixtures hold every 64th waveform sample, float64 sums / energies of every 256-sample frame (so
    every output sample is covered), per-utterance sums of the latents and z of two utterances.  At these sizes `infer` runs
    the grouped / pair / merged-accumulate Winograd F(4,3)/F(4,4) launches, the F(4,2) upsamplers and the fused F(2,5) WN
    layers - none of which the small reference fixtures select."""
    c = cases.REF_LARGE_CASES[name]
    mel, ln, eps = cases.large_inputs(name)
    M.native.stats_reset()
    o, mask, (z, z_p, m_p, logs_p) = net.infer(T(mel).cuda(), T(ln).cuda(), noise_scale=c["noise_scale"], eps=T(eps).cuda())
    st = M.native.stats_get()
    assert st["executed_flops"] < 0.6 * st["conv_flops"], "the Winograd throughput kernels did not run at this size"
    assert bool(mask.all())
    d = cases.large_digest(c, o.cpu().numpy(), z.cpu().numpy(), z_p.cpu().numpy(), m_p.cpu().numpy(), logs_p.cpu().numpy())
    g = cases.golden("ref_" + name)
    check(name + " waveform subsample", d["o_sub"], g["o_sub"])
    err = (d["o_sub"].astype(np.float64) - g["o_sub"])
    rms, ref = float(np.sqrt((err ** 2).mean())), float(np.sqrt((g["o_sub"].astype(np.float64) ** 2).mean()))
    fs = float(np.abs(d["o_frame_sum"] - g["o_frame_sum"]).max())
    fe = float((np.abs(d["o_frame_sumsq"] - g["o_frame_sumsq"]) / (g["o_frame_sumsq"] + 1e-3)).max())
    print(f"{name} vs REFERENCE: subsample rms err {rms:.3e} (rel {rms / ref:.2e}), frame sums max {fs:.2e}, frame energies rel {fe:.2e}")
    assert rms <= 1e-3 and rms / ref <= 1e-4, (rms, rms / ref)
    assert fs <= 1e-3 and fe <= 1e-4, (fs, fe)
    check(name + " z rows", d["z_rows"], g["z_rows"], 5e-5, 1e-4)
    for nm in ("z", "z_p", "m_p", "logs_p"):
        assert np.abs(d[nm + "_sum"] - g[nm + "_sum"]).max() <= 5e-2, nm
        assert (np.abs(d[nm + "_sumsq"] - g[nm + "_sumsq"]) / g[nm + "_sumsq"]).max() <= 2e-5, nm


def test_variant_batch_pins_kernel_choice(M, net):
    """svoc_set_variant_batch(n): kernel variants are chosen as if the batch held n utterances, so a shard of a job
    reproduces the whole job's bits ("8-GPU output == 1-GPU output bitwise").  16 x 200 frames: alone, a
    2-utterance shard takes the short-input variants (K-split convolutions, unfused WN layers, per-chain streams); pinned to
    16 it must equal rows of the 16-utterance run bit for bit."""
    Bn, Tn = 16, 200
    mel = T(sw.synthetic_mel(77, Bn, Tn)).cuda(); eps = T(sw.synthetic_eps(77, Bn, Tn)).cuda()
    ln = torch.full((Bn,), Tn, dtype=torch.int64).cuda(); ln[5] = 131
    M.native.profile_enable(True)                 # keeps every call on direct launches (graph replay is bit-identical anyway)
    try:
        full = net.infer(mel, ln, noise_scale=0.667, eps=eps)[0]
        assert M.native.set_variant_batch(0) == 0
        loose = net.infer(mel[4:6], ln[4:6], noise_scale=0.667, eps=eps[4:6])[0]
        with M.native.variant_batch(Bn):
            pinned = net.infer(mel[4:6], ln[4:6], noise_scale=0.667, eps=eps[4:6])[0]
            pinned1 = net.infer(mel[15:16], ln[15:16], noise_scale=0.667, eps=eps[15:16])[0]
        assert M.native.set_variant_batch(0) == 0     # the context manager restored the default
    finally:
        M.native.profile_enable(False)
    assert torch.equal(pinned, full[4:6]) and torch.equal(pinned1, full[15:16])
    assert (loose - full[4:6]).abs().max().item() <= BATCH_TOL


def test_captured_plans_are_keyed_on_the_variant_batch(M, net):
    """Kernel variants are frozen into a captured plan, so the plan key carries the effective variant batch: a shard
    shape whose plan was captured under the default setting must NOT be replayed under variant_batch(n), and the other way
    round.  No profiler here: every call goes through the plan cache.  2 x 200 frames alone take the short-input variants;
    pinned to 16 they must reproduce rows of the 16-utterance job bit for bit, also after a loose plan for the same shape
    exists - and the loose calls must keep their own bits afterwards."""
    Bn, Tn = 16, 200
    mel = T(sw.synthetic_mel(78, Bn, Tn)).cuda(); eps = T(sw.synthetic_eps(78, Bn, Tn)).cuda()
    ln = torch.full((Bn,), Tn, dtype=torch.int64).cuda(); ln[5] = 131
    ns = 0.4321                                                   # no other test shares these plan keys
    M.native.profile_enable(True)
    full = net.infer(mel, ln, noise_scale=ns, eps=eps)[0].clone()
    loose_direct = net.infer(mel[4:6], ln[4:6], noise_scale=ns, eps=eps[4:6])[0].clone()
    M.native.profile_enable(False)
    loose = [net.infer(mel[4:6], ln[4:6], noise_scale=ns, eps=eps[4:6])[0].clone() for _ in range(3)]      # 2nd call captures, 3rd replays
    assert all(torch.equal(o, loose_direct) for o in loose)
    with M.native.variant_batch(Bn):
        pinned = [net.infer(mel[4:6], ln[4:6], noise_scale=ns, eps=eps[4:6])[0].clone() for _ in range(3)]  # its own plan
    assert all(torch.equal(o, full[4:6]) for o in pinned), "a plan captured under the default variants served a pinned call"
    again = net.infer(mel[4:6], ln[4:6], noise_scale=ns, eps=eps[4:6])[0]
    assert torch.equal(again, loose_direct), "a plan captured under a pinned batch served a default call"
    assert (loose_direct - full[4:6]).abs().max().item() <= BATCH_TOL


def test_plan_cache_many_shapes_never_syncs_the_device(M, net):
    """A serving process sees many distinct short lengths.  First sights only bump a counter (no plan, no capture); a
    shape earns a captured plan on its second call; beyond 32 plans the least recently used is retired WITHOUT a device
    synchronisation (svoc_synth_plan_stats: waits == 0 here because retired plans have long completed); results stay
    equal to direct launches throughout."""
    import ctypes
    N = M.native
    h = net._native()
    def stats():
        out = (ctypes.c_int64 * 5)()
        N.check(N.lib().svoc_synth_plan_stats(h, out))
        return list(out)
    base = stats()                                                # (noise_scale 0.123: no other test shares these plan keys)
    lens = list(range(40, 40 + 2 * 40, 2))                        # 40 distinct T
    data = {t: (T(sw.synthetic_mel(900 + t, 1, t)).cuda(), torch.tensor([t]).cuda(), T(sw.synthetic_eps(900 + t, 1, t)).cuda()) for t in lens}
    first = {}
    for t in lens:                                                # first sight: counters only
        mel, ln, eps = data[t]
        first[t] = net.infer(mel, ln, noise_scale=0.123, eps=eps)[0].clone()
    s1 = stats()
    assert s1[1] == base[1] and s1[0] == base[0], (base, s1)      # no capture, no new plan
    for t in lens:                                                # second sight: capture + replay
        mel, ln, eps = data[t]
        assert torch.equal(net.infer(mel, ln, noise_scale=0.123, eps=eps)[0], first[t]), t
    s2 = stats()
    assert s2[1] - base[1] == len(lens) and s2[0] <= 32 and s2[2] - base[2] >= len(lens) - 32, (base, s2)
    torch.cuda.synchronize()
    for t in lens[-8:]:                                           # the most recent ones are still planned: pure replay
        mel, ln, eps = data[t]
        assert torch.equal(net.infer(mel, ln, noise_scale=0.123, eps=eps)[0], first[t]), t
    s3 = stats()
    assert s3[1] == s2[1] and s3[3] == base[3], (s2, s3)           # no re-capture, and no eviction ever waited


def test_generator_mixed_dilation_orders_vs_oracle(M):
    """resblock_dilation_sizes that differ between the chains ([[1,3,5],[3,5,1],[5,1,3]]): a grouped MRF launch carries ONE
    dilation for its three members, so such a configuration must not take the grouped plan (it silently used the first
    chain's dilations before).  C = 128 / 64 stages at a length where the grouped Winograd launches would apply."""
    rds = [[1, 3, 5], [3, 5, 1], [5, 1, 3]]
    c = dict(initial_channel=32, resblock="1", rks=[3, 7, 11], rds=rds, ur=[4, 4], uic=256, uks=[8, 8], gin=0)
    sd = sw.fill_state_dict(cases.generator_shapes(c), 7711, 1.0)
    m = load(M.models.Generator(c["initial_channel"], c["resblock"], c["rks"], c["rds"], c["ur"], c["uic"], c["uks"], gin_channels=0), sd)
    x = T(cases.rnd(7711, "x", (4, 32, 600), 1.0))
    y = m(x.cuda())
    with torch.no_grad():
        ref = O.generator(sdT(sd), x, prefix="", resblock="1", resblock_kernel_sizes=c["rks"], resblock_dilation_sizes=rds,
                          upsample_rates=c["ur"], upsample_kernel_sizes=c["uks"])
    check("generator mixed dilations", y, ref, 5e-5, 1e-4)


@pytest.mark.parametrize("what", ["one_chain", "two_chains", "k5_resblocks"])
def test_fallback_shapes(M, what):
    """Paths that a SHAPE reaches and that, until round 5, were also behind environment switches (now removed with the A/B arms that only a
    switch could reach): a Generator with ONE ResBlock chain per stage (everything on the caller's stream: no fork / join, no grouped launches;
    was SVOC_STREAMS=0), one with TWO chains at a length where the grouped launches apply (the chains' last convolutions accumulate one by one:
    the merged accumulate launch needs k = 3 / 7 / 11; was SVOC_W4_ACCUM=0), and ResBlocks with k = 5 at C = 64 / 32 (no Winograd form, and since
    the generic fused kernel was deleted each iteration runs as two direct-form convolutions).  Against the oracle."""
    cfg = {"one_chain": dict(rks=[7], rds=[[1, 3, 5]], B=3, T=300),
           "two_chains": dict(rks=[3, 7], rds=[[1, 3, 5], [1, 3, 5]], B=8, T=1600),
           "k5_resblocks": dict(rks=[5, 3], rds=[[1, 2, 4], [1, 3, 5]], B=2, T=160)}[what]
    c = dict(initial_channel=32, resblock="1", rks=cfg["rks"], rds=cfg["rds"], ur=[4, 2, 2], uic=256, uks=[8, 4, 4], gin=0)
    sd = sw.fill_state_dict(cases.generator_shapes(c), 7750 + len(what), 1.0)
    m = load(M.models.Generator(c["initial_channel"], c["resblock"], c["rks"], c["rds"], c["ur"], c["uic"], c["uks"], gin_channels=0), sd)
    x = T(cases.rnd(7751, "x", (cfg["B"], 32, cfg["T"]), 1.0))
    M.native.stats_reset()
    y = m(x.cuda())
    st = M.native.stats_get()
    with torch.no_grad():
        ref = O.generator(sdT(sd), x, prefix="", resblock="1", resblock_kernel_sizes=c["rks"], resblock_dilation_sizes=c["rds"],
                          upsample_rates=c["ur"], upsample_kernel_sizes=c["uks"])
    check("generator " + what, y, ref, 5e-5, 1e-4)
    if what == "two_chains":
        assert st["executed_flops"] < 0.8 * st["conv_flops"], "the grouped Winograd launches did not run"


def test_generator_winograd_upsamplers_vs_oracle(M):
    """Stride-8 and stride-2 upsamplers inside the decoder at a shape that passes the F(4,2) kernel's gate, with an input length
    that is NOT a multiple of four (row stride != length; the window tiles then carry the column q = L themselves) at the first
    upsampler and one that is (the separate tail launch) at the second."""
    c = dict(initial_channel=32, resblock="1", rks=[3, 7], rds=[[1, 3, 5], [1, 3, 5]], ur=[8, 2], uic=256, uks=[16, 4], gin=0)
    sd = sw.fill_state_dict(cases.generator_shapes(c), 7723, 1.0)
    m = load(M.models.Generator(c["initial_channel"], c["resblock"], c["rks"], c["rds"], c["ur"], c["uic"], c["uks"], gin_channels=0), sd)
    x = T(cases.rnd(7723, "x", (20, 32, 403), 1.0))
    y = m(x.cuda())
    with torch.no_grad():
        ref = O.generator(sdT(sd), x, prefix="", resblock="1", resblock_kernel_sizes=c["rks"], resblock_dilation_sizes=c["rds"],
                          upsample_rates=c["ur"], upsample_kernel_sizes=c["uks"])
    check("generator F(4,2) upsamplers", y, ref, 5e-5, 1e-4)


def test_generator_dilated_row_tails_vs_oracle(M):
    """Round 6: a row of L outputs has D * ceil(L / 4D) windows of a dilated Winograd convolution; at L = 4096 / 16384 (T a power of two) the partial last
    q block's D windows are alone in one more column tile per row (33 instead of 32).  Where dropping that tile saves a round of the persistent workgroups
    the grouped launch stops at the last full tile and conv_wino4_tail_kernel computes the dropped windows' outputs in direct form (csrc/conv_wino4.hip).
    A decoder with C = 128 / 64 stages at 8 x 1024 frames: both dilated launches of both stages take that path (four more small launches than with
    SVOC_W4_TAIL=0 - counted), window-major hand-over included, and the waveform must match the oracle; 8 x 1000 frames (no such tile) must not take it."""
    c = dict(initial_channel=32, resblock="1", rks=[3, 7, 11], rds=[[1, 3, 5]] * 3, ur=[4, 4], uic=256, uks=[8, 8], gin=0)
    sd = sw.fill_state_dict(cases.generator_shapes(c), 7741, 1.0)
    m = load(M.models.Generator(c["initial_channel"], c["resblock"], c["rks"], c["rds"], c["ur"], c["uic"], c["uks"], gin_channels=0), sd)
    counts = {}
    for Tn in (1024, 1000):
        x = T(cases.rnd(7741 + Tn, "x", (8, 32, Tn), 1.0))
        M.native.stats_reset()
        y = m(x.cuda())
        torch.cuda.synchronize()
        counts[Tn] = M.native.stats_get()["other_launches"]
        pick = [0, 7]
        with torch.no_grad():
            ref = O.generator(sdT(sd), x[pick], prefix="", resblock="1", resblock_kernel_sizes=c["rks"], resblock_dilation_sizes=c["rds"],
                              upsample_rates=c["ur"], upsample_kernel_sizes=c["uks"])
        check(f"generator row tails T{Tn}", y[pick], ref, 5e-5, 1e-4)
        # the last columns of every row are the tail kernel's: compare them on their own
        check(f"generator row tails T{Tn}, last 64 samples", y[pick][..., -64:], ref[..., -64:], 5e-5, 1e-4)
    if not any(k.startswith("SVOC_") and k not in ("SVOC_LIB", "SVOC_VARIANT_JOBS") for k in os.environ):      # (a variant run may take other kernels)
        assert counts[1024] - counts[1000] == 4, counts


def test_generator_conv_pre_winograd_vs_oracle(M):
    """Round 6: conv_pre (k = 7, 192 -> 512 channels at iitp_base) runs on the F(4,4) Winograd kernel when nothing masks or conditions its input and the launch
    has half a tile per CU (csrc/engine.hip Generator::forward).  A Generator with iitp_base's conv_pre (192 -> 512) and ONE upsampler stage, at a batch that
    passes the gate: the event profiler must name the Winograd launch, and the output must match the oracle; a length that is not a multiple of four takes the
    direct kernel (the F(4,x) kernels want whole windows) and must match as well."""
    c = dict(initial_channel=192, resblock="1", rks=[3], rds=[[1, 3, 5]], ur=[2], uic=512, uks=[4], gin=0)
    sd = sw.fill_state_dict(cases.generator_shapes(c), 7731, 1.0)
    m = load(M.models.Generator(c["initial_channel"], c["resblock"], c["rks"], c["rds"], c["ur"], c["uic"], c["uks"], gin_channels=0), sd)
    for Tn, want in ((256, True), (254, False)):
        x = T(cases.rnd(7731 + Tn, "x", (16, 192, Tn), 1.0))
        M.native.profile_enable(True)
        y = m(x.cuda())
        torch.cuda.synchronize()
        rep = M.native.profile_report()
        M.native.profile_enable(False)
        ran = any(l.startswith("wino4 ") and "Ci192" in l and "Co512" in l and "k7" in l for l in rep.splitlines())
        if "SVOC_WINO" not in os.environ and "SVOC_WINO_F4" not in os.environ:
            assert ran == want, rep
        pick = [0, 7, 15]
        with torch.no_grad():
            ref = O.generator(sdT(sd), x[pick], prefix="", resblock="1", resblock_kernel_sizes=c["rks"], resblock_dilation_sizes=c["rds"],
                              upsample_rates=c["ur"], upsample_kernel_sizes=c["uks"])
        check(f"generator conv_pre T{Tn}", y[pick], ref, 5e-5, 1e-4)


def test_infer_long_form_tiling(M, net):
    """Long input (T=1024, B=1) against the oracle over the WHOLE waveform: many time tiles and every halo.  (The full C5
    shape, 8 x 4096, is test_c5_full_size below.)"""
    Tn = 1024
    mel = sw.synthetic_mel(4242, 1, Tn); eps = sw.synthetic_eps(4242, 1, Tn)
    ln = np.array([Tn], dtype=np.int64)
    o, *_ = net.infer(T(mel).cuda(), T(ln).cuda(), noise_scale=0.667, eps=T(eps).cuda())
    # the oracle's waveform is deterministic (seeded inputs and weights); the variant runs (tests/test_gpu_variants.py) call this
    # test once per process, so it is computed once per box and kept in the temp directory
    import os, tempfile
    cache = os.path.join(tempfile.gettempdir(), f"svoc_oracle_longform_{Tn}_4242.npy")
    if os.path.isfile(cache):
        o_ref = T(np.load(cache))
    else:
        with torch.no_grad():
            o_ref, *_ = O.infer(sdT(cases.full_model_weights()), T(mel), T(ln), T(eps), 0.667)
        np.save(cache + ".tmp.npy", o_ref.numpy()); os.replace(cache + ".tmp.npy", cache)
    err = (o.cpu() - o_ref).numpy()
    rms, ref = float(np.sqrt((err ** 2).mean())), float(o_ref.pow(2).mean().sqrt())
    assert rms <= 1e-3 and rms / ref <= 1e-4, (rms, rms / ref)


def test_infer_chunked_vs_oracle_and_one_shot(M, net, tmp_path):
    """SURVEY §8 f3, pinned against the ORACLE (not the HIP path itself): time tiling with a receptive-field halo on
    T = 1500 frames equals the CPU oracle's one-shot waveform within the path's fp32 tolerance, and equals the HIP
    one-shot run to rounding (kernel variants are chosen per shape, so bit equality is not promised).  Then the wav
    writer: utils.infer_to_wavs writes what scipy reads back."""
    from smart_vocoder_amd import utils
    Tn = 1500
    mel = sw.synthetic_mel(777, 2, Tn); eps = sw.synthetic_eps(777, 2, Tn)
    ln = np.array([Tn, 1100], dtype=np.int64)
    tiled = net.infer_chunked(T(mel).cuda(), T(ln).cuda(), chunk_frames=400, noise_scale=0.667, eps=T(eps).cuda())
    with torch.no_grad():
        o_ref, *_ = O.infer(sdT(cases.full_model_weights()), T(mel[:1]), T(ln[:1]), T(eps[:1]), 0.667)
    err = (tiled[:1].cpu() - o_ref).numpy()
    rms, ref = float(np.sqrt((err ** 2).mean())), float(o_ref.pow(2).mean().sqrt())
    assert rms <= 1e-3 and rms / ref <= 1e-4, (rms, rms / ref)
    full = net.infer(T(mel).cuda(), T(ln).cuda(), noise_scale=0.667, eps=T(eps).cuda())[0]
    assert (full - tiled).abs().max().item() <= 1e-5
    paths = [str(tmp_path / "a.wav"), str(tmp_path / "sub" / "b.wav")]
    o = utils.infer_to_wavs(net, T(mel).cuda(), T(ln).cuda(), paths, sampling_rate=22050, chunk_frames=400, noise_scale=0.667,
                            eps=T(eps).cuda())
    from scipy.io.wavfile import read
    for b, pth in enumerate(paths):
        sr, data = read(pth)
        assert sr == 22050 and data.dtype == np.float32 and data.shape == (int(ln[b]) * 256,)
        assert np.array_equal(data, o[b, 0, :int(ln[b]) * 256].cpu().numpy())


def test_infer_chunked_large_batch(M, net):
    """chunked == one-shot at a batch where one-shot and the chunks take different WN kernel variants (B=16, T=2048:
    one-shot NR=2 tiles without K split, 1024+halo chunks the K-split kernel): equal to fp32 rounding."""
    Bn, Tn = 16, 2048
    mel = T(sw.synthetic_mel(778, Bn, Tn)).cuda(); eps = T(sw.synthetic_eps(778, Bn, Tn)).cuda()
    ln = torch.full((Bn,), Tn, dtype=torch.int64).cuda(); ln[5] = 1234
    full = net.infer(mel, ln, noise_scale=0.667, eps=eps)[0]
    tiled = net.infer_chunked(mel, ln, chunk_frames=1024, noise_scale=0.667, eps=eps)
    d = (full - tiled).abs().max().item()
    assert d <= 1e-5, d


def test_c5_full_size(M, net):
    """BASELINE.json configs[4]: 8 x 4096-frame mels (1 M-sample rows, 9.7 GB workspace, 33 k tiles per launch).
    (a) bit-identical between two runs; (b) every checked utterance equals its own B=1 run to fp32 rounding (the WN
    kernel variant depends on B*T); (c) finite, inside tanh's range, right shape; (d) ORACLE: the last 64 frames of two
    utterances (one ragged) against the CPU oracle run on the last 64+192 frames (receptive field 128 frames); (e) ORACLE:
    64 frames from the MIDDLE of a third utterance (frames 2000..2063: tiles far from both edges, every kernel's interior
    path) against the oracle run on those frames with 192 frames of real context on either side."""
    Bn, Tn = 8, 4096
    mel = sw.synthetic_mel(1005, Bn, Tn); eps = sw.synthetic_eps(1005, Bn, Tn)
    ln = np.full((Bn,), Tn, dtype=np.int64); ln[2] = 3001
    melc, epsc, lnc = T(mel).cuda(), T(eps).cuda(), T(ln).cuda()
    o1, mask, (z, z_p, m_p, logs_p) = net.infer(melc, lnc, noise_scale=0.667, eps=epsc)
    o2 = net.infer(melc, lnc, noise_scale=0.667, eps=epsc)[0]
    assert o1.shape == (Bn, 1, Tn * 256)
    assert torch.equal(o1, o2)                                                     # (a)
    assert torch.isfinite(o1).all() and o1.abs().max().item() <= 1.0               # (c)
    assert float(mask.sum()) == float(ln.sum())
    for b in (0, 2, 7):                                                            # (b)
        ob = net.infer(melc[b:b + 1], lnc[b:b + 1], noise_scale=0.667, eps=epsc[b:b + 1])[0]
        assert (ob[0] - o1[b]).abs().max().item() <= 1e-5, b
    sd = sdT(cases.full_model_weights())
    for b in (7, 2):                                                               # (d)
        # frames beyond an utterance's length are masked to z = 0 but still decoded (the decoder takes no mask,
        # models.py:338), and their bias-driven signal reaches the valid tail: give the oracle the same right context
        end = int(ln[b]); ctx = 256; a = end - ctx; e2 = min(Tn, end + 128)
        with torch.no_grad():
            o_ref, *_ = O.infer(sd, T(mel[b:b + 1, :, a:e2]), torch.tensor([ctx]), T(eps[b:b + 1, :, a:e2]), 0.667)
        got = o1[b, 0, (end - 64) * 256:end * 256].cpu().numpy()
        want = o_ref[0, 0, (ctx - 64) * 256:ctx * 256].numpy()
        err = got - want
        rms, ref = float(np.sqrt((err ** 2).mean())), float(np.sqrt((want ** 2).mean()))
        assert rms <= 1e-3 and rms / ref <= 1e-4, (b, rms, rms / ref)
    b, f0, nf, ctx = 5, 2000, 64, 192                                              # (e)
    a, e2 = f0 - ctx, f0 + nf + ctx
    with torch.no_grad():
        o_ref, *_ = O.infer(sd, T(mel[b:b + 1, :, a:e2]), torch.tensor([e2 - a]), T(eps[b:b + 1, :, a:e2]), 0.667)
    got = o1[b, 0, f0 * 256:(f0 + nf) * 256].cpu().numpy()
    want = o_ref[0, 0, ctx * 256:(ctx + nf) * 256].numpy()
    err = got - want
    rms, ref = float(np.sqrt((err ** 2).mean())), float(np.sqrt((want ** 2).mean()))
    assert rms <= 1e-3 and rms / ref <= 1e-4, ("middle slice", rms, rms / ref)


def test_large_batch_awkward_length_vs_oracle(M, net):
    """9 x 509 frames: large enough for the grouped F(4,3) launches, the window-major hand-over and the merged accumulate launch,
    with stage lengths (4072, 32576, 65152, 130304) that are multiples of neither 12 nor 20 - the last q block of every dilated
    row is partial, edge tiles take the clamped paths.  Oracle: the last 64 frames of a ragged utterance and of a full one, and 64
    frames from the middle of a third, each with 192 frames of real context (receptive field 128)."""
    Bn, Tn = 9, 509
    mel = sw.synthetic_mel(1007, Bn, Tn); eps = sw.synthetic_eps(1007, Bn, Tn)
    ln = np.full((Bn,), Tn, dtype=np.int64); ln[3] = 401
    o1 = net.infer(T(mel).cuda(), T(ln).cuda(), noise_scale=0.667, eps=T(eps).cuda())[0]
    assert o1.shape == (Bn, 1, Tn * 256) and torch.isfinite(o1).all()
    sd = sdT(cases.full_model_weights())
    for b, f0 in ((3, 401 - 64), (8, Tn - 64), (5, 230)):
        end = int(ln[b]); ctx = 192
        a, e2 = max(0, f0 - ctx), min(Tn, f0 + 64 + ctx)
        valid = max(0, min(end, e2) - a)
        with torch.no_grad():
            o_ref, *_ = O.infer(sd, T(mel[b:b + 1, :, a:e2]), torch.tensor([valid]), T(eps[b:b + 1, :, a:e2]), 0.667)
        got = o1[b, 0, f0 * 256:(f0 + 64) * 256].cpu().numpy()
        want = o_ref[0, 0, (f0 - a) * 256:(f0 - a + 64) * 256].numpy()
        err = got - want
        rms, ref = float(np.sqrt((err ** 2).mean())), float(np.sqrt((want ** 2).mean()))
        assert rms <= 1e-3 and rms / ref <= 1e-4, (b, f0, rms, rms / ref)


def test_c3_full_size_with_speaker_conditioning(M):
    """BASELINE.json configs[2] (iitp_base_ms, batch 32 x 512): SynthesizerTrn.infer hard-codes g=None (models.py:332), so
    the speaker-conditioned path is exercised where the reference can run it, at module level, at FULL size:
    ResidualCouplingBlock(g) reverse and Generator(g) with g [32,256,1].  Properties: determinism, per-utterance
    equality with B=1 runs, flow(g) round trip, finiteness; plus two utterances of flow(g) + Generator(g) (one ragged) and one of
    WN(g) against the oracle."""
    Bn, Tn = 32, 512
    sd = cases.full_model_weights()
    flow_sd = {k[len("flow."):]: v for k, v in sd.items() if k.startswith("flow.")}
    dec_sd = {k[len("dec."):]: v for k, v in sd.items() if k.startswith("dec.")}
    c = cases.IITP_MODEL
    flow = load(M.models.ResidualCouplingBlock(192, 192, 5, 1, 8, gin_channels=256), flow_sd)
    dec = load(M.models.Generator(192, c["resblock"], c["resblock_kernel_sizes"], c["resblock_dilation_sizes"], c["upsample_rates"],
                                  c["upsample_initial_channel"], c["upsample_kernel_sizes"], gin_channels=256), dec_sd)
    zp = T(cases.rnd(1003, "zp", (Bn, 192, Tn), 1.0)); g = T(cases.rnd(1003, "g", (Bn, 256, 1), 1.0))
    lens = [Tn] * Bn; lens[4] = 300; lens[17] = 64
    mask = T(cases.lengths_mask(lens, Tn))
    zpc, gc, mc = (zp * mask).cuda(), g.cuda(), mask.cuda()
    z = flow(zpc, mc, g=gc, reverse=True)
    assert torch.equal(z, flow(zpc, mc, g=gc, reverse=True))
    back = flow(z, mc, g=gc, reverse=False)
    assert ((back - zpc).abs().max() / zpc.abs().max()).item() <= 1e-4
    o = dec(z * mc, g=gc)
    assert o.shape == (Bn, 1, Tn * 256) and torch.isfinite(o).all() and o.abs().max().item() <= 1.0
    assert torch.equal(o, dec(z * mc, g=gc))
    for b in (0, 4, 31):
        zb = flow(zpc[b:b + 1], mc[b:b + 1], g=gc[b:b + 1], reverse=True)
        assert (zb[0] - z[b]).abs().max().item() <= 1e-5, b
        ob = dec(z[b:b + 1] * mc[b:b + 1], g=gc[b:b + 1])
        assert (ob[0] - o[b]).abs().max().item() <= BATCH_TOL, b
    sdt = sdT(sd)
    for b in (4, 31):            # a ragged utterance (300 of 512 frames) and the last, full-length one
        with torch.no_grad():
            z_ref = O.flow(sdt, (zp * mask)[b:b + 1], mask[b:b + 1], g[b:b + 1], reverse=True)
            o_ref = O.generator(sdt, z_ref * mask[b:b + 1], g[b:b + 1])
        check(f"c3 flow(g) vs oracle, utterance {b}", z[b:b + 1], z_ref.numpy(), 5e-5, 1e-4)
        err = (o[b:b + 1].cpu() - o_ref).numpy()
        rms, ref = float(np.sqrt((err ** 2).mean())), float(o_ref.pow(2).mean().sqrt())
        assert rms <= 1e-3 and rms / ref <= 1e-4, (b, rms, rms / ref)
    b = 4
    # WN with g at the full C3 column count (the conditioned gate epilogue of the fused WN kernel, NR = 2 tiles)
    wn_sd = {k[len("flow.flows.0.enc."):]: v for k, v in sd.items() if k.startswith("flow.flows.0.enc.")}
    wn = load(M.modules.WN(192, 5, 1, 8, gin_channels=256), wn_sd)
    x = T(cases.rnd(1003, "x", (Bn, 192, Tn), 1.0)) * mask
    y = wn(x.cuda(), mc, g=gc)
    with torch.no_grad():
        y_ref = O.wn(sdT(wn_sd), "", x[b:b + 1], mask[b:b + 1], g[b:b + 1], hidden=192, kernel_size=5, dilation_rate=1, n_layers=8)
    check("c3 WN(g) vs oracle", y[b:b + 1], y_ref.numpy())


def test_small_shape_graph_replay_is_bit_identical(M, net):
    """Short inputs are replayed from a captured hipGraph from their third call on (first call: direct launches, second:
    capture + replay).  The replay reads its inputs through staging copies, so feed different data call by call:
    results must equal the direct (uncaptured) launches bit for bit, ragged lengths and max_len included."""
    Bn, Tn = 2, 100
    outs = {}
    data = {}
    for seed in (41, 42):
        data[seed] = (T(sw.synthetic_mel(seed, Bn, Tn)).cuda(), torch.tensor([Tn, 61 + seed % 7]).cuda(), T(sw.synthetic_eps(seed, Bn, Tn)).cuda())
    for seed in (41, 42):                       # reference results: profiler on => the graph path is bypassed
        M.native.profile_enable(True)
        mel, ln, eps = data[seed]
        outs[seed] = [t.clone() for t in (lambda r: (r[0], r[1], *r[2]))(net.infer(mel, ln, noise_scale=0.667, eps=eps, max_len=90))]
        M.native.profile_enable(False)
    for rep in range(3):
        for seed in (41, 42):
            mel, ln, eps = data[seed]
            r = net.infer(mel, ln, noise_scale=0.667, eps=eps, max_len=90)
            got = (r[0], r[1], *r[2])
            for i, (a, b) in enumerate(zip(got, outs[seed])):
                assert torch.equal(a, b), (rep, seed, i)
    st = M.native.stats_get()
    assert st["conv_launches"] > 0


def test_device_side_give_up_makes_the_waveform_of_that_call_non_finite(M, net):
    """... and through `SynthesizerTrn.infer`, direct launches and a captured plan alike: `o` of the affected call is not finite (what bench.py's
    `finite` check and any caller that looks at its audio sees), the next call raises, later calls are clean."""
    mel, ln, eps = (T(a).cuda() for a in cases.infer_inputs("c1"))
    M.native.check_async_error()
    o = [net.infer(mel, ln, noise_scale=0.667, eps=eps)[0] for _ in range(3)][-1]        # (the third call replays a plan)
    torch.cuda.synchronize()
    assert torch.isfinite(o).all()
    try:
        for _ in range(2):                                           # first sight of the new launch arguments: direct launches; second: a fresh plan
            M.native.debug_persist_control(fault_tile=1, timeout_ms=5, reenable=True)
            ob = net.infer(mel, ln, noise_scale=0.667, eps=eps)[0]
            torch.cuda.synchronize()
            assert not torch.isfinite(ob).all()
            with pytest.raises(RuntimeError, match="persistent WN launch"):
                net.infer(mel, ln, noise_scale=0.667, eps=eps)
        o_layers = net.infer(mel, ln, noise_scale=0.667, eps=eps)[0]
        assert torch.isfinite(o_layers).all()
        assert float((o_layers - o).pow(2).mean().sqrt()) <= 1e-5
    finally:
        M.native.debug_persist_control(reenable=True)
    for _ in range(3):
        assert torch.equal(net.infer(mel, ln, noise_scale=0.667, eps=eps)[0], o)
    torch.cuda.synchronize()
    M.native.check_async_error()


def test_mel_encoder_standalone(M):
    for name, c in cases.MELENC_CASES.items():
        sd = sw.fill_state_dict(cases.melenc_shapes(c["Cout"], c["H"], c["k"], c["n"], c["gin"]), c["seed"])
        m = load(M.models.MelEncoder(c["Cout"], c["H"], 768, c["n"], c["k"], c["dr"], c["gin"]), sd)
        x = T(cases.rnd(c["seed"], "x", (c["B"], 80, c["T"]), 1.0))
        xo, mm, logs, mask = m(x.cuda(), torch.tensor(c["lengths"]).cuda(), g=torch.zeros(1, device="cuda"))   # g is discarded (models.py:36)
        gold = cases.golden(name)
        check(name + ".x", xo, gold["x"]); check(name + ".m", mm, gold["m"]); check(name + ".logs", logs, gold["logs"])
        assert np.array_equal(mask.cpu().numpy(), gold["mask"])


def test_mel_encoder_of_the_synthesizer(M, net):
    """net_g.enc_p(mel, lengths) standalone returns the (m_p, logs_p, x_mask) that infer reports."""
    mel, ln, eps = cases.infer_inputs("ragged")
    g = cases.golden("infer_ragged")
    xo, m_p, logs_p, mask = net.enc_p(T(mel).cuda(), T(ln).cuda())
    check("enc_p m", m_p, g["m_p"]); check("enc_p logs", logs_p, g["logs_p"])
    assert np.array_equal(mask.cpu().numpy(), g["mask"]) and xo.shape == (3, 192, 64)


@pytest.mark.parametrize("name", list(cases.LAYERNORM_CASES))
def test_layer_norm_standalone(M, name):
    c = cases.LAYERNORM_CASES[name]
    m = M.modules.LayerNorm(c["C"])
    m.gamma.data.copy_(T(1.0 + cases.rnd(c["seed"], "gamma", (c["C"],), 0.3)))
    m.beta.data.copy_(T(cases.rnd(c["seed"], "beta", (c["C"],), 0.2)))
    m = m.cuda()
    x = T(cases.rnd(c["seed"], "x", c["shape"], 1.5))
    check(name, m(x.cuda()), cases.golden(name)["y"], 1e-5, 1e-5)


@pytest.mark.parametrize("name", list(cases.CONVFLOW_G_CASES))
def test_convflow_with_g(M, name):
    c = cases.CONVFLOW_G_CASES[name]
    sd = sw.fill_state_dict(cases.convflow_shapes(c["Cin"], c["F"], c["k"], c["n"]), c["seed"], 2.0)
    m = load(M.modules.ConvFlow(c["Cin"], c["F"], c["k"], c["n"]), sd)
    x = T(cases.rnd(c["seed"], "x", (c["B"], c["Cin"], c["T"]), 2.5))
    mask = T(cases.lengths_mask(c["lengths"], c["T"]))
    g = T(cases.rnd(c["seed"], "g", (c["B"], c["F"], c["gT"]), 0.7))
    r = m(x.cuda(), mask.cuda(), g=g.cuda(), reverse=c["reverse"])
    gold = cases.golden(name)
    if c["reverse"]:
        check(name, r, gold["y"], 1e-3, 0)
    else:
        check(name, r[0], gold["y"], 1e-3, 0)
        check(name + " logdet", r[1], gold["logdet"], 2e-2, 1e-3)


@pytest.mark.parametrize("name", list(cases.SPLINE_EXTRA_CASES))
def test_spline_no_tails_and_minimums(M, name):
    c = cases.SPLINE_EXTRA_CASES[name]
    x, uw, uh, ud = cases.spline_extra_inputs(name)
    mw, mh, md = c["mins"]
    kw = dict(min_bin_width=mw, min_bin_height=mh, min_derivative=md)
    if c["tails"] is not None:
        kw.update(tails="linear", tail_bound=5.0)
    y, lad = M.transforms.piecewise_rational_quadratic_transform(T(x).cuda(), T(uw).cuda(), T(uh).cuda(), T(ud).cuda(),
                                                                 inverse=c["inverse"], **kw)
    gold = cases.golden(name)
    ey = np.abs(y.cpu().numpy() - gold["y"]); el = np.abs(lad.cpu().numpy() - gold["logabsdet"])
    assert ey.max() <= 1e-3, ey.max()
    assert (el > 1e-2).mean() <= 0.01 and np.median(el) <= 1e-5, ((el > 1e-2).mean(), np.median(el))
    if c["tails"] is None:      # the reference refuses inputs outside [0,1] (transforms.py:105-106)
        with pytest.raises(ValueError):
            M.transforms.piecewise_rational_quadratic_transform(T(x).cuda() + 0.5, T(uw).cuda(), T(uh).cuda(), T(ud).cuda())
    with pytest.raises(ValueError):
        M.transforms.piecewise_rational_quadratic_transform(T(x).cuda(), T(uw).cuda(), T(uh).cuda(), T(ud).cuda(), min_bin_width=0.2,
                                                            **{k: v for k, v in kw.items() if k.startswith("tail")})


def test_reference_written_checkpoint_runs(M):
    """SURVEY 8 f2 on the GPU: load tests/golden/G_7.pth (written by the reference's utils.save_checkpoint) with
    utils.load_checkpoint and reproduce the reference's own output for that Generator (fixture gen_small_rb2)."""
    import os
    from smart_vocoder_amd import utils
    rc = cases.REF_CHECKPOINT
    c = cases.GENERATOR_CASES[rc["case"]]
    m = M.models.Generator(c["initial_channel"], c["resblock"], c["rks"], c["rds"], c["ur"], c["uic"], c["uks"], gin_channels=c["gin"]).cuda().eval()
    x = T(cases.rnd(c["seed"], "x", (c["B"], c["initial_channel"], c["T"]), 1.0)).cuda()
    y0 = m(x)                                                   # default-init weights: a handle exists before the load
    _, _, lr, it = utils.load_checkpoint(os.path.join(cases.GOLDEN_DIR, rc["file"]), m, None)
    assert it == rc["iteration"]
    y = m(x)
    assert not torch.equal(y, y0)                               # the cached handle was rebuilt from the loaded weights
    check("G_7.pth", y, cases.golden(rc["case"])["y"], 5e-5, 1e-4)


def test_wrong_channel_counts_are_refused(M):
    """The reference raises a conv shape error; the library would index out of bounds, so the Python surface refuses."""
    sd = sw.fill_state_dict(cases.wn_shapes(64, 3, 2, 16), 1)
    wn = load(M.modules.WN(64, 3, 1, 2, gin_channels=16), sd)
    mask = torch.ones(1, 1, 40, device="cuda")
    with pytest.raises(ValueError):
        wn(torch.zeros(1, 32, 40, device="cuda"), mask)
    with pytest.raises(ValueError):
        wn(torch.zeros(1, 64, 40, device="cuda"), mask, g=torch.zeros(1, 8, 1, device="cuda"))
    with pytest.raises(ValueError):
        wn(torch.zeros(2, 64, 40, device="cuda"), mask.expand(2, 1, 40), g=torch.zeros(3, 16, 1, device="cuda"))
    rb = load(M.modules.ResBlock1(32, 3), sw.fill_state_dict(cases.resblock1_shapes(32, 3), 2, 1.0))
    with pytest.raises(ValueError):
        rb(torch.zeros(1, 64, 100, device="cuda"))
    c = cases.GENERATOR_CASES["gen_small_g"]
    gen = load(M.models.Generator(c["initial_channel"], c["resblock"], c["rks"], c["rds"], c["ur"], c["uic"], c["uks"], gin_channels=c["gin"]),
               sw.fill_state_dict(cases.generator_shapes(c), c["seed"], 1.0))
    with pytest.raises(ValueError):
        gen(torch.zeros(1, c["initial_channel"] + 1, 10, device="cuda"))
    with pytest.raises(ValueError):
        gen(torch.zeros(1, c["initial_channel"], 10, device="cuda"), g=torch.zeros(1, c["gin"] + 3, 1, device="cuda"))
    fl = M.models.ResidualCouplingBlock(192, 192, 5, 1, 2, gin_channels=0).cuda()
    with pytest.raises(ValueError):
        fl(torch.zeros(1, 96, 20, device="cuda"), torch.ones(1, 1, 20, device="cuda"))


def test_deepcopy_and_invalidate(M):
    """copy.deepcopy after the first forward (the ctypes handle is dropped, not copied) and the documented invalidate()
    for writes through .data that bypass the version counter."""
    import copy
    sd = sw.fill_state_dict(cases.resblock1_shapes(32, 3), 77, 1.0)
    rb = load(M.modules.ResBlock1(32, 3), sd)
    x = T(cases.rnd(77, "x", (1, 32, 300), 0.5)).cuda()
    y = rb(x)
    rb2 = copy.deepcopy(rb)
    assert torch.equal(rb2(x), y)
    rb2.convs1[0].bias.data.add_(1.0)           # bypasses Parameter._version
    rb2.invalidate()
    assert not torch.equal(rb2(x), y) and torch.equal(rb(x), y)
