// Module-level ops that the reference defines next to the inference path but never
// instantiates from SynthesizerTrn: DDSConv, ConvFlow and the
// piecewise rational-quadratic spline.  The dense 1x1 convolutions reuse the MFMA
// kernel; depthwise conv + LayerNorm(C) + GELU and the spline are HBM-bound
// elementwise/reduction kernels with time-contiguous (coalesced) accesses.
#include "svoc_internal.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace svoc {

static ConvArgs mk_args2() {
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.pre_slope = 1.0f;
  a.split_row = 1 << 30;
  a.mode = EPI_PLAIN;
  a.out[0].nrows = 1 << 30;
  a.out[1].nrows = 1 << 30;
  a.out[0].div = 1.0f;
  return a;
}

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// ---------------------------------------------------------------------------------------------
// [optional depthwise dilated conv on x*mask] -> LayerNorm over channels -> GELU [-> x += . ; * mask]
//   MODE 0: y = gelu(LN(dwconv(x*mask)))                      (modules.py:99-101)
//   MODE 1: x = x + gelu(LN(y_in)); if last: x *= mask        (modules.py:103-108)
//   MODE 2: y = LN(x)   (standalone modules.LayerNorm.forward, modules.py:28-32; mask unused)
// One block owns TT time steps x all C channels; the tile lives in LDS between the passes.
template <int MODE>
__global__ void __launch_bounds__(256) dds_ln_gelu_kernel(const float* __restrict__ src, long long s_bs, int s_ld,
                                                          const float* __restrict__ mask, long long mask_bs,
                                                          const float* __restrict__ dw_w, const float* __restrict__ dw_b,
                                                          int K, int dil, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, float* __restrict__ dst,
                                                          long long d_bs, int d_ld, int C, int T, int TT, int last) {
  extern __shared__ float tile[];          // [C][TT]
  float* red = tile + (size_t)C * TT;      // [2][256]
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * TT;
  const int tl = threadIdx.x % TT;
  const int cg = threadIdx.x / TT;
  const int ncg = 256 / TT;
  const int t = t0 + tl;
  const bool tv = t < T;
  const float* sb = src + (long long)b * s_bs;
  const float* mb = MODE == 2 ? nullptr : mask + (long long)b * mask_bs;
  float s1 = 0.f;
  for (int c = cg; c < C; c += ncg) {
    float v = 0.f;
    if (tv) {
      if (MODE == 0) {
        const int pad = (K * dil - dil) / 2;
        v = dw_b[c];
        for (int j = 0; j < K; ++j) {
          const int tt = t + j * dil - pad;
          if (tt >= 0 && tt < T) v = fmaf(dw_w[c * K + j], sb[(long long)c * s_ld + tt] * mb[tt], v);
        }
      } else {
        v = sb[(long long)c * s_ld + t];
      }
    }
    tile[(size_t)c * TT + tl] = v;
    s1 += v;
  }
  red[threadIdx.x] = s1;
  __syncthreads();
  float mean = 0.f;
  for (int q = 0; q < ncg; ++q) mean += red[q * TT + tl];
  mean /= (float)C;
  float s2 = 0.f;
  for (int c = cg; c < C; c += ncg) {
    const float dv = tile[(size_t)c * TT + tl] - mean;
    s2 += dv * dv;
  }
  red[256 + threadIdx.x] = s2;
  __syncthreads();
  float var = 0.f;
  for (int q = 0; q < ncg; ++q) var += red[256 + q * TT + tl];
  var /= (float)C;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (!tv) return;
  float* db = dst + (long long)b * d_bs;
  const float mk = MODE == 2 ? 1.0f : mb[t];
  for (int c = cg; c < C; c += ncg) {
    const float ln = (tile[(size_t)c * TT + tl] - mean) * rstd * gamma[c] + beta[c];
    const float v = MODE == 2 ? ln : gelu_erf(ln);
    float* dp = db + (long long)c * d_ld + t;
    if (MODE == 0 || MODE == 2) {
      *dp = v;
    } else {
      float r = *dp + v;
      if (last) r *= mk;
      *dp = r;
    }
  }
}

// ---- vectorised form of the three modes (rows 16-byte aligned, C <= 256): a block owns 64 time steps x all C channels;
// thread (tq, cg) = (tid & 15, tid >> 4) holds the float4 of time steps t0 + 4 tq .. + 3 for channels cg, cg + 16, ... in
// REGISTERS between the passes (twelve independent 16-byte loads in flight per thread at C = 192), so the only LDS traffic
// is the 16-way channel reduction of the LayerNorm sums - and, for MODE 0, the dilated receptive field: (x * mask) with its
// halo [t0 - pad, t0 + 64 + pad) is staged through LDS once with coalesced 16-byte row reads and every tap of the depthwise
// convolution is an LDS read.  Measured at [16,192,4096] (tools/offgraph_bench.py): see profiles/r02_offgraph_kernels.txt.
template <int MODE>
__global__ void __launch_bounds__(256) dds_ln_v2_kernel(const float* __restrict__ src, long long s_bs, int s_ld,
                                                        const float* __restrict__ mask, long long mask_bs,
                                                        const float* __restrict__ dw_w, const float* __restrict__ dw_b, int K, int dil,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                        float* __restrict__ dst, long long d_bs, int d_ld, int C, int T, int last) {
  constexpr int TT = 64, MAXI = 16;
  extern __shared__ __attribute__((aligned(16))) float tile[];
  float4* red = reinterpret_cast<float4*>(tile);           // [2][16 cg][16 tq]
  float* xt = tile + 2 * 256 * 4;                          // MODE 0: [C][XW] x * mask with halo
  const int tid = threadIdx.x;
  const int tq = tid & 15, cg = tid >> 4;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * TT;
  const int t4 = t0 + 4 * tq;
  const bool tv = t4 < T;
  const float* sb = src + (long long)b * s_bs;
  const float* mb = MODE == 2 ? nullptr : mask + (long long)b * mask_bs;
  const int ni = (C - cg + 15) >> 4;
  float4 vals[MAXI];
  if constexpr (MODE == 0) {
    const int pad = (K * dil - dil) / 2, pad4 = (pad + 3) & ~3;
    const int XW = TT + 2 * pad4, XQ = XW >> 2;
    for (int idx = tid; idx < C * XQ; idx += 256) {        // 16-byte groups, rows contiguous in time
      const int c = idx / XQ, q = idx - c * XQ;
      const int t = t0 - pad4 + 4 * q;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const float* row = sb + (long long)c * s_ld;
      if (t >= 0 && t + 3 < T) {
        v = *reinterpret_cast<const float4*>(row + t);
        v.x *= mb[t]; v.y *= mb[t + 1]; v.z *= mb[t + 2]; v.w *= mb[t + 3];
      } else {
        if (t >= 0 && t < T) v.x = row[t] * mb[t];
        if (t + 1 >= 0 && t + 1 < T) v.y = row[t + 1] * mb[t + 1];
        if (t + 2 >= 0 && t + 2 < T) v.z = row[t + 2] * mb[t + 2];
        if (t + 3 >= 0 && t + 3 < T) v.w = row[t + 3] * mb[t + 3];
      }
      *reinterpret_cast<float4*>(xt + (size_t)c * XW + 4 * q) = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      if (i < ni) {
        const int c = cg + 16 * i;
        const float bv = dw_b[c];
        float4 v = make_float4(bv, bv, bv, bv);
        const float* xr = xt + (size_t)c * XW + pad4 - pad + 4 * tq;
        for (int j = 0; j < K; ++j) {
          const float w = dw_w[c * K + j];
          const float* xp = xr + j * dil;
          v.x = fmaf(w, xp[0], v.x); v.y = fmaf(w, xp[1], v.y); v.z = fmaf(w, xp[2], v.z); v.w = fmaf(w, xp[3], v.w);
        }
        vals[i] = v;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
      if (i < ni) vals[i] = tv ? *reinterpret_cast<const float4*>(sb + (long long)(cg + 16 * i) * s_ld + t4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // LayerNorm over channels: per-thread partial sums over its channels, 16-way combine through LDS (fixed order)
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < MAXI; ++i)
    if (i < ni) { s1.x += vals[i].x; s1.y += vals[i].y; s1.z += vals[i].z; s1.w += vals[i].w; }
  red[cg * 16 + tq] = s1;
  __syncthreads();
  float4 mean = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int q = 0; q < 16; ++q) { const float4 r = red[q * 16 + tq]; mean.x += r.x; mean.y += r.y; mean.z += r.z; mean.w += r.w; }
  const float invC = 1.0f / (float)C;
  mean.x *= invC; mean.y *= invC; mean.z *= invC; mean.w *= invC;
  float4 s2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < MAXI; ++i)
    if (i < ni) {
      const float dx = vals[i].x - mean.x, dy = vals[i].y - mean.y, dz = vals[i].z - mean.z, dw = vals[i].w - mean.w;
      s2.x += dx * dx; s2.y += dy * dy; s2.z += dz * dz; s2.w += dw * dw;
    }
  red[256 + cg * 16 + tq] = s2;
  __syncthreads();
  float4 var = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int q = 0; q < 16; ++q) { const float4 r = red[256 + q * 16 + tq]; var.x += r.x; var.y += r.y; var.z += r.z; var.w += r.w; }
  const float4 rstd = make_float4(1.0f / sqrtf(var.x * invC + eps), 1.0f / sqrtf(var.y * invC + eps), 1.0f / sqrtf(var.z * invC + eps),
                                  1.0f / sqrtf(var.w * invC + eps));
  if (!tv) return;
  float* db = dst + (long long)b * d_bs;
  float4 mk = make_float4(1.f, 1.f, 1.f, 1.f);
  if (MODE == 1 && last) {
    mk.x = mb[t4];
    mk.y = t4 + 1 < T ? mb[t4 + 1] : 0.f; mk.z = t4 + 2 < T ? mb[t4 + 2] : 0.f; mk.w = t4 + 3 < T ? mb[t4 + 3] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    if (i < ni) {
      const int c = cg + 16 * i;
      const float ga = gamma[c], be = beta[c];
      float4 o;
      o.x = (vals[i].x - mean.x) * rstd.x * ga + be; o.y = (vals[i].y - mean.y) * rstd.y * ga + be;
      o.z = (vals[i].z - mean.z) * rstd.z * ga + be; o.w = (vals[i].w - mean.w) * rstd.w * ga + be;
      if (MODE != 2) { o.x = gelu_erf(o.x); o.y = gelu_erf(o.y); o.z = gelu_erf(o.z); o.w = gelu_erf(o.w); }
      float4* dp = reinterpret_cast<float4*>(db + (long long)c * d_ld + t4);
      if (MODE == 1) {
        const float4 r = *dp;
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        if (last) { o.x *= mk.x; o.y *= mk.y; o.z *= mk.z; o.w *= mk.w; }
      }
      *dp = o;
    }
  }
}
// eligibility of the vectorised kernels: 16-byte aligned rows with room for whole float4 groups, at most 256 channels
static bool ln_v2_ok(const void* src, long long s_bs, int s_ld, const void* dst, long long d_bs, int d_ld, int C, int T) {
  const int T4 = (T + 3) & ~3;
  return C <= 256 && s_ld >= T4 && d_ld >= T4 && (s_ld & 3) == 0 && (d_ld & 3) == 0 && (s_bs & 3) == 0 && (d_bs & 3) == 0 &&
         (reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
}

// dst = a (+ g): a [B][rows][a_ld] strided, g NULL or [B][rows][g_T] contiguous with g_T == T or 1 (broadcast over time,
// the speaker-embedding case of DDSConv's `x = x + g`, modules.py:97-98)
__global__ void add2d_kernel(const float* __restrict__ a, long long a_bs, int a_ld, const float* __restrict__ g, int g_T,
                             float* __restrict__ dst, long long d_bs, int d_ld, int rows, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  float v = a[(long long)b * a_bs + (long long)r * a_ld + t];
  if (g) v += g[((long long)b * rows + r) * g_T + (g_T == 1 ? 0 : t)];
  dst[(long long)b * d_bs + (long long)r * d_ld + t] = v;
}

struct DDS {
  int C = 0, K = 0, NL = 0;
  std::vector<std::unique_ptr<PackedConv>> c1x1;
  DevBuf params;     // per layer: dw_w [C*K], dw_b [C], g1 [C], b1 [C], g2 [C], b2 [C]
  DevBuf ws;
  size_t stride() const { return (size_t)C * K + 5 * (size_t)C; }

  int create(int channels, int k, int nl, const TensorTable& tab, const std::string& prefix, hipStream_t st) {
    if (channels <= 0 || k <= 0 || (k % 2) == 0 || nl <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "DDSConv: bad hyper-parameters");
    C = channels; K = k; NL = nl;
    SVOC_TRY(params.ensure(stride() * NL * sizeof(float)));
    for (int i = 0; i < NL; ++i) {
      const std::string s = std::to_string(i);
      const char* names[6] = {"convs_sep.", "convs_sep.", "norms_1.", "norms_1.", "norms_2.", "norms_2."};
      const char* leaf[6] = {".weight", ".bias", ".gamma", ".beta", ".gamma", ".beta"};
      size_t off = stride() * i;
      for (int q = 0; q < 6; ++q) {
        const std::string nm = prefix + names[q] + s + leaf[q];
        const svoc_tensor* t = tab.find(nm);
        if (!t) SVOC_FAIL(SVOC_ERR_MISSING_TENSOR, "missing tensor %s", nm.c_str());
        const size_t n = q == 0 ? (size_t)C * K : (size_t)C;
        size_t have = 1;
        for (int d = 0; d < t->ndim; ++d) have *= (size_t)t->shape[d];
        if (have != n) SVOC_FAIL(SVOC_ERR_SHAPE, "tensor %s has %zu elements, expected %zu", nm.c_str(), have, n);
        SVOC_HIP(hipMemcpyAsync(params.f() + off, t->data, n * sizeof(float), hipMemcpyDeviceToDevice, st));
        off += n;
      }
      PackSpec sp{}; sp.Cin = C; sp.Cout = C; sp.K = 1;
      c1x1.emplace_back(new PackedConv());
      SVOC_TRY(pack_conv_named(*c1x1.back(), sp, tab, prefix + "convs_1x1." + s, st));
    }
    SVOC_HIP(hipStreamSynchronize(st));
    return SVOC_OK;
  }

  int tile_tt() const {
    int tt = 64;
    while (tt > 1 && ((size_t)C * tt + 512) * sizeof(float) > 64 * 1024) tt >>= 1;
    return tt;
  }

  // x [B][C][x_ld] -> y [B][C][y_ld]; g nullable, contiguous [B][C][g_T], g_T == T or 1
  int forward(hipStream_t st, const float* x, long long x_bs, int x_ld, const float* mask, long long mask_bs, const float* g, int g_T,
              float* y, long long y_bs, int y_ld, int B, int T) {
    if (g && g_T != 1 && g_T != T) SVOC_FAIL(SVOC_ERR_SHAPE, "DDSConv: g must have 1 or T=%d frames, got %d", T, g_T);
    const int Tp = round_up(T, 4);
    const long long per = (long long)C * Tp;
    SVOC_TRY(ws.ensure((size_t)(3 * per * B) * sizeof(float)));
    float* xw = ws.f();
    float* y1 = xw + per * B;
    float* y2 = y1 + per * B;
    const int TT = tile_tt();
    const size_t lds = ((size_t)C * TT + 512) * sizeof(float);
    if (lds > 64 * 1024) SVOC_FAIL(SVOC_ERR_UNSUPPORTED, "DDSConv: %d channels do not fit the LayerNorm tile", C);
    // xw = x (+ g)
    hipLaunchKernelGGL(add2d_kernel, dim3((T + 255) / 256, C, B), dim3(256), 0, st, x, x_bs, x_ld, g, g_T, xw, per, Tp, C, T);
    SVOC_HIP(hipGetLastError());
    stats_add_other();
    int d = 1;
    for (int i = 0; i < NL; ++i) {
      const float* P = params.f() + stride() * i;
      const float* dw_w = P; const float* dw_b = dw_w + (size_t)C * K;
      const float* g1 = dw_b + C; const float* b1 = g1 + C; const float* g2 = b1 + C; const float* b2 = g2 + C;
      dim3 grid((T + TT - 1) / TT, B);
      const bool v2 = ln_v2_ok(xw, per, Tp, y1, per, Tp, C, T);
      if (v2) {   // depthwise conv (halo staged through LDS) + LayerNorm + GELU, values in registers
        const int pad4 = (((K * d - d) / 2) + 3) & ~3;
        const size_t lds0 = ((size_t)2 * 256 * 4 + (size_t)C * (64 + 2 * pad4)) * sizeof(float);
        if (lds0 <= 160 * 1024) {
          SVOC_TRY(ensure_max_dyn_lds((const void*)dds_ln_v2_kernel<0>));
          hipLaunchKernelGGL(dds_ln_v2_kernel<0>, dim3((T + 63) / 64, B), dim3(256), lds0, st, xw, per, Tp, mask, mask_bs, dw_w, dw_b, K, d, g1, b1,
                             1e-5f, y1, per, Tp, C, T, 0);
        } else {
          hipLaunchKernelGGL(dds_ln_gelu_kernel<0>, grid, dim3(256), lds, st, xw, per, Tp, mask, mask_bs, dw_w, dw_b, K, d, g1, b1, 1e-5f,
                             y1, per, Tp, C, T, TT, 0);
        }
      } else {
        hipLaunchKernelGGL(dds_ln_gelu_kernel<0>, grid, dim3(256), lds, st, xw, per, Tp, mask, mask_bs, dw_w, dw_b, K, d, g1, b1, 1e-5f,
                           y1, per, Tp, C, T, TT, 0);
      }
      SVOC_HIP(hipGetLastError());
      stats_add_other();
      ConvArgs a = mk_args2();
      a.x = y1; a.x_bs = per; a.x_ld = Tp; a.Lin = T; a.Ncols = T;
      a.out[0].y = y2; a.out[0].y_bs = per; a.out[0].y_ld = Tp; a.out[0].nrows = C;
      SVOC_TRY(launch_conv(*c1x1[i], a, B, st));
      if (v2)
        hipLaunchKernelGGL(dds_ln_v2_kernel<1>, dim3((T + 63) / 64, B), dim3(256), (size_t)2 * 256 * 4 * sizeof(float), st, y2, per, Tp, mask, mask_bs,
                           nullptr, nullptr, K, d, g2, b2, 1e-5f, xw, per, Tp, C, T, i == NL - 1 ? 1 : 0);
      else
        hipLaunchKernelGGL(dds_ln_gelu_kernel<1>, grid, dim3(256), lds, st, y2, per, Tp, mask, mask_bs, nullptr, nullptr, K, d, g2, b2,
                           1e-5f, xw, per, Tp, C, T, TT, i == NL - 1 ? 1 : 0);
      SVOC_HIP(hipGetLastError());
      stats_add_other();
      d *= K;
    }
    return k_copy2d(st, xw, per, Tp, y, y_bs, y_ld, B, C, T, nullptr, 0);
  }
};

// ---------------------------------------------------------------------------------------------
// Rational-quadratic spline (transforms.py:96-193) with optional linear tails (transforms.py:55-94).
// One lane per element.  Parameter p of element e is read at prm[e_off + p * p_stride] so that both the
// [n, bins] layout of the Python API and ConvFlow's [b, c*(3*bins-1)+p, t] layout are coalesced.
constexpr int MAXB = 32;

__device__ __forceinline__ float softplus_f(float v) { return v > 20.0f ? v : log1pf(expf(v)); }

struct SplineOut { float y, lad; };

struct SplineMin { float w, h, d; };     // min_bin_width / min_bin_height / min_derivative (transforms.py:7-9, 20-22)

template <class LoadW, class LoadH, class LoadD>
__device__ SplineOut rq_spline_elem(float x, int nb, bool inverse, float left, float right, float bottom, float top,
                                    LoadW lw, LoadH lh, LoadD ld_, SplineMin mn = SplineMin{1e-3f, 1e-3f, 1e-3f}) {
  const float min_w = mn.w, min_h = mn.h, min_d = mn.d;
  float cw[MAXB + 1], chh[MAXB + 1];
  {   // knots from softmax widths / heights
    float mx = -INFINITY;
    for (int k = 0; k < nb; ++k) mx = fmaxf(mx, lw(k));
    float sum = 0.f;
    for (int k = 0; k < nb; ++k) sum += expf(lw(k) - mx);
    float run = 0.f;
    cw[0] = left;
    for (int k = 0; k < nb; ++k) {
      const float w = min_w + (1.0f - min_w * nb) * (expf(lw(k) - mx) / sum);
      run += w;
      cw[k + 1] = (right - left) * run + left;
    }
    cw[nb] = right;
    mx = -INFINITY;
    for (int k = 0; k < nb; ++k) mx = fmaxf(mx, lh(k));
    sum = 0.f;
    for (int k = 0; k < nb; ++k) sum += expf(lh(k) - mx);
    run = 0.f;
    chh[0] = bottom;
    for (int k = 0; k < nb; ++k) {
      const float h = min_h + (1.0f - min_h * nb) * (expf(lh(k) - mx) / sum);
      run += h;
      chh[k + 1] = (top - bottom) * run + bottom;
    }
    chh[nb] = top;
  }
  // bin search: sum(x >= knot) - 1 with the last knot nudged by 1e-6 (transforms.py:47-52)
  int idx = -1;
  for (int k = 0; k <= nb; ++k) {
    float kn = inverse ? chh[k] : cw[k];
    if (k == nb) kn += 1e-6f;
    idx += (x >= kn) ? 1 : 0;
  }
  idx = idx < 0 ? 0 : (idx > nb - 1 ? nb - 1 : idx);
  float x_k = cw[0], x_k1 = cw[1], y_k = chh[0], y_k1 = chh[1];
  for (int k = 0; k < nb; ++k)
    if (k == idx) { x_k = cw[k]; x_k1 = cw[k + 1]; y_k = chh[k]; y_k1 = chh[k + 1]; }
  const float w_k = x_k1 - x_k, h_k = y_k1 - y_k;
  const float s_k = h_k / w_k;
  const float d0 = min_d + softplus_f(ld_(idx));
  const float d1 = min_d + softplus_f(ld_(idx + 1));
  SplineOut o;
  float th;
  if (inverse) {
    const float dy = x - y_k;
    const float tq = dy * (d0 + d1 - 2.0f * s_k);
    const float a = tq + h_k * (s_k - d0);
    const float bq = h_k * d0 - tq;
    const float c = -s_k * dy;
    const float disc = bq * bq - 4.0f * a * c;
    th = (2.0f * c) / (-bq - sqrtf(disc));
    o.y = th * w_k + x_k;
  } else {
    th = (x - x_k) / w_k;
  }
  const float tt = th * (1.0f - th);
  const float den = s_k + (d0 + d1 - 2.0f * s_k) * tt;
  const float num = s_k * s_k * (d1 * th * th + 2.0f * s_k * tt + d0 * (1.0f - th) * (1.0f - th));
  const float lad = logf(num) - 2.0f * logf(den);
  if (inverse) {
    o.lad = -lad;
  } else {
    o.y = y_k + h_k * (s_k * th * th + d0 * tt) / den;
    o.lad = lad;
  }
  return o;
}

// derivative loader for linear tails: index 0 and nb are the constant log(exp(1-1e-3)-1) (transforms.py:72-75)
__global__ void rq_spline_kernel(const float* __restrict__ x, const float* __restrict__ uw, const float* __restrict__ uh,
                                 const float* __restrict__ ud, long long n, int nb, int inverse, int linear_tails, float bound,
                                 float tail_const, SplineMin mn, float* __restrict__ y, float* __restrict__ lad) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const float xv = x[e];
  if (linear_tails && !(xv >= -bound && xv <= bound)) { y[e] = xv; lad[e] = 0.f; return; }
  const float* pw = uw + e * nb;
  const float* ph = uh + e * nb;
  const float* pd = ud + e * (linear_tails ? nb - 1 : nb + 1);
  const float lo = linear_tails ? -bound : 0.f, hi = linear_tails ? bound : 1.f;
  SplineOut o = rq_spline_elem(
      xv, nb, inverse != 0, lo, hi, lo, hi, [&](int k) { return pw[k]; }, [&](int k) { return ph[k]; },
      [&](int k) { return linear_tails ? ((k == 0 || k == nb) ? tail_const : pd[k - 1]) : pd[k]; }, mn);
  y[e] = o.y;
  lad[e] = o.lad;
}

// ConvFlow tail (modules.py:374-390): h [B][half*(3nb-1)][ld] -> spline on x1, cat, * mask, logdet
__global__ void convflow_spline_kernel(const float* __restrict__ x, const float* __restrict__ h, long long h_bs, int h_ld,
                                       const float* __restrict__ mask, int half, int T, int nb, float inv_sqrt_fc_den,
                                       int inverse, float bound, float tail_const, float* __restrict__ y,
                                       float* __restrict__ logdet) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y, b = blockIdx.z;
  float contrib = 0.f;
  if (t < T) {
    const int C = 2 * half;
    const float mk = mask[(long long)b * T + t];
    const long long i0 = ((long long)b * C + c) * T + t, i1 = ((long long)b * C + half + c) * T + t;
    y[i0] = x[i0] * mk;
    const float xv = x[i1];
    float out = xv, lad = 0.f;
    if (xv >= -bound && xv <= bound) {
      const int P = 3 * nb - 1;
      const float* hp = h + (long long)b * h_bs + (long long)c * P * h_ld + t;
      const float dn = inv_sqrt_fc_den;
      SplineOut o = rq_spline_elem(
          xv, nb, inverse != 0, -bound, bound, -bound, bound, [&](int k) { return hp[(long long)k * h_ld] / dn; },
          [&](int k) { return hp[(long long)(nb + k) * h_ld] / dn; },
          [&](int k) { return (k == 0 || k == nb) ? tail_const : hp[(long long)(2 * nb + k - 1) * h_ld]; });
      out = o.y;
      lad = o.lad;
    }
    y[i1] = out * mk;
    contrib = lad * mk;
  }
  if (logdet) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) contrib += __shfl_xor(contrib, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(logdet + b, contrib);
  }
}

// boundary derivative of the linear tails: softplus(c) + min_derivative == 1 (transforms.py:73)
static float tail_constant(double min_derivative = 1e-3) { return (float)std::log(std::exp(1.0 - min_derivative) - 1.0); }

struct ConvFlow {
  int Cin = 0, half = 0, F = 0, K = 0, NL = 0, nb = 10;
  float bound = 5.0f;
  PackedConv pre, proj;
  DDS dds;
  DevBuf ws;

  int create(int in_channels, int filter_channels, int k, int nl, int num_bins, float tail_bound, const TensorTable& tab,
             const std::string& prefix, hipStream_t st) {
    if (in_channels <= 0 || in_channels % 2 || num_bins <= 0 || num_bins > MAXB) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "ConvFlow: bad hyper-parameters");
    Cin = in_channels; half = Cin / 2; F = filter_channels; K = k; NL = nl; nb = num_bins; bound = tail_bound;
    PackSpec ps{}; ps.Cin = half; ps.Cout = F; ps.K = 1;
    SVOC_TRY(pack_conv_named(pre, ps, tab, prefix + "pre", st));
    SVOC_TRY(dds.create(F, k, nl, tab, prefix + "convs.", st));
    PackSpec qs{}; qs.Cin = F; qs.Cout = half * (3 * nb - 1); qs.K = 1;
    SVOC_TRY(pack_conv_named(proj, qs, tab, prefix + "proj", st));
    return SVOC_OK;
  }

  // g: NULL or [B][F][g_T] (g_T == 1 or T), added to pre(x0) inside DDSConv (modules.py:365-366 -> 97-98)
  int forward(hipStream_t st, const float* x, const float* mask, const float* g, int g_T, int reverse, float* y, float* logdet, int B, int T) {
    const int Tp = round_up(T, 4);
    const long long fper = (long long)F * Tp;
    const int P = half * (3 * nb - 1);
    const long long pper = (long long)P * Tp;
    SVOC_TRY(ws.ensure((size_t)((2 * fper + pper) * B) * sizeof(float)));
    float* h0 = ws.f();
    float* h1 = h0 + fper * B;
    float* hp = h1 + fper * B;
    {
      ConvArgs a = mk_args2();
      a.x = x; a.x_bs = (long long)Cin * T; a.x_ld = T; a.Lin = T; a.Ncols = T;
      a.out[0].y = h0; a.out[0].y_bs = fper; a.out[0].y_ld = Tp; a.out[0].nrows = F;
      SVOC_TRY(launch_conv(pre, a, B, st));
    }
    SVOC_TRY(dds.forward(st, h0, fper, Tp, mask, T, g, g_T, h1, fper, Tp, B, T));
    {
      ConvArgs a = mk_args2();
      a.x = h1; a.x_bs = fper; a.x_ld = Tp; a.Lin = T; a.Ncols = T;
      a.mask = mask; a.mask_bs = T;
      a.out[0].y = hp; a.out[0].y_bs = pper; a.out[0].y_ld = Tp; a.out[0].nrows = P; a.out[0].flags = F_OUTMASK;
      SVOC_TRY(launch_conv(proj, a, B, st));
    }
    if (logdet) SVOC_TRY(k_fill(st, logdet, (size_t)B, 0.f));
    hipLaunchKernelGGL(convflow_spline_kernel, dim3((T + 63) / 64, half, B), dim3(64), 0, st, x, hp, pper, Tp, mask, half, T, nb,
                       sqrtf((float)F), reverse, bound, tail_constant(), y, reverse ? nullptr : logdet);
    SVOC_HIP(hipGetLastError());
    stats_add_other();
    return SVOC_OK;
  }
};

}  // namespace svoc

using namespace svoc;
struct svoc_dds : svoc::HandleDevice { DDS m; };
struct svoc_convflow : svoc::HandleDevice { ConvFlow m; };

#define SVOC_GUARD_BEGIN try {
#define SVOC_GUARD_END } catch (const std::exception& e) { ::svoc::set_error("exception: %s", e.what()); return SVOC_ERR_NOMEM; }

extern "C" {

int svoc_dds_create(svoc_dds** out, int channels, int kernel_size, int n_layers, const svoc_tensor* tensors, int n_tensors,
                    const char* prefix) {
  if (!out || !tensors) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_dds_create: null argument");
  *out = nullptr;
  SVOC_GUARD_BEGIN
  std::unique_ptr<svoc_dds> h(new svoc_dds());
  TensorTable tab(tensors, n_tensors);
  SVOC_TRY(h->m.create(channels, kernel_size, n_layers, tab, prefix ? prefix : "", nullptr));
  *out = h.release();
  return SVOC_OK;
  SVOC_GUARD_END
}
int svoc_dds_forward(svoc_dds* h, void* stream, const float* x, const float* x_mask, const float* g, int g_T, float* y, int B, int T) {
  if (!h || !x || !x_mask || !y || B <= 0 || T <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_dds_forward: bad arguments");
  SVOC_GUARD_BEGIN
  const long long bs = (long long)h->m.C * T;
  return h->m.forward(as_stream(stream), x, bs, T, x_mask, T, g, g_T, y, bs, T, B, T);
  SVOC_GUARD_END
}
void svoc_dds_destroy(svoc_dds* h) { svoc::destroy_handle(h); }

int svoc_convflow_create(svoc_convflow** out, int in_channels, int filter_channels, int kernel_size, int n_layers, int num_bins,
                         float tail_bound, const svoc_tensor* tensors, int n_tensors, const char* prefix) {
  if (!out || !tensors) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_convflow_create: null argument");
  *out = nullptr;
  SVOC_GUARD_BEGIN
  std::unique_ptr<svoc_convflow> h(new svoc_convflow());
  TensorTable tab(tensors, n_tensors);
  SVOC_TRY(h->m.create(in_channels, filter_channels, kernel_size, n_layers, num_bins, tail_bound, tab, prefix ? prefix : "", nullptr));
  *out = h.release();
  return SVOC_OK;
  SVOC_GUARD_END
}
int svoc_convflow_forward(svoc_convflow* h, void* stream, const float* x, const float* x_mask, const float* g, int g_T, int reverse,
                          float* y, float* logdet, int B, int T) {
  if (!h || !x || !x_mask || !y || B <= 0 || T <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_convflow_forward: bad arguments");
  SVOC_GUARD_BEGIN
  return h->m.forward(as_stream(stream), x, x_mask, g, g_T, reverse, y, logdet, B, T);
  SVOC_GUARD_END
}
void svoc_convflow_destroy(svoc_convflow* h) { svoc::destroy_handle(h); }

/* modules.LayerNorm.forward (modules.py:28-32): layer norm over dim 1 of x [B, C, T] */
int svoc_layer_norm(void* stream, const float* x, const float* gamma, const float* beta, float eps, float* y, int B, int C, int T) {
  if (!x || !gamma || !beta || !y || B <= 0 || C <= 0 || T <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_layer_norm: bad arguments");
  int TT = 64;
  while (TT > 1 && ((size_t)C * TT + 512) * sizeof(float) > 64 * 1024) TT >>= 1;
  const size_t lds = ((size_t)C * TT + 512) * sizeof(float);
  if (lds > 64 * 1024) SVOC_FAIL(SVOC_ERR_UNSUPPORTED, "LayerNorm: %d channels do not fit the tile", C);
  const long long bs = (long long)C * T;
  if ((T & 3) == 0 && ln_v2_ok(x, bs, T, y, bs, T, C, T))
    hipLaunchKernelGGL(dds_ln_v2_kernel<2>, dim3((T + 63) / 64, B), dim3(256), (size_t)2 * 256 * 4 * sizeof(float), as_stream(stream), x, bs, T,
                       nullptr, 0, nullptr, nullptr, 1, 1, gamma, beta, eps, y, bs, T, C, T, 0);
  else
    hipLaunchKernelGGL(dds_ln_gelu_kernel<2>, dim3((T + TT - 1) / TT, B), dim3(256), lds, as_stream(stream), x, bs, T, nullptr, 0,
                       nullptr, nullptr, 1, 1, gamma, beta, eps, y, bs, T, C, T, TT, 0);
  SVOC_HIP(hipGetLastError());
  stats_add_other();
  return SVOC_OK;
}

int svoc_rq_spline(void* stream, const float* inputs, const float* unnorm_widths, const float* unnorm_heights,
                   const float* unnorm_derivs, int64_t n, int num_bins, int inverse, int linear_tails, float tail_bound,
                   float min_bin_width, float min_bin_height, float min_derivative, float* outputs, float* logabsdet) {
  if (min_bin_width * num_bins > 1.0f) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "Minimal bin width too large for the number of bins");
  if (min_bin_height * num_bins > 1.0f) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "Minimal bin height too large for the number of bins");
  if (!inputs || !unnorm_widths || !unnorm_heights || !unnorm_derivs || !outputs || !logabsdet || n < 0 || num_bins <= 0 ||
      num_bins > MAXB)
    SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_rq_spline: bad arguments (num_bins must be in 1..%d)", MAXB);
  if (n == 0) return SVOC_OK;
  hipLaunchKernelGGL(rq_spline_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, as_stream(stream), inputs, unnorm_widths,
                     unnorm_heights, unnorm_derivs, (long long)n, num_bins, inverse, linear_tails, tail_bound,
                     tail_constant(min_derivative), SplineMin{min_bin_width, min_bin_height, min_derivative}, outputs, logabsdet);
  SVOC_HIP(hipGetLastError());
  stats_add_other();
  return SVOC_OK;
}

}  // extern "C"
