// Winograd F(4,3) / F(4,4) convolutions (kernels: conv_wino4_kernels.h): weight transform + packing, and the dispatch to the
// instantiations (conv_wino4_r4.hip / _r2.hip / _r1.hip: one per row-tile layout in F(4,3) form; conv_wino44_r*.hip: k = 7 / 11 in F(4,4) form).
#include "conv_wino4.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace svoc {

// ------------------------------------------------------------------ weight transform + packing
// wp[m-tile][chunk][slot][k-group][lane][4] as pack_wino (conv_wino.hip) with slots = 6 G + ND: slot 6 g + p holds U_p of the
// three-tap group g, slot 6 G + t the plain tap 4 t + 3.  f44: slots = 7 G, slot 7 g + p holds U_p of the four-tap group g (taps
// beyond the kernel are zero).
__global__ void pack_wino4_kernel(const float* __restrict__ src, const float* __restrict__ scale, float* __restrict__ wp, int Cin,
                                  int Cout, int K, int nchunks, int slots, long long total, int f44) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int s = (int)(e & 3);
  const int lane = (int)((e >> 2) & 63);
  long long rest = e >> 8;
  const int kg = (int)(rest & 3); rest >>= 2;
  const int slot = (int)(rest % slots); rest /= slots;
  const int ch = (int)(rest % nchunks);
  const int mt = (int)(rest / nchunks);
  const int row = mt * 32 + (lane & 31);
  const int chan = ch * KC + 2 * (4 * kg + s) + (lane >> 5);
  float val = 0.f;
  if (row < Cout && chan < Cin) {
    const float* w = src + ((long long)row * Cin + chan) * K;
    const float sc = scale ? scale[row] : 1.0f;
    const int G = (K + 1) / 4;
    if (f44) {
      const int g = slot / 7, pp = slot % 7;
      const float w0 = w[4 * g] * sc, w1 = w[4 * g + 1] * sc, w2 = w[4 * g + 2] * sc, w3 = 4 * g + 3 < K ? w[4 * g + 3] * sc : 0.f;
      switch (pp) {
        case 0: val = w0; break;
        case 1: val = -((w0 + w2) + (w1 + w3)) * (2.0f / 9.0f); break;
        case 2: val = -((w0 + w2) - (w1 + w3)) * (2.0f / 9.0f); break;
        case 3: val = ((w0 + 4.f * w2) + (2.f * w1 + 8.f * w3)) * (1.0f / 90.0f); break;
        case 4: val = ((w0 + 4.f * w2) - (2.f * w1 + 8.f * w3)) * (1.0f / 90.0f); break;
        case 5: val = ((w0 + 0.25f * w2) + (0.5f * w1 + 0.125f * w3)) * (32.0f / 45.0f); break;
        default: val = ((w0 + 0.25f * w2) - (0.5f * w1 + 0.125f * w3)) * (32.0f / 45.0f); break;
      }
    } else if (slot < 6 * G) {
      const int g = slot / 6, pp = slot % 6;
      const float w0 = w[4 * g] * sc, w1 = w[4 * g + 1] * sc, w2 = w[4 * g + 2] * sc;
      switch (pp) {
        case 0: val = 0.25f * w0; break;
        case 1: val = -((w0 + w2) + w1) * (1.0f / 6.0f); break;
        case 2: val = -((w0 + w2) - w1) * (1.0f / 6.0f); break;
        case 3: val = ((w0 + 4.f * w2) + 2.f * w1) * (1.0f / 24.0f); break;
        case 4: val = ((w0 + 4.f * w2) - 2.f * w1) * (1.0f / 24.0f); break;
        default: val = w2; break;
      }
    } else {
      val = w[4 * (slot - 6 * G) + 3] * sc;               // taps 3, 7
    }
  }
  wp[e] = val;
}

int wino4_slots(int K, bool f44) { const int G = (K + 1) / 4; return f44 ? 7 * G : 6 * G + (G - 1); }

int pack_wino4_image(float* wp4, int Cin, int Cout, int K, bool f44, const float* w_or_v, const float* scale, hipStream_t st) {
  const int nchunks = (Cin + KC - 1) / KC, mtiles = (Cout + 31) / 32, slots = wino4_slots(K, f44);
  const long long total = (long long)mtiles * nchunks * slots * 4 * 256;
  hipLaunchKernelGGL(pack_wino4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w_or_v, scale, wp4, Cin, Cout, K, nchunks,
                     slots, total, f44 ? 1 : 0);
  SVOC_HIP(hipGetLastError());
  return SVOC_OK;
}

// ------------------------------------------------------------------ row tails of the dilated convolutions (round 6)
// A row of L outputs has D * ceil(L / 4D) windows; the decoder's lengths are 256 * 2^j * T, never a multiple of 4 D = 12 / 20, so the last q block is
// partial and - at T a power of two - its D windows are the ONLY occupants of one more column tile per row: 33 tiles instead of 32 at 16 x 512's
// first stage, 13 rounds of the 256 persistent workgroups instead of 12 for a dozen valid outputs per row.  Where dropping that tile saves a round
// (wino4_tail_plan) the launch stops at the last full tile and this kernel computes the dropped windows' outputs in direct form from the plain weights
// (PackedWino::wraw), one workgroup per work item (w4_tail_item below).  Window-major rows (out_perm) receive the value where the main launch would have put
// it.  Measured: profiles/r06_dilated_row_tails.txt (4 x 512 8.02 -> 7.86 ms; nothing at 16 x 512, which runs at the power cap).
struct W4TailMember { const float* x; const float* w; const float* bias; float* y; long long x_bs, y_bs; int x_ld, y_ld, Cin, Cout, K; };
// nv valid output columns per row: natural column n[v], position in the output row ypos[v]; pairs = (batch element, v), batch-major
struct W4Tail { W4TailMember m[3]; int L, D, nv, npairs; float slope; int n[32]; int ypos[32]; };
constexpr int W4T_P = 8;                                   // output columns per work item: their inputs share one pass over the weight rows
constexpr int W4T_R = 16;                                  // output channels per work item

// One work item: W4T_R output channels (block cb) x W4T_P (batch element, column) pairs (group pg) of one member, by a workgroup of NTH threads with
// W4T_P * Cin * K floats of LDS.  The pairs' inputs (leaky-relu applied, zero outside the row) are parked in LDS; every wave then takes W4T_R / waves
// weight rows: the lanes split the Cin x K products of each dot product (coalesced 16-byte loads along the row), a butterfly adds the 64 partial sums.
// Everything is issued in as few dependent round trips as the registers allow: the item is a chain of memory latencies, not of arithmetic.
template <int NTH>
__device__ __forceinline__ void w4_tail_item(const W4Tail& t, const int member, const int pg, const int cb, float* const xs) {
  constexpr int NWV = NTH / 64, RPW = W4T_R / NWV, CH = 12 / RPW;      // rows per wave; 256-float slices of them per round trip
  const W4TailMember& m = t.m[member];
  if (W4T_R * cb >= m.Cout) return;
  const int F = m.Cin * m.K, pad = (m.K - 1) / 2, K = m.K, D = t.D;
  const int p0 = W4T_P * pg, tid = threadIdx.x;
  const float* xbp[W4T_P];
  int ncol[W4T_P];
#pragma unroll
  for (int pp = 0; pp < W4T_P; ++pp) {
    const int pr = min(p0 + pp, t.npairs - 1), bz = pr / t.nv;
    xbp[pp] = m.x + (long long)bz * m.x_bs;
    ncol[pp] = p0 + pp < t.npairs ? t.n[pr - bz * t.nv] : -(1 << 28);      // (a column whose every input lies outside the row: zeros)
  }
  for (int e0 = tid; e0 < F; e0 += 4 * NTH) {
    float v[4][W4T_P];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = min(e0 + NTH * u, F - 1), ci = e / K, k = e - ci * K, off = (k - pad) * D;
      const long long ro = (long long)ci * m.x_ld;
#pragma unroll
      for (int pp = 0; pp < W4T_P; ++pp) {
        const int c = ncol[pp] + off;
        v[u][pp] = (c >= 0 && c < t.L) ? xbp[pp][ro + c] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + NTH * u;
      if (e < F) {
#pragma unroll
        for (int pp = 0; pp < W4T_P; ++pp) xs[pp * F + e] = v[u][pp] >= 0.f ? v[u][pp] : v[u][pp] * t.slope;
      }
    }
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int row0 = W4T_R * cb + RPW * wave;
  const float* wr = m.w + (long long)row0 * F;
  float acc[RPW][W4T_P];
#pragma unroll
  for (int u = 0; u < RPW; ++u)
#pragma unroll
    for (int pp = 0; pp < W4T_P; ++pp) acc[u][pp] = 0.f;
  for (int e0 = 4 * lane; e0 < F; e0 += 256 * CH) {       // (F is a multiple of 4: Cin % 32 == 0)
    float4 wv[CH][RPW];
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      const int e = min(e0 + 256 * q, F - 4);
#pragma unroll
      for (int u = 0; u < RPW; ++u) wv[q][u] = *reinterpret_cast<const float4*>(wr + (long long)u * F + e);
    }
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      const int e = e0 + 256 * q;
      if (e < F) {
#pragma unroll
        for (int pp = 0; pp < W4T_P; ++pp) {
          const float4 xv = *reinterpret_cast<const float4*>(xs + pp * F + e);
#pragma unroll
          for (int u = 0; u < RPW; ++u) {
            float a = acc[u][pp];
            a = __builtin_fmaf(wv[q][u].x, xv.x, a); a = __builtin_fmaf(wv[q][u].y, xv.y, a);
            a = __builtin_fmaf(wv[q][u].z, xv.z, a); a = __builtin_fmaf(wv[q][u].w, xv.w, a);
            acc[u][pp] = a;
          }
        }
      }
    }
  }
  float out = 0.f;
#pragma unroll
  for (int u = 0; u < RPW; ++u)
#pragma unroll
    for (int pp = 0; pp < W4T_P; ++pp) {
      float a = acc[u][pp];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
      out = lane == u * W4T_P + pp ? a : out;
    }
  const int pr = p0 + (lane & (W4T_P - 1)), row = row0 + lane / W4T_P;
  if (lane < RPW * W4T_P && pr < t.npairs) {
    const int bz = pr / t.nv, v = pr - bz * t.nv;
    m.y[(long long)bz * m.y_bs + (long long)row * m.y_ld + t.ypos[v]] = out + m.bias[row];
  }
}

__global__ void __launch_bounds__(512) conv_wino4_tail_kernel(const W4Tail t) {
  extern __shared__ __attribute__((aligned(16))) float xs[];     // [W4T_P][F]
  w4_tail_item<512>(t, (int)blockIdx.z, (int)blockIdx.x, (int)blockIdx.y, xs);
}

// plain weights [Cout][Cin * K] (weight norm applied) for the tail kernel
__global__ void wino4_raw_kernel(const float* __restrict__ src, const float* __restrict__ scale, float* __restrict__ dst, int inner, long long total) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < total) dst[e] = src[e] * (scale ? scale[e / inner] : 1.0f);
}
int pack_wino4_raw(float* dst, int Cin, int Cout, int K, const float* w_or_v, const float* scale, hipStream_t st) {
  const long long total = (long long)Cout * Cin * K;
  hipLaunchKernelGGL(wino4_raw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w_or_v, scale, dst, Cin * K, total);
  SVOC_HIP(hipGetLastError());
  return SVOC_OK;
}

// SVOC_W4_TAIL=0: every row keeps its partial last tile
bool wino4_tail_enabled() {
  static const bool on = !(getenv("SVOC_W4_TAIL") && atoi(getenv("SVOC_W4_TAIL")) == 0);
  return on;
}
// Tiles per row without the partial last one, or 0 when the row has none worth dropping: at most eight windows (a quarter of a column tile at most -
// the tail kernel is a dot product per output) behind at least one full tile.  w_first = first dropped window, nwin = their number.
int wino4_tail_plan(int L, int D, int NRT, int* w_first, int* nwin) {
  if (D == 1 || !wino4_tail_enabled()) return 0;
  const long long nw = (long long)D * ((L + 4 * D - 1) / (4 * D));
  const int nwt = 32 * (4 / NRT);
  const int rem = (int)(nw % nwt), full = (int)(nw / nwt);
  if (rem == 0 || rem > 8 || full < 1) return 0;
  *w_first = full * nwt; *nwin = rem;
  return full;
}
// the tails of a grouped launch's members (same L, D, window range) in one launch behind it
static int wino4_launch_tail(const WinoArgs* a, const float* const* wraw, const int* Cout, const int* K, int n, int D, int w_first, int nwin, int B, hipStream_t st) {
  W4Tail t{};
  int fmax = 0, cmax = 0;
  for (int i = 0; i < n; ++i) {
    W4TailMember& m = t.m[i];
    m.x = a[i].x; m.w = wraw[i]; m.bias = a[i].bias; m.y = a[i].y; m.x_bs = a[i].x_bs; m.y_bs = a[i].y_bs; m.x_ld = a[i].x_ld; m.y_ld = a[i].y_ld;
    m.Cin = a[i].Cin; m.Cout = Cout[i]; m.K = K[i];
    fmax = std::max(fmax, m.Cin * m.K); cmax = std::max(cmax, m.Cout);
  }
  t.L = a[0].L; t.D = D; t.slope = a[0].pre_slope;
  for (int wi = 0; wi < nwin; ++wi)
    for (int r = 0; r < 4; ++r) {
      const int w = w_first + wi, b = w / D, ph = w - b * D, col = 4 * D * b + ph + r * D;
      if (col < t.L) { t.n[t.nv] = col; t.ypos[t.nv] = a[0].out_perm ? 4 * w + r : col; ++t.nv; }
    }
  if (t.nv == 0) return SVOC_OK;
  t.npairs = t.nv * B;
  SVOC_TRY(ensure_max_dyn_lds((const void*)conv_wino4_tail_kernel));
  hipLaunchKernelGGL(conv_wino4_tail_kernel, dim3((unsigned)((t.npairs + W4T_P - 1) / W4T_P), (unsigned)(cmax / W4T_R), (unsigned)n), dim3(512),
                     (size_t)W4T_P * fmax * sizeof(float), st, t);
  SVOC_HIP(hipGetLastError());
  stats_add_other();
  return SVOC_OK;
}

// ------------------------------------------------------------------ launches
bool wino4_enabled() {
  static const bool on = !(getenv("SVOC_WINO_F4") && atoi(getenv("SVOC_WINO_F4")) == 0);      // SVOC_WINO_F4=0: the F(2,3) kernels
  return on;
}
// k = 7 / 11 run in F(4,4) form wherever the F(4,3) family applies (the F(4,3) form of those kernel sizes was an A/B arm until round 5)
bool wino44_enabled() { return wino4_enabled(); }
// SVOC_W4_C32=0: the C = 32 stage keeps the fused direct-form ResBlock kernel (resblock_fused.hip)
bool wino4_c32_enabled() {
  static const bool on = wino4_enabled() && !(getenv("SVOC_W4_C32") && atoi(getenv("SVOC_W4_C32")) == 0);
  return on;
}
// column tiles per row: a tile is 32 * (4 / NRT) consecutive windows; a row of L outputs has D * ceil(L / 4D) windows
int wino4_ntn(int L, int D, int NRT) {
  const long long nw = (long long)D * ((L + 4 * D - 1) / (4 * D));
  const int nwt = 32 * (4 / NRT);
  return (int)((nw + nwt - 1) / nwt);
}
// one persistent workgroup per CU (eight waves of up to 256 registers)
unsigned wino4_grid(long long total) { return (unsigned)std::min<long long>(total, (long long)device_cu_count()); }

// the instantiations, one translation unit per row-tile layout and form (conv_wino4_launch.h)
template <int NRT, bool F44> int wino4_launch_nrt(const WinoArgs& w, int K, int D, long long total, hipStream_t st);
template <int NRT, bool F44> int wino4_launch_group_nrt(const WinoGroup& g, int D, int in_perm, int out_perm, long long total, hipStream_t st);
#define SVOC_W4_EXTERN(NRT)                                                                                               \
  extern template int wino4_launch_nrt<NRT, false>(const WinoArgs&, int, int, long long, hipStream_t);                    \
  extern template int wino4_launch_nrt<NRT, true>(const WinoArgs&, int, int, long long, hipStream_t);                     \
  extern template int wino4_launch_group_nrt<NRT, true>(const WinoGroup&, int, int, int, long long, hipStream_t);
SVOC_W4_EXTERN(4) SVOC_W4_EXTERN(2) SVOC_W4_EXTERN(1)
#undef SVOC_W4_EXTERN

// f44: the weight image is in F(4,4) form (k = 7 / 11); k = 3: F(4,3)
int wino4_launch(const WinoArgs& w, int K, int D, int NRT, bool f44, long long total, hipStream_t st) {
  if (f44 != (K >= 7)) return 1;
  if (f44) return NRT == 4 ? wino4_launch_nrt<4, true>(w, K, D, total, st) : (NRT == 2 ? wino4_launch_nrt<2, true>(w, K, D, total, st) : wino4_launch_nrt<1, true>(w, K, D, total, st));
  return NRT == 4 ? wino4_launch_nrt<4, false>(w, K, D, total, st) : (NRT == 2 ? wino4_launch_nrt<2, false>(w, K, D, total, st) : wino4_launch_nrt<1, false>(w, K, D, total, st));
}
// in_perm (D = 1): 0, or the dilation of the convolutions that wrote the members' inputs window-major; out_perm (D > 1): nonzero =
// the members write window-major.  Members k = 11 / 7 in F(4,4) form, k = 3 in F(4,3)
// tail_nw > 0: the members' rows end at window tail_w0; the dropped windows' outputs follow in a launch of their own (conv_wino4_tail_kernel).  (Measured
// against the same work items taken by the grouped kernel's own workgroups once their tiles are done: 4 x 512 7.85 / 7.86 ms both ways, 16 x 512
// 25.17 / 25.17 - the kernels stay as they were.)
int wino4_launch_group(const WinoGroup& g, int D, int NRT, int in_perm, int out_perm, bool f44, long long total, hipStream_t st, const float* const* wraw,
                       const int* Cout, int tail_w0, int tail_nw, int B) {
  if (!f44) return 1;
  const int rc = NRT == 4 ? wino4_launch_group_nrt<4, true>(g, D, in_perm, out_perm, total, st)
                          : (NRT == 2 ? wino4_launch_group_nrt<2, true>(g, D, in_perm, out_perm, total, st) : wino4_launch_group_nrt<1, true>(g, D, in_perm, out_perm, total, st));
  if (rc == SVOC_OK && tail_nw > 0) return wino4_launch_tail(g.a, wraw, Cout, g.k, 3, D, tail_w0, tail_nw, B, st);
  return rc;
}

}  // namespace svoc
