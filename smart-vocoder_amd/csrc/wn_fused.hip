// One WN layer in one kernel (reference modules.py:160-175, commons.py:100-107):
//     x_in = in_layer(x)            Conv1d(H -> 2H, k, dilation d)
//     acts = tanh(x_in[:H] + g[:H]) * sigmoid(x_in[H:] + g[H:])
//     rs   = res_skip(acts)         Conv1d(H -> 2H, 1)   (last layer: H -> H)
//     x    = (x + rs[:H]) * mask ;  out += rs[H:]        (last layer: out = (out + rs) * mask)
// The 48 WN layers of the path work on only B*T columns (8192 at the headline config), so conv-by-conv execution
// is launch/occupancy bound.  Here one workgroup owns a time tile for ALL channels: wave w computes the (tanh,
// sigmoid) row pair w of the in_layer GEMM (K = H*k), the gate runs on the accumulators, the acts tile is parked in
// LDS (aliasing the x tile) and becomes the B operand of the 1x1 res_skip GEMM (K = H), whose two row tiles per wave
// (x-part w, skip-part w) feed the residual/skip epilogue.  acts never goes to HBM and one launch replaces two.
// x is ping-ponged between two buffers by the caller: a tile reads a halo that its neighbours overwrite.
#include "svoc_internal.h"
#include "wino_common.h"
#include "wn_f25.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace svoc {

struct WnArgs {
  const float* x; long long x_bs; int x_ld;            // layer input  [B][H][x_ld]
  float* xo; long long xo_bs; int xo_ld;               // layer output x (unused on the last layer)
  float* out; long long out_bs; int out_ld;            // skip accumulator
  const float* mask; long long mask_bs;
  const float* gadd; long long gadd_bs; int gadd_ld; int gadd_ts;   // conditioning slice of this layer: rows [0,2H)
  const float* wp1; const float* bias1; int ksg1; int dil; int pad;  // in_layer (paired packing)
  const float* wp2; const float* bias2; int ksg2;                     // res_skip (split packing, or plain on the last layer)
  const float* wpf;                                                   // in_layer in Winograd F(2,5) form (wn_layer_f25_kernel), or null
  int H; int ktaps; int nchunks; int npairs; int T;
  int xoff0; int xrow; int arow;
  int first; int last;
  long long* dbg;                                      // diagnostics: [workgroup][8] cycle stamps of wave 0 (tools/wn_timeline.py)
};

__device__ __forceinline__ float wn_pick4(const float4& v, int s) { return s == 0 ? v.x : (s == 1 ? v.y : (s == 2 ? v.z : v.w)); }

// acc[2][NR] += W[two 32-row tiles][K] * B with B fragments from an LDS tile ([channels][row_len]); NT2 = number of
// row tiles actually used (1 on the last layer's res_skip).
template <int NR>
__device__ __forceinline__ void wn_gemm(f32x16 (&acc)[2][NR], const float4* __restrict__ wp4, long long abase0, long long abase1,
                                        bool two, int ksg_total, const float* tile, int row_len, int col0, int ktaps, int dil,
                                        int nchunks, int hi, int ch_begin = 0) {
  // channel chunks [ch_begin, nchunks) of the K dimension (the K-split kernel gives each wave half of them);
  // ping-pong fragment registers; the next group's requests are issued after the first k-step's MFMAs
  float4 a0[2], a1[2];
  int ksg = ch_begin * ktaps * (KC / 8);
  a0[0] = wp4[abase0 + (long long)ksg * 64];
  a0[1] = wp4[abase1 + (long long)ksg * 64];
  for (int ch = ch_begin; ch < nchunks; ++ch) {
    const float* bp = tile + (ch * KC + hi) * row_len + col0;
    float b0[4][NR], b1[4][NR];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int nr = 0; nr < NR; ++nr) b0[s][nr] = bp[(2 * s) * row_len + nr * 32];
    const int ngroups = ktaps * (KC / 8);
    int g = 0;
    auto run_group = [&](float4(&ac)[2], float(&bc)[4][NR], float4(&an)[2], float(&bn)[4][NR], bool last_group) {
#pragma unroll
      for (int nr = 0; nr < NR; ++nr) {
        acc[0][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[0].x, bc[0][nr], acc[0][nr], 0, 0, 0);
        if (two) acc[1][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[1].x, bc[0][nr], acc[1][nr], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      ++ksg;
      const long long kn = (long long)(ksg < ksg_total ? ksg : 0) * 64;
      an[0] = wp4[abase0 + kn];
      an[1] = wp4[abase1 + kn];
      const float* bpn = (g == KC / 8 - 1) ? bp + dil - (KC - 8) * row_len : bp + 8 * row_len;
      if (!last_group) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int nr = 0; nr < NR; ++nr) bn[s][nr] = bpn[(2 * s) * row_len + nr * 32];
      }
      bp = bpn;
      g = (g + 1) & (KC / 8 - 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 1; s < 4; ++s) {
        const float av0 = wn_pick4(ac[0], s);
        const float av1 = wn_pick4(ac[1], s);
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
          acc[0][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bc[s][nr], acc[0][nr], 0, 0, 0);
          if (two) acc[1][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bc[s][nr], acc[1][nr], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    for (int gi = 0; gi < ngroups; gi += 2) {
      run_group(a0, b0, a1, b1, false);
      run_group(a1, b1, a0, b0, gi + 2 >= ngroups);
    }
  }
}

template <int NR>
__global__ void __launch_bounds__(512) wn_layer_fused_kernel(const WnArgs p) {
  constexpr int NA = NR * 32;
  constexpr int SU = 8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const XT = lds;      // x tile [H][xrow]; later aliased by the acts tile [H][arow]
  float* const AT = lds;

  const int tid = threadIdx.x;
  const int nthr = blockDim.x;
  const int lane = tid & 63;
  const int pi = __builtin_amdgcn_readfirstlane(tid >> 6);   // this wave's row pair
  const int l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.z;
  const int t0 = blockIdx.x * NA;
  const int H = p.H;

  // ---- stage the x tile: all H channels, columns [t0 + xoff0, +xrow), zero outside [0, T)
  {
    const int R4 = p.xrow >> 2;
    const int total = H * R4;
    const int xs_start = t0 + p.xoff0;
    const float* xb = p.x + (long long)b * p.x_bs;
    const bool vec = ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0) && (p.x_ld & 3) == 0 && (p.x_bs & 3) == 0;
    int wc = tid / R4, wg = tid - wc * R4;
    const int dc = nthr / R4, dg = nthr - dc * R4;
    for (int base = tid; base < total; base += nthr * SU) {
      float4 v[SU];
      const int wc_s = wc, wg_s = wg;
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int c = min(wc, H - 1);
        int t = xs_start + 4 * wg;
        const float* row = xb + (long long)c * p.x_ld;
        if (vec) {
          t = (t >= 0 && t < p.T) ? t : 0;
          v[u] = *reinterpret_cast<const float4*>(row + t);
        } else {
          v[u].x = (t >= 0 && t < p.T) ? row[t] : 0.f;
          v[u].y = (t + 1 >= 0 && t + 1 < p.T) ? row[t + 1] : 0.f;
          v[u].z = (t + 2 >= 0 && t + 2 < p.T) ? row[t + 2] : 0.f;
          v[u].w = (t + 3 >= 0 && t + 3 < p.T) ? row[t + 3] : 0.f;
        }
        wc += dc; wg += dg;
        if (wg >= R4) { wg -= R4; ++wc; }
      }
      int wc2 = wc_s, wg2 = wg_s;
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        if (base + u * nthr < total) {
          const int t = xs_start + 4 * wg2;
          float4 q = v[u];
          q.x = (t >= 0 && t < p.T) ? q.x : 0.f;
          q.y = (t + 1 >= 0 && t + 1 < p.T) ? q.y : 0.f;
          q.z = (t + 2 >= 0 && t + 2 < p.T) ? q.z : 0.f;
          q.w = (t + 3 >= 0 && t + 3 < p.T) ? q.w : 0.f;
          *reinterpret_cast<float4*>(XT + wc2 * p.xrow + 4 * wg2) = q;
        }
        wc2 += dc; wg2 += dg;
        if (wg2 >= R4) { wg2 -= R4; ++wc2; }
      }
    }
  }

  // ---- phase A: in_layer rows (tanh tile 2*pi, sigmoid tile 2*pi+1) on columns [t0, t0+NA)
  f32x16 acc[2][NR];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int nr = 0; nr < NR; ++nr)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[h][nr][i] = p.bias1[(2 * pi + h) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi];
  __syncthreads();
  {
    const float4* wp4 = reinterpret_cast<const float4*>(p.wp1);
    const long long ab0 = (long long)(2 * pi) * p.ksg1 * 64 + lane;
    wn_gemm<NR>(acc, wp4, ab0, ab0 + (long long)p.ksg1 * 64, true, p.ksg1, XT, p.xrow, l31 - p.pad - p.xoff0, p.ktaps, p.dil,
                p.nchunks, hi);
  }
  __syncthreads();     // everybody is done reading the x tile -> its LDS is reused for acts
  {
    const float* gb = p.gadd ? p.gadd + (long long)b * p.gadd_bs : nullptr;
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
      const int m = nr * 32 + l31;
      const int t = min(t0 + m, p.T - 1);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int chn = pi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        float vA = acc[0][nr][r], vB = acc[1][nr][r];
        if (gb) {
          vA += gb[(long long)chn * p.gadd_ld + (long long)t * p.gadd_ts];
          vB += gb[(long long)(H + chn) * p.gadd_ld + (long long)t * p.gadd_ts];
        }
        AT[chn * p.arow + m] = gate_tanh_sigmoid(vA, vB);
      }
    }
  }
  // ---- phase B: res_skip 1x1 on the acts tile; tile pi = x part, tile npairs + pi = skip part (last layer: tile pi only)
  const bool two = !p.last;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int nr = 0; nr < NR; ++nr)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        acc[h][nr][i] = (h == 0 || two) ? p.bias2[(h * p.npairs + pi) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi] : 0.f;
  __syncthreads();
  {
    const float4* wp4 = reinterpret_cast<const float4*>(p.wp2);
    const long long ab0 = (long long)pi * p.ksg2 * 64 + lane;
    const long long ab1 = two ? (long long)(p.npairs + pi) * p.ksg2 * 64 + lane : ab0;
    wn_gemm<NR>(acc, wp4, ab0, ab1, two, p.ksg2, AT, p.arow, l31, 1, 1, p.nchunks, hi);
  }
  // ---- epilogue (modules.py:168-175)
  const float* mb = p.mask + (long long)b * p.mask_bs;
#pragma unroll
  for (int nr = 0; nr < NR; ++nr) {
    const int t = t0 + nr * 32 + l31;
    if (t >= p.T) continue;
    const float mk = mb[t];
    const int row0 = pi * 32 + 4 * hi;
    float* ob = p.out + (long long)b * p.out_bs + (long long)row0 * p.out_ld + t;
    if (two) {
      const float* xr = p.x + (long long)b * p.x_bs + (long long)row0 * p.x_ld + t;
      float* xw = p.xo + (long long)b * p.xo_bs + (long long)row0 * p.xo_ld + t;
      float rv[16], ov[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) rv[r] = xr[(long long)((r & 3) + 8 * (r >> 2)) * p.x_ld];
      if (!p.first) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ov[r] = ob[(long long)((r & 3) + 8 * (r >> 2)) * p.out_ld];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr = (r & 3) + 8 * (r >> 2);
        xw[(long long)lr * p.xo_ld] = (rv[r] + acc[0][nr][r]) * mk;
        ob[(long long)lr * p.out_ld] = p.first ? acc[1][nr][r] : ov[r] + acc[1][nr][r];
      }
    } else {
      float ov[16];
      if (!p.first) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ov[r] = ob[(long long)((r & 3) + 8 * (r >> 2)) * p.out_ld];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr = (r & 3) + 8 * (r >> 2);
        const float v = p.first ? acc[0][nr][r] : ov[r] + acc[0][nr][r];
        ob[(long long)lr * p.out_ld] = v * mk;
      }
    }
  }
}

// K-split variant for short inputs (one 32-column workgroup per CU at the headline config).  With one wave per row
// pair a workgroup has H/32 = 6 waves on 4 SIMDs (2,2,1,1: the matrix pipes are 75 % used at best).  Here 2*npairs
// waves run: wave (pi, kh) multiplies row pair pi over half kh of the input channels; partial accumulators are
// exchanged through LDS so that each wave finishes HALF of the pair's outputs (gate: 8 of the 16 accumulator
// registers; res_skip: kh=0 the residual tile, kh=1 the skip tile).  12 waves = 3 per SIMD, every phase balanced.
// CT: compile-time streams (H = 192, k = 5, dilation 1: x rows of 40 floats, acts rows of 33); LAST: the stack's last layer
template <bool CT, bool LAST>
__global__ void __launch_bounds__(768) wn_layer_fused_ks_kernel(const WnArgs p) {
  constexpr int NR = 1;
  constexpr int NA = 32;
  constexpr int SU = 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const XT = lds;      // x tile [H][xrow]; later aliased by the acts tile [H][arow]
  float* const AT = lds;
  const int H = p.H;
  float* const RED = lds + H * max(p.xrow, p.arow);   // exchange area: [2*npairs waves][16][64]

  const int tid = threadIdx.x;
  const int nthr = blockDim.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = wave >= p.npairs ? 1 : 0;                 // K half
  const int pi = wave - kh * p.npairs;                     // row pair
  const int l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.z;
  const int t0 = blockIdx.x * NA;
  const int chh = p.nchunks >> 1;                          // chunks per half (nchunks is even)
  const int ch_lo = kh * chh, ch_hi = ch_lo + chh;
  float* const red_mine = RED + (wave * 16) * 64 + lane;
  float* const red_peer = RED + ((kh ? pi : pi + p.npairs) * 16) * 64 + lane;

  long long ts[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  auto stamp = [&](int i) { if (p.dbg) ts[i] = (long long)__builtin_readcyclecounter(); };
  stamp(0);
  // ---- stage the x tile: all H channels, columns [t0 + xoff0, +xrow), zero outside [0, T)
  {
    const int R4 = p.xrow >> 2;
    const int total = H * R4;
    const int xs_start = t0 + p.xoff0;
    const float* xb = p.x + (long long)b * p.x_bs;
    const bool vec = ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0) && (p.x_ld & 3) == 0 && (p.x_bs & 3) == 0;
    int wc = tid / R4, wg = tid - wc * R4;
    const int dc = nthr / R4, dg = nthr - dc * R4;
    for (int base = tid; base < total; base += nthr * SU) {
      float4 v[SU];
      const int wc_s = wc, wg_s = wg;
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int c = min(wc, H - 1);
        int t = xs_start + 4 * wg;
        const float* row = xb + (long long)c * p.x_ld;
        if (vec) {
          t = (t >= 0 && t < p.T) ? t : 0;
          v[u] = *reinterpret_cast<const float4*>(row + t);
        } else {
          v[u].x = (t >= 0 && t < p.T) ? row[t] : 0.f;
          v[u].y = (t + 1 >= 0 && t + 1 < p.T) ? row[t + 1] : 0.f;
          v[u].z = (t + 2 >= 0 && t + 2 < p.T) ? row[t + 2] : 0.f;
          v[u].w = (t + 3 >= 0 && t + 3 < p.T) ? row[t + 3] : 0.f;
        }
        wc += dc; wg += dg;
        if (wg >= R4) { wg -= R4; ++wc; }
      }
      int wc2 = wc_s, wg2 = wg_s;
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        if (base + u * nthr < total) {
          const int t = xs_start + 4 * wg2;
          float4 q = v[u];
          q.x = (t >= 0 && t < p.T) ? q.x : 0.f;
          q.y = (t + 1 >= 0 && t + 1 < p.T) ? q.y : 0.f;
          q.z = (t + 2 >= 0 && t + 2 < p.T) ? q.z : 0.f;
          q.w = (t + 3 >= 0 && t + 3 < p.T) ? q.w : 0.f;
          *reinterpret_cast<float4*>(XT + wc2 * p.xrow + 4 * wg2) = q;
        }
        wc2 += dc; wg2 += dg;
        if (wg2 >= R4) { wg2 -= R4; ++wc2; }
      }
    }
  }

  // ---- phase A: in_layer rows (tanh tile 2*pi, sigmoid tile 2*pi+1), this wave's half of the input channels
  f32x16 acc[2][NR];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 16; ++i)
      acc[h][0][i] = kh ? 0.f : p.bias1[(2 * pi + h) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi];
  __syncthreads();
  stamp(1);
  {
    const float4* wp4 = reinterpret_cast<const float4*>(p.wp1);
    const long long ab0 = (long long)(2 * pi) * p.ksg1 * 64 + lane;
    if constexpr (CT) {
      const unsigned baddr = (unsigned)(size_t)XT + (unsigned)((ch_lo * KC + hi) * 40 + (l31 - p.pad - p.xoff0)) * 4u;
      const int w0 = __builtin_amdgcn_readfirstlane(((2 * pi) * p.ksg1 + ch_lo * 5 * 4) * 1024);
      wn_gemm_ct<40, 5, 1, true>(acc, p.wp1, w0, w0 + p.ksg1 * 1024, baddr, (unsigned)lane * 16u);
    } else
    wn_gemm<NR>(acc, wp4, ab0, ab0 + (long long)p.ksg1 * 64, true, p.ksg1, XT, p.xrow, l31 - p.pad - p.xoff0, p.ktaps, p.dil,
                ch_hi, hi, ch_lo);
  }
  stamp(2);
  // exchange: this wave finishes registers [8*kh, 8*kh+8) of both tiles and hands the other 8 to its peer
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int q = 0; q < 8; ++q) red_mine[(h * 8 + q) * 64] = kh ? acc[h][0][q] : acc[h][0][8 + q];
  stamp(7);
  __syncthreads();     // partials published; everybody is done reading the x tile -> its LDS is reused for acts
  stamp(8);
  {
    const float* gb = p.gadd ? p.gadd + (long long)b * p.gadd_bs : nullptr;
    const int m = l31;
    const int t = min(t0 + m, p.T - 1);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int r = kh * 8 + q;
      const int chn = pi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      float vA = (kh ? acc[0][0][8 + q] : acc[0][0][q]) + red_peer[q * 64];
      float vB = (kh ? acc[1][0][8 + q] : acc[1][0][q]) + red_peer[(8 + q) * 64];
      if (gb) {
        vA += gb[(long long)chn * p.gadd_ld + (long long)t * p.gadd_ts];
        vB += gb[(long long)(H + chn) * p.gadd_ld + (long long)t * p.gadd_ts];
      }
      AT[chn * p.arow + m] = gate_tanh_sigmoid(vA, vB);
    }
  }
  stamp(9);
  // ---- phase B: res_skip 1x1 on the acts tile; tile pi = x part, tile npairs + pi = skip part (last layer: tile pi only)
  const bool two = CT ? !LAST : !p.last;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 16; ++i)
      acc[h][0][i] = (!kh && (h == 0 || two)) ? p.bias2[(h * p.npairs + pi) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi] : 0.f;
  __syncthreads();     // acts tile complete, exchange area free again
  stamp(3);
  {
    const float4* wp4 = reinterpret_cast<const float4*>(p.wp2);
    const long long ab0 = (long long)pi * p.ksg2 * 64 + lane;
    const long long ab1 = two ? (long long)(p.npairs + pi) * p.ksg2 * 64 + lane : ab0;
    if constexpr (CT) {
      const unsigned baddr = (unsigned)(size_t)AT + (unsigned)((ch_lo * KC + hi) * 33 + l31) * 4u;
      const int w0 = __builtin_amdgcn_readfirstlane((pi * p.ksg2 + ch_lo * 4) * 1024);
      const int w1 = __builtin_amdgcn_readfirstlane(((p.npairs + pi) * p.ksg2 + ch_lo * 4) * 1024);
      wn_gemm_ct<33, 1, 1, !LAST>(acc, p.wp2, w0, LAST ? w0 : w1, baddr, (unsigned)lane * 16u);
    } else
    wn_gemm<NR>(acc, wp4, ab0, ab1, two, p.ksg2, AT, p.arow, l31, 1, 1, ch_hi, hi, ch_lo);
  }
  stamp(4);
  // exchange: kh=0 finishes tile 0 (residual part; on the last layer the only tile), kh=1 finishes tile 1 (skip part)
  if (two) {
#pragma unroll
    for (int q = 0; q < 16; ++q) red_mine[q * 64] = kh ? acc[0][0][q] : acc[1][0][q];
  } else if (kh) {
#pragma unroll
    for (int q = 0; q < 16; ++q) red_mine[q * 64] = acc[0][0][q];
  }
  __syncthreads();
  float fin[16];
  if (two || !kh) {
#pragma unroll
    for (int q = 0; q < 16; ++q) fin[q] = (kh ? acc[1][0][q] : acc[0][0][q]) + red_peer[q * 64];
  }
  stamp(5);
  auto dump = [&]() {
    if (p.dbg && tid == 0) {
      long long* d = p.dbg + 8 * (long long)(blockIdx.x + gridDim.x * blockIdx.z);
      for (int i = 0; i < 6; ++i) d[i] = ts[i];
      d[6] = (long long)__builtin_readcyclecounter(); d[7] = ((ts[7] - ts[2]) << 42) | ((ts[8] - ts[7]) << 21) | (ts[9] - ts[8]);
    }
  };
  // ---- epilogue (modules.py:168-175)
  const int t = t0 + l31;
  if (t >= p.T) { dump(); return; }
  const float* mb = p.mask + (long long)b * p.mask_bs;
  const float mk = mb[t];
  const int row0 = pi * 32 + 4 * hi;
  float* ob = p.out + (long long)b * p.out_bs + (long long)row0 * p.out_ld + t;
  if (two) {
    if (!kh) {         // x = (x + rs[:H]) * mask
      const float* xr = p.x + (long long)b * p.x_bs + (long long)row0 * p.x_ld + t;
      float* xw = p.xo + (long long)b * p.xo_bs + (long long)row0 * p.xo_ld + t;
      float rv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) rv[r] = xr[(long long)((r & 3) + 8 * (r >> 2)) * p.x_ld];
#pragma unroll
      for (int r = 0; r < 16; ++r) xw[(long long)((r & 3) + 8 * (r >> 2)) * p.xo_ld] = (rv[r] + fin[r]) * mk;
    } else {           // out += rs[H:]
      float ov[16];
      if (!p.first) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ov[r] = ob[(long long)((r & 3) + 8 * (r >> 2)) * p.out_ld];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) ob[(long long)((r & 3) + 8 * (r >> 2)) * p.out_ld] = p.first ? fin[r] : ov[r] + fin[r];
    }
  } else if (!kh) {    // last layer: out = (out + rs) * mask
    float ov[16];
    if (!p.first) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ov[r] = ob[(long long)((r & 3) + 8 * (r >> 2)) * p.out_ld];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = p.first ? fin[r] : ov[r] + fin[r];
      ob[(long long)((r & 3) + 8 * (r >> 2)) * p.out_ld] = v * mk;
    }
  }
  dump();
}


// ---------------------------------------------------------------------------------------------------------------------------
// Round 4: the in_layer (k = 5: 960 of the 1152 k-steps of a layer) in Winograd F(2,5) form - two outputs from six products
// instead of ten, 3/5 of the direct form's multiply-adds - with the matrix pipe's 16-column tile (v_mfma_f32_16x16x4_f32: same
// FLOP rate as 32x32x2, half the columns), so that a 32-column workgroup still fills its MFMAs: 16 windows of two outputs.
// Interpolation points 0, +-1, +-2, infinity: the input transform is conv_wino4.hip's (it depends on the points only),
//     V0 = 4 d0 - 5 d2 + d4     V1 = -4 d1 - 4 d2 + d3 + d4     V2 = 4 d1 - 4 d2 - d3 + d4     d_j = x[2q - 2 + j]
//     V3 = -2 d1 - d2 + 2 d3 + d4     V4 = 2 d1 - d2 - 2 d3 + d4     V5 = 4 d1 - 5 d3 + d5
//     U0 = w0/4   U1 = -(w0+w1+w2+w3+w4)/6   U2 = -(w0-w1+w2-w3+w4)/6   U3 = (w0+2w1+4w2+8w3+16w4)/24
//     U4 = (w0-2w1+4w2-8w3+16w4)/24   U5 = w4          M_p = sum_c U_p[c] V_p[c]  (the GEMMs)
//     y[2q] = M0 + M1 + M2 + M3 + M4          y[2q+1] = M1 - M2 + 2 M3 - 2 M4 + M5
// fp32 error 3.6 x the direct form's (tools/wino_numerics.py) - the F(4,3) decoder kernels' constant.  Same workgroup as the
// K-split kernel: 12 waves, wave (pi, kh) = row pair pi (tanh rows 32 pi .., sigmoid rows H + 32 pi ..: four 16-row tiles) over
// half kh of the input channels, 4 x 6 accumulator tiles of 4 registers; per (k-step, product) ONE ds_read_b32 (the B fragment is
// shared by the four row tiles) and ONE 16-byte buffer load (the four tiles' A values): a VALU-free stream of 576 MFMAs of 32
// cycles (the direct form: 480 of 64).  The six planes [p][H][16 windows] are written by all waves ahead of the stream (4 windows
// per thread); the output transform runs on the partial sums before the K-half exchange.  Phases B (res_skip 1x1) and the
// epilogue are the K-split kernel's.  H = 192, k = 5, dilation 1 only (every WN of the path).

template <bool LAST>
__global__ void __launch_bounds__(768) wn_layer_f25_kernel(const WnArgs p) {
  constexpr int H = WNF_H, NPAIRS = WNF_NP;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const XT = lds;                                    // x tile [H][40]; later aliased by the acts tile [H][33]
  float* const AT = lds;
  float* const PLN = lds + H * WNF_XROW;                    // V_p [6][H][16]
  float* const RED = PLN + 6 * WNF_PLANE;                   // exchange area [12 waves][16][64]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = wave >= NPAIRS ? 1 : 0;                    // K half
  const int pi = wave - kh * NPAIRS;                        // row pair
  const int l31 = lane & 31, hi = lane >> 5;                // 32x32x2 fragment coordinates (phase B, epilogue)
  const int col = lane & 15, k4 = lane >> 4;                // 16x16x4 fragment coordinates (phase A)
  const int b = blockIdx.z;
  const int t0 = blockIdx.x * 32;
  float* const red_mine = RED + (wave * 16) * 64 + lane;
  float* const red_peer = RED + ((kh ? pi : pi + NPAIRS) * 16) * 64 + lane;

  long long ts[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  auto stamp = [&](int i) { if (p.dbg) ts[i] = (long long)__builtin_readcyclecounter(); };
  stamp(0);
  // phase A's weight stream: descriptor, ring of four 16-byte sets, three requests ahead.  The first three go out here, ahead of the x tile
  // (round 5: they used to wait behind the planes' barrier, and the stream then started with an L2 round trip)
  constexpr int NST = WNF_KS * 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpf), 0, 0x7fffffff, 0x00020000);
  const int w0 = __builtin_amdgcn_readfirstlane((pi * 2 + kh) * NST * 1024);
  const unsigned wlane = (unsigned)lane * 16u;
  float4 a[4];
  auto wload = [&](float4& d, int soff) {
    const wn_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)wlane, soff, 0);
    d = *reinterpret_cast<const float4*>(&t);
  };
  wload(a[0], w0); wload(a[1], w0 + 1024); wload(a[2], w0 + 2048);
  // ---- stage the x tile: all H channels, columns [t0 - 4, t0 + 36), zero outside [0, T)
  {
    constexpr int R4 = WNF_XROW / 4, total = H * R4, SU = 3;       // 1920 groups of four floats: 2.5 per thread
    const int xs_start = t0 - 4;
    const float* xb = p.x + (long long)b * p.x_bs;
    const bool vec = ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0) && (p.x_ld & 3) == 0 && (p.x_bs & 3) == 0;
    float4 v[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int it = min(tid + 768 * u, total - 1);
      const int c = it / R4, g4 = it - c * R4;
      int t = xs_start + 4 * g4;
      const float* row = xb + (long long)c * p.x_ld;
      if (vec) {
        t = (t >= 0 && t < p.T) ? t : 0;
        v[u] = *reinterpret_cast<const float4*>(row + t);
      } else {
        v[u].x = (t >= 0 && t < p.T) ? row[t] : 0.f;
        v[u].y = (t + 1 >= 0 && t + 1 < p.T) ? row[t + 1] : 0.f;
        v[u].z = (t + 2 >= 0 && t + 2 < p.T) ? row[t + 2] : 0.f;
        v[u].w = (t + 3 >= 0 && t + 3 < p.T) ? row[t + 3] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int it = tid + 768 * u;
      if (it < total) {
        const int c = it / R4, g4 = it - c * R4;
        const int t = xs_start + 4 * g4;
        float4 q = v[u];
        q.x = (t >= 0 && t < p.T) ? q.x : 0.f;
        q.y = (t + 1 >= 0 && t + 1 < p.T) ? q.y : 0.f;
        q.z = (t + 2 >= 0 && t + 2 < p.T) ? q.z : 0.f;
        q.w = (t + 3 >= 0 && t + 3 < p.T) ? q.w : 0.f;
        *reinterpret_cast<float4*>(XT + c * WNF_XROW + 4 * g4) = q;
      }
    }
  }
  __syncthreads();
  stamp(1);
  // ---- input transform: window q of channel c reads x[t0 + 2q - 2 .. + 3] = tile columns 2q + 2 .. 2q + 7 (8-byte aligned)
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = (tid >> 4) + 48 * u, q = tid & 15;
    const float* r = XT + c * WNF_XROW + 2 * q + 2;
    const float2 f0 = *reinterpret_cast<const float2*>(r), f1 = *reinterpret_cast<const float2*>(r + 2), f2 = *reinterpret_cast<const float2*>(r + 4);
    const float d0 = f0.x, d1 = f0.y, d2 = f1.x, d3 = f1.y, d4 = f2.x, d5 = f2.y;
    const float a_ = __builtin_fmaf(-4.f, d2, d4), b_ = __builtin_fmaf(-4.f, d1, d3);      // d4 - 4 d2, d3 - 4 d1
    const float c_ = d4 - d2, e_ = 2.f * (d3 - d1);
    float* o = PLN + c * WNF_NQ + q;
    o[0] = __builtin_fmaf(4.f, d0, __builtin_fmaf(-5.f, d2, d4));
    o[WNF_PLANE] = a_ + b_;
    o[2 * WNF_PLANE] = a_ - b_;
    o[3 * WNF_PLANE] = c_ + e_;
    o[4 * WNF_PLANE] = c_ - e_;
    o[5 * WNF_PLANE] = __builtin_fmaf(4.f, d1, __builtin_fmaf(-5.f, d3, d5));
  }
  __syncthreads();
  stamp(2);
  // ---- phase A: 24 k-steps x 6 products x 4 row tiles of v_mfma_f32_16x16x4_f32
  wn_f32x4 M[4][6];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int q = 0; q < 6; ++q) M[rt][q] = (wn_f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const unsigned baddr0 = (unsigned)(size_t)PLN + (unsigned)(((kh * 96 + k4) * WNF_NQ + col) * 4);
    const unsigned baddr1 = baddr0 + 3u * WNF_PLANE * 4u;
    float fb[2];
    auto rdb = [&](auto ic) {
      constexpr int I = decltype(ic)::value;
      if constexpr (I < NST) {
        constexpr int KS_ = I / 6, P_ = I % 6;
        constexpr int O = ((P_ % 3) * WNF_PLANE + KS_ * 4 * WNF_NQ) * 4;
        fb[I & 1] = wino_lds_rd<O>(P_ < 3 ? baddr0 : baddr1);
      }
    };
    auto rqw = [&](auto ic) {
      constexpr int I = decltype(ic)::value;
      if constexpr (I < NST) wload(a[I & 3], w0 + I * 1024);
    };
    auto step = [&](auto ic) {
      constexpr int I = decltype(ic)::value;
      constexpr int P_ = I % 6;
      rqw(std::integral_constant<int, I + 3>{});
      {
        float& bq = fb[I & 1];
        if constexpr (I + 1 < NST) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(bq));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq));
      }
      const float4 av = a[I & 3];
      const float bv = fb[I & 1];
      M[0][P_] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv, M[0][P_], 0, 0, 0);
      M[1][P_] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv, M[1][P_], 0, 0, 0);
      M[2][P_] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv, M[2][P_], 0, 0, 0);
      M[3][P_] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv, M[3][P_], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      rdb(std::integral_constant<int, I + 2>{});
      __builtin_amdgcn_sched_barrier(0);
    };
    rdb(std::integral_constant<int, 0>{}); rdb(std::integral_constant<int, 1>{});
    wino_static_for<0, NST>(step);
  }
  stamp(3);
  // ---- output transform of the partial sums; this wave finishes 16-row tile kh of both halves and hands the other to its peer
  float own[2][4][2];                                       // [tanh | sigmoid][register i][output 0 | 1]
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    const int h = rt >> 1;
    const bool mine = (rt & 1) == kh;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float s12 = M[rt][1][i] + M[rt][2][i], d12 = M[rt][1][i] - M[rt][2][i];
      const float s34 = M[rt][3][i] + M[rt][4][i], d34 = M[rt][3][i] - M[rt][4][i];
      const float y0 = M[rt][0][i] + (s12 + s34);
      const float y1 = __builtin_fmaf(2.f, d34, d12) + M[rt][5][i];
      if (mine) { own[h][i][0] = y0; own[h][i][1] = y1; }
      else { red_mine[(h * 8 + 2 * i) * 64] = y0; red_mine[(h * 8 + 2 * i + 1) * 64] = y1; }
    }
  }
  // The epilogue's read operands - kh = 0: the residual rows of x (on the last layer: the rows of `out`), kh = 1: the rows of `out` - are requested
  // here, their latency under the exchange barriers, the gate and phase B (round 5: the epilogue used to start with these sixteen loads).
  const int te = t0 + l31, tec = min(te, p.T - 1);
  const int row0 = pi * 32 + 4 * hi;
  float pre[16];
  {
    const bool want_x = !LAST && !kh, want_o = (LAST ? !kh : kh) && !p.first;
    const float* src = want_x ? p.x + (long long)b * p.x_bs + (long long)row0 * p.x_ld + tec
                              : p.out + (long long)b * p.out_bs + (long long)row0 * p.out_ld + tec;
    const long long sld = want_x ? p.x_ld : p.out_ld;
    if (want_x || want_o) {
#pragma unroll
      for (int r = 0; r < 16; ++r) pre[r] = src[(long long)((r & 3) + 8 * (r >> 2)) * sld];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) pre[r] = 0.f;
    }
  }
  stamp(7);
  __syncthreads();     // partials published
  stamp(8);
  {
    const float* gb = p.gadd ? p.gadd + (long long)b * p.gadd_bs : nullptr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = 16 * kh + 4 * k4 + i;                  // row inside the pair's 32-row tiles
      const int chn = pi * 32 + rr;
      const float bA = p.bias1[(2 * pi) * 32 + rr], bB = p.bias1[(2 * pi + 1) * 32 + rr];
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        const int m = 2 * col + o;
        float vA = (own[0][i][o] + red_peer[(2 * i + o) * 64]) + bA;
        float vB = (own[1][i][o] + red_peer[(8 + 2 * i + o) * 64]) + bB;
        if (gb) {
          const int t = min(t0 + m, p.T - 1);
          vA += gb[(long long)chn * p.gadd_ld + (long long)t * p.gadd_ts];
          vB += gb[(long long)(H + chn) * p.gadd_ld + (long long)t * p.gadd_ts];
        }
        AT[chn * WNF_AROW + m] = gate_tanh_sigmoid(vA, vB);
      }
    }
  }
  stamp(9);
  // ---- phase B: res_skip 1x1 on the acts tile (the K-split kernel's): tile pi = x part, tile npairs + pi = skip part
  f32x16 acc[2][1];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 16; ++i)
      acc[h][0][i] = (!kh && (h == 0 || !LAST)) ? p.bias2[(h * NPAIRS + pi) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi] : 0.f;
  __syncthreads();     // acts tile complete, exchange area free again
  stamp(4);
  {
    const unsigned baddr = (unsigned)(size_t)AT + (unsigned)((kh * 96 + hi) * WNF_AROW + l31) * 4u;
    const int w0 = __builtin_amdgcn_readfirstlane((pi * p.ksg2 + kh * 12) * 1024);
    const int w1 = __builtin_amdgcn_readfirstlane(((NPAIRS + pi) * p.ksg2 + kh * 12) * 1024);
    wn_gemm_ct<WNF_AROW, 1, 1, !LAST>(acc, p.wp2, w0, LAST ? w0 : w1, baddr, (unsigned)lane * 16u);
  }
  stamp(5);
  // exchange: kh=0 finishes tile 0 (residual part; on the last layer the only tile), kh=1 finishes tile 1 (skip part)
  if (!LAST) {
#pragma unroll
    for (int q = 0; q < 16; ++q) red_mine[q * 64] = kh ? acc[0][0][q] : acc[1][0][q];
  } else if (kh) {
#pragma unroll
    for (int q = 0; q < 16; ++q) red_mine[q * 64] = acc[0][0][q];
  }
  __syncthreads();
  float fin[16];
  if (!LAST || !kh) {
#pragma unroll
    for (int q = 0; q < 16; ++q) fin[q] = (kh ? acc[1][0][q] : acc[0][0][q]) + red_peer[q * 64];
  }
  auto dump = [&]() {
    if (p.dbg && tid == 0) {
      long long* d = p.dbg + 8 * (long long)(blockIdx.x + gridDim.x * blockIdx.z);
      // [0..6]: start | x tile staged | planes written | phase A stream done | acts complete (phase B starts) | phase B done | end;
      // [7]: output transform + partial-sum writes, barrier, gate (21 bits each); a negative marker tells the tool which kernel stamped
      d[0] = ts[0]; d[1] = ts[1]; d[2] = ts[2]; d[3] = ts[3]; d[4] = ts[4]; d[5] = ts[5];
      d[6] = -(long long)__builtin_readcyclecounter(); d[7] = ((ts[7] - ts[3]) << 42) | ((ts[8] - ts[7]) << 21) | (ts[9] - ts[8]);
    }
  };
  // ---- epilogue (modules.py:168-175)
  if (te >= p.T) { dump(); return; }
  const float* mb = p.mask + (long long)b * p.mask_bs;
  const float mk = mb[te];
  float* ob = p.out + (long long)b * p.out_bs + (long long)row0 * p.out_ld + te;
  if (!LAST) {
    if (!kh) {         // x = (x + rs[:H]) * mask
      float* xw = p.xo + (long long)b * p.xo_bs + (long long)row0 * p.xo_ld + te;
#pragma unroll
      for (int r = 0; r < 16; ++r) xw[(long long)((r & 3) + 8 * (r >> 2)) * p.xo_ld] = (pre[r] + fin[r]) * mk;
    } else {           // out += rs[H:]   (first layer: pre = 0)
#pragma unroll
      for (int r = 0; r < 16; ++r) ob[(long long)((r & 3) + 8 * (r >> 2)) * p.out_ld] = p.first ? fin[r] : pre[r] + fin[r];
    }
  } else if (!kh) {    // last layer: out = (out + rs) * mask
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = p.first ? fin[r] : pre[r] + fin[r];
      ob[(long long)((r & 3) + 8 * (r >> 2)) * p.out_ld] = v * mk;
    }
  }
  dump();
}

// F(2,5) image of one in_layer: [pair 6][K half 2][k-step 24][product 6][lane 64][row tile 4]; lane = (k4 = lane / 16, r = lane % 16)
// holds U_p[row(rt, r)][channel 96 kh + 4 ks + k4] for the four 16-row tiles rt of the pair (tanh rows 32 pi + 16 (rt & 1) + r,
// sigmoid rows H + the same): one 16-byte load = the A operands of the four MFMAs of a (k-step, product).
__global__ void pack_wn_f25_kernel(const float* __restrict__ src, const float* __restrict__ scale, float* __restrict__ img, long long total) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int rt = (int)(e & 3);
  const int lane = (int)((e >> 2) & 63);
  long long rest = e >> 8;
  const int pp = (int)(rest % 6); rest /= 6;
  const int ks = (int)(rest % WNF_KS); rest /= WNF_KS;
  const int kh = (int)(rest & 1);
  const int pi = (int)(rest >> 1);
  const int row = (rt >> 1) * WNF_H + pi * 32 + 16 * (rt & 1) + (lane & 15);
  const int chan = 96 * kh + 4 * ks + (lane >> 4);
  const float* w = src + ((long long)row * WNF_H + chan) * 5;
  const float sc = scale ? scale[row] : 1.0f;
  const float w0 = w[0] * sc, w1 = w[1] * sc, w2 = w[2] * sc, w3 = w[3] * sc, w4 = w[4] * sc;
  float val;
  switch (pp) {
    case 0: val = 0.25f * w0; break;
    case 1: val = -(((w0 + w2) + w4) + (w1 + w3)) * (1.0f / 6.0f); break;
    case 2: val = -(((w0 + w2) + w4) - (w1 + w3)) * (1.0f / 6.0f); break;
    case 3: val = (((w0 + 4.f * w2) + 16.f * w4) + (2.f * w1 + 8.f * w3)) * (1.0f / 24.0f); break;
    case 4: val = (((w0 + 4.f * w2) + 16.f * w4) - (2.f * w1 + 8.f * w3)) * (1.0f / 24.0f); break;
    default: val = w4; break;
  }
  img[e] = val;
}
__global__ void wnf_scale_kernel(const float* __restrict__ v, const float* __restrict__ g, float* __restrict__ scale, long long inner) {
  __shared__ float red[256];
  const int i = blockIdx.x;
  const float* q = v + (long long)i * inner;
  float s = 0.f;
  for (long long k = threadIdx.x; k < inner; k += blockDim.x) s += q[k] * q[k];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) scale[i] = g[i] / sqrtf(red[0]);
}

bool wn_layer_prefers_unfused(int B, int T) {
  static const bool small_on = !(getenv("SVOC_WN_SMALL") && atoi(getenv("SVOC_WN_SMALL")) == 0);
  static const int tiles = getenv("SVOC_WN_SMALL_TILES") ? atoi(getenv("SVOC_WN_SMALL_TILES")) : 0;      // tunable: 32-column tiles below which a layer is "short"
  // default 3/8 of the CUs (96 tiles; half of them until round 5): ms per infer call with the short-input chain | with the persistent stack launch at
  // 72 tiles (9 x 256) 9.68 | 10.06, 80 (5 x 512) 10.18 | 10.58, 96 (6 x 512) 12.37 | 11.85, 112 (7 x 512) 13.79 | 13.33: profiles/r05_wn_short_input_threshold.txt
  return small_on && (long long)variant_batch(B) * ((T + 31) / 32) < (tiles > 0 ? tiles : device_cu_count() * 3 / 8);
}
bool wn_f25_enabled() {
  static const bool on = !(getenv("SVOC_WN_F25") && atoi(getenv("SVOC_WN_F25")) == 0);      // SVOC_WN_F25=0: the direct-form layer kernels
  return on;
}
// Builds the F(2,5) image of `prefix` (an in_layer of a WN with H = 192, k = 5) when the form applies; leaves `img` empty otherwise.
int pack_wn_f25_named(DevBuf& img, int H, int K, int dil, const TensorTable& tab, const std::string& prefix, hipStream_t st) {
  if (!wn_f25_enabled() || H != WNF_H || K != 5 || dil != 1) return SVOC_OK;
  const svoc_tensor* w = tab.find(prefix + ".weight");
  const svoc_tensor* v = tab.find(prefix + ".weight_v");
  const svoc_tensor* g = tab.find(prefix + ".weight_g");
  const svoc_tensor* src = w ? w : v;
  if (!src || (!w && !g)) SVOC_FAIL(SVOC_ERR_MISSING_TENSOR, "missing tensor %s.weight / .weight_v / .weight_g", prefix.c_str());
  if (src->ndim != 3 || src->shape[0] != 2 * H || src->shape[1] != H || src->shape[2] != 5) SVOC_FAIL(SVOC_ERR_SHAPE, "tensor %s has the wrong shape", src->name);
  const long long total = (long long)WNF_NP * 2 * WNF_KS * 6 * 256;
  SVOC_TRY(img.ensure((size_t)(total + 1024) * sizeof(float)));
  SVOC_HIP(hipMemsetAsync(img.f() + total, 0, 1024 * sizeof(float), st));
  DevBuf scale;
  if (!w) {
    SVOC_TRY(scale.ensure((size_t)2 * H * sizeof(float)));
    hipLaunchKernelGGL(wnf_scale_kernel, dim3((unsigned)(2 * H)), dim3(256), 0, st, (const float*)src->data, (const float*)g->data, scale.f(), (long long)H * 5);
  }
  hipLaunchKernelGGL(pack_wn_f25_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const float*)src->data, w ? nullptr : scale.f(), img.f(), total);
  SVOC_HIP(hipGetLastError());
  SVOC_HIP(hipStreamSynchronize(st));                      // `scale` is freed on return
  return SVOC_OK;
}

// (A compile-time specialised copy of the K-split kernel - immediate-offset fragment reads, uniform-base weight stream, static
// group list as in conv_wino.hip - measured 73.8 us per layer against 72.1 us for this one: the layer is not bound by the
// instructions around its MFMAs but by the serial phases of the ONE workgroup a CU holds: staging, two exchanges, gate,
// epilogue and the launch / drain between dependent layers add ~26 us to 46 us of matrix-pipe time.  What did help: the
// gate from hardware exp2 / rcp (79 -> 72 us).  Requesting the first weight group of each GEMM, the res_skip bias and the
// epilogue's operands (residual / skip accumulator / mask) ahead of the phases that precede their use: 72 -> 76 us, dropped.)
// Returns 1 when the fused layer does not apply (caller runs in_layer and res_skip as two convolutions).
int launch_wn_layer_fused(const PackedConv& in_l, const PackedConv& rs_l, int H, const float* x, long long x_bs, int x_ld,
                          float* xo, long long xo_bs, int xo_ld, float* out, long long out_bs, int out_ld, const float* mask,
                          long long mask_bs, const float* gadd, long long gadd_bs, int gadd_ld, int gadd_ts, int first, int last,
                          int B, int T, hipStream_t st, const float* wpf) {
  static const bool enabled = !(getenv("SVOC_FUSE_WN") && atoi(getenv("SVOC_FUSE_WN")) == 0);
  if (!enabled) return 1;
  if (H % 32 != 0 || H / 32 > 8 || !in_l.paired || in_l.Cin != H || in_l.Cout != 2 * H || rs_l.Cin != H) return 1;
  if (rs_l.ktaps != 1 || (last ? rs_l.Cout != H : rs_l.Cout != 2 * H)) return 1;
  const int npairs = H / 32;
  WnArgs a;
  a.x = x; a.x_bs = x_bs; a.x_ld = x_ld;
  a.xo = xo; a.xo_bs = xo_bs; a.xo_ld = xo_ld;
  a.out = out; a.out_bs = out_bs; a.out_ld = out_ld;
  a.mask = mask; a.mask_bs = mask_bs;
  a.gadd = gadd; a.gadd_bs = gadd_bs; a.gadd_ld = gadd_ld; a.gadd_ts = gadd_ts;
  a.wp1 = in_l.wp.f(); a.bias1 = in_l.bias.f(); a.ksg1 = in_l.ksg_total; a.dil = in_l.dil; a.pad = in_l.pad;
  a.wp2 = rs_l.wp.f(); a.bias2 = rs_l.bias.f(); a.ksg2 = rs_l.ksg_total;
  a.H = H; a.ktaps = in_l.ktaps; a.nchunks = H / KC; a.npairs = npairs; a.T = T;
  a.first = first; a.last = last;
  a.wpf = wpf;
  a.dbg = debug_stamp_buffer();
  // narrow tiles when there are few columns: every CU should get a workgroup
  const int ncu = device_cu_count();
  {   // Short inputs: a fused layer is one workgroup per 32 columns carrying the whole K = H*k chain of every row pair
      // (76 us at H=192, k=5 whatever the batch).  With fewer than ncu/2 such workgroups the chip is mostly idle and
      // the layer is latency-bound: run in_layer and res_skip as two K-split convolutions instead (conv_ksplit_kernel:
      // one row pair per workgroup, K split over its four waves: 42 + 84 workgroups with 240 / 24-MFMA chains at 1 x 200).
    if (wn_layer_prefers_unfused(B, T)) return 1;
  }
  // 64-column tiles (the six-wave direct-form kernel) once they still give every CU two workgroups - except where the Winograd F(2,5) layer
  // kernel applies: it exists for 32-column tiles only and issues 3/5 of the in_layer's multiply-adds (round 5: long inputs such as 8 x 4096
  // took the direct-form kernel here until then)
  const bool f25_shape = wpf != nullptr && wn_f25_enabled() && H == 192 && in_l.ktaps == 5 && in_l.dil == 1 && rs_l.ksg_total == 24;
  const int NR = (!f25_shape && (long long)variant_batch(B) * ((T + 63) / 64) >= 2LL * ncu) ? 2 : 1;
  const int NA = NR * 32;
  const int minoff = -in_l.pad, maxoff = (in_l.ktaps - 1) * in_l.dil - in_l.pad;
  a.xoff0 = minoff & ~3;
  a.xrow = round_up(NA + maxoff - a.xoff0, 4);
  a.arow = NA + 1;
  size_t lds = (size_t)H * std::max(a.xrow, a.arow) * sizeof(float);
  // short inputs: 2*npairs waves per workgroup, K split in two halves (wn_layer_fused_ks_kernel)
  const bool ksplit = NR == 1 && (a.nchunks % 2) == 0 && 2 * npairs * 64 <= 768;
  if (ksplit) lds += (size_t)2 * npairs * 16 * 64 * sizeof(float);
  if (lds > 160 * 1024) return 1;
  dim3 grid((T + NA - 1) / NA, 1, B);
  const double flops = (in_l.flops_per_col + rs_l.flops_per_col) * (double)B * (double)T;
  const bool ct = ksplit && H == 192 && in_l.ktaps == 5 && in_l.dil == 1 && a.xrow == 40 && a.arow == 33 && a.nchunks == 6 &&
                  (long long)in_l.mtiles * in_l.ksg_total * 1024 < (1LL << 31);
  const bool f25 = ksplit && ct && wpf != nullptr && wn_f25_enabled() && rs_l.ksg_total == 24;
  // F(2,5): 6 products per two outputs instead of 10 for the in_layer
  stats_add_conv(flops, 2, f25 ? (0.6 * in_l.flops_per_col + rs_l.flops_per_col) * (double)B * (double)T : -1.0);
  int prof_idx = -1;
  if (prof_enabled()) {
    char d[160];
    snprintf(d, sizeof(d), "fusedWN H%-4d k%-2d d%-2d N%-7d B%-3d NA%d%s%s%s", H, in_l.ktaps, in_l.dil, T, B, NA, ksplit ? " ksplit" : "", f25 ? " F(2,5)" : "", last ? " last" : "");
    prof_idx = prof_begin(st, d, flops);
  }
  if (f25) {
    const size_t ldsf = (size_t)WNF_LDS_FLOATS * sizeof(float);
    if (last) {
      auto kern = wn_layer_f25_kernel<true>;
      SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
      hipLaunchKernelGGL(kern, grid, dim3(768), ldsf, st, a);
    } else {
      auto kern = wn_layer_f25_kernel<false>;
      SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
      hipLaunchKernelGGL(kern, grid, dim3(768), ldsf, st, a);
    }
  } else if (ksplit && ct) {
    if (last) {
      auto kern = wn_layer_fused_ks_kernel<true, true>;
      SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
      hipLaunchKernelGGL(kern, grid, dim3(2 * npairs * 64), lds, st, a);
    } else {
      auto kern = wn_layer_fused_ks_kernel<true, false>;
      SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
      hipLaunchKernelGGL(kern, grid, dim3(2 * npairs * 64), lds, st, a);
    }
  } else if (ksplit) {
    auto kern = wn_layer_fused_ks_kernel<false, false>;
    SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
    hipLaunchKernelGGL(kern, grid, dim3(2 * npairs * 64), lds, st, a);
  } else if (NR == 2) {
    auto kern = wn_layer_fused_kernel<2>;
    SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
    hipLaunchKernelGGL(kern, grid, dim3(npairs * 64), lds, st, a);
  } else {
    auto kern = wn_layer_fused_kernel<1>;
    SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
    hipLaunchKernelGGL(kern, grid, dim3(npairs * 64), lds, st, a);
  }
  prof_end(st, prof_idx);
  SVOC_HIP(hipGetLastError());
  return SVOC_OK;
}

}  // namespace svoc
