// WN layers for SHORT inputs (round 4; reference modules.py:148-176) - one launch per layer where the unfused path needs two.
//
// Below ~half a 32-column tile per CU a fused layer kernel (wn_fused.hip: one workgroup owns all 384 in_layer rows of its columns)
// leaves most of the chip idle, so short inputs ran every layer as two dependent K-split convolutions: in_layer + gate (20 us at
// 1 x 200: 42 workgroups, 240-MFMA chains) and the 1 x 1 res_skip (12.5 us, all latency): 48 x 32.5 us = 1.55 of the 3.9 ms of a
// 1 x 200 call.  Here a workgroup is (32-column tile, ROW PAIR pi of the in_layer): six workgroups per column tile.  The 1 x 1 of the
// PREVIOUS layer moves to the head of the kernel - it is a 192 x 192 GEMM on the 36 columns the k = 5 convolution reads, cheap enough
// for each of the six workgroups of a tile to recompute (1728 MFMAs of 16x16x4 over 12 waves) - so that x_i never makes a global round
// trip between the 1 x 1 and the in_layer:
//     x_i   = (x_{i-1} + rs_{i-1}[:H]) * mask     on columns t0 - 8 .. t0 + 39 (LDS; rows 32 pi .. of the centre also to global)
//     out  += rs_{i-1}[H:]                         rows 32 pi .. of the centre columns
//     acts_i = tanh(in_i(x_i)[:H] + g) * sigmoid(in_i(x_i)[H:] + g)      rows 32 pi .. of the centre columns -> global
// with in_i in Winograd F(2,5) form (wn_fused.hip's image and matrices; 1152 MFMAs per workgroup, the K dimension split over the
// twelve waves: 96 each, partial sums of the OUTPUT-transformed tiles reduced through LDS).  The last layer's 1 x 1 runs as the usual
// convolution behind the last kernel.  H = 192, k = 5, dilation 1.
// Round 6 (mid-size batches): template parameter PP = row pairs per workgroup.  With PP = 1 a batch of 43 .. 85 tiles (3 x 512 ..
// 5 x 512) is 258 .. 510 workgroups - two rounds on 256 CUs, 44.6 us per layer at 4 x 512 where 2 x 512 takes 24.7.  PP = 2: three workgroups per
// tile, each staging and recomputing the residual half ONCE and then running the input transform, the in_layer stream, the reduction and the gate for
// its two pairs one after the other (the reduction area aliases the planes, so the transform runs again for the second pair: 0.8 us) - one round.
// Per row the arithmetic and its order are PP = 1's: bit-identical results.
#include "svoc_internal.h"
#include "wino_common.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace svoc {

typedef float wns_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int wns_u32x4 __attribute__((ext_vector_type(4)));
constexpr int WNS_H = 192, WNS_XROW = 40, WNS_NQ = 16, WNS_PLANE = WNS_H * WNS_NQ, WNS_AROW = 48;
constexpr int WNS_XT = WNS_H * WNS_XROW, WNS_PLN = 6 * WNS_PLANE, WNS_AT = WNS_H * WNS_AROW;
constexpr int WNS_LDS_FLOATS = WNS_XT + WNS_PLN + WNS_AT + 64 + 4 * WNS_H;       // + mask tile [48] + the in_layer's bias [2 H] + the previous res_skip's [2 H]
static_assert(12 * 32 * 64 <= WNS_PLN + WNS_AT, "the reduction area aliases the planes and the acts tile");

struct WnSmallArgs {
  const float* x; long long x_bs; int x_ld;            // x_{i-1}  [B][H][x_ld]
  const float* ap; long long ap_bs; int ap_ld;         // acts_{i-1} (null on the first layer)
  float* xo; long long xo_bs; int xo_ld;               // x_i out
  float* out; long long out_bs; int out_ld;            // skip accumulator
  float* ao; long long ao_bs; int ao_ld;               // acts_i out
  const float* mask; long long mask_bs;
  const float* gadd; long long gadd_bs; int gadd_ld; int gadd_ts;
  const float* wpf; const float* bias1;                // in_layer i: F(2,5) image (wn_fused.hip), bias in paired tile order
  const float* wrs;                                    // res_skip i-1: 16x16x4 image + natural-order bias (pack_wn_rs16)
  int T; int skip_first;                               // skip_first: res_skip i-1 is the stack's first one (out = ..., not +=)
  long long* dbg;                                      // optional [workgroup][16] wall-clock stamps of thread 0 (tools/wn_small_timeline.py)
};

template <int PP>
__global__ void __launch_bounds__(768) wn_small_f25_kernel(const WnSmallArgs p) {
  constexpr int H = WNS_H;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const XT = lds;                                    // x tile [H][40]: columns t0 - 4 .. t0 + 35
  float* const PLN = lds + WNS_XT;                          // V_p [6][H][16]
  float* const AT = PLN + WNS_PLN;                          // acts_{i-1} tile [H][48]: columns t0 - 8 .. t0 + 39
  float* const MK = AT + WNS_AT;                            // mask of those 48 columns
  // the two bias vectors the kernel adds (round 6): requested at the top, parked here by the staging pass - read from global memory where they are used they were
  // exposed L2 round trips behind the 1 x 1's MFMAs and in the gate
  float* const BI = MK + 64;                                // in_layer i: [2 H] paired tile order
  float* const BR = BI + 2 * WNS_H;                         // res_skip i-1: [2 H] natural order (residual | skip)
  float* const RED = PLN;                                   // [12 waves][32][64] partial outputs (after the stream)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, k4 = lane >> 4;
  const int pi0 = blockIdx.y * PP;                          // this workgroup's row pairs: pi0 .. pi0 + PP - 1
  const int b = blockIdx.z;
  const int t0 = blockIdx.x * 32;
  const int T = p.T;
  const bool has_prev = p.ap != nullptr;
  const bool stamped = p.dbg != nullptr && tid == 0;
  const long long wg_ = ((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  auto stamp = [&](int i) { if (stamped) p.dbg[wg_ * 16 + i] = (long long)__builtin_amdgcn_s_memrealtime(); };
  stamp(0);

  // ---- the A operands of the previous layer's res_skip for this wave (12 + 12 sixteen-byte loads) are requested first: their L2
  // round trips run under the staging of the tiles (requested inside the GEMM loops they cost 12 exposed latencies: +7 us per layer)
  const __amdgpu_buffer_rsrc_t rs16 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wrs), 0, 0x7fffffff, 0x00020000);
  auto rsload = [&](int soff) -> float4 {
    const wns_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs16, lane * 16, soff, 0);
    return *reinterpret_cast<const float4*>(&t);
  };
  const float bias_v = tid < 2 * H ? p.bias1[tid] : (has_prev ? p.wrs[24 * 12 * 256 + tid - 2 * H] : 0.f);      // 768 threads = 2 H + 2 H values
  float4 awx[12], aws[12];
  if (has_prev) {
    const int wbx = __builtin_amdgcn_readfirstlane(wave * 12 * 1024);
#pragma unroll
    for (int ks4 = 0; ks4 < 12; ++ks4) awx[ks4] = rsload(wbx + ks4 * 1024);
    if (wave < 4 * PP) {                                    // skip part: waves 4 j .. 4 j + 3 take pair pi0 + j
      const int wbs = __builtin_amdgcn_readfirstlane((12 + 2 * (pi0 + (wave >> 2)) + (wave & 1)) * 12 * 1024);
#pragma unroll
      for (int ks4 = 0; ks4 < 12; ++ks4) aws[ks4] = rsload(wbs + ks4 * 1024);
    }
  }
  // ---- stage x_{i-1} (columns t0 - 4 .. t0 + 35), acts_{i-1} (t0 - 8 .. t0 + 39) and the mask; zero outside [0, T)
  {
    constexpr int R4 = WNS_XROW / 4, total = H * R4;
    const float* xb = p.x + (long long)b * p.x_bs;
    const bool vec = ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0) && (p.x_ld & 3) == 0 && (p.x_bs & 3) == 0;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int it = tid + 768 * u;
      if (it < total) {
        const int c = it / R4, g4 = it - c * R4;
        const int t = t0 - 4 + 4 * g4;
        const float* row = xb + (long long)c * p.x_ld;
        float4 q;
        if (vec && t >= 0 && t + 3 < T) q = *reinterpret_cast<const float4*>(row + t);
        else {
          q.x = (t >= 0 && t < T) ? row[t] : 0.f;
          q.y = (t + 1 >= 0 && t + 1 < T) ? row[t + 1] : 0.f;
          q.z = (t + 2 >= 0 && t + 2 < T) ? row[t + 2] : 0.f;
          q.w = (t + 3 >= 0 && t + 3 < T) ? row[t + 3] : 0.f;
        }
        *reinterpret_cast<float4*>(XT + c * WNS_XROW + 4 * g4) = q;
      }
    }
    if (has_prev) {
      constexpr int A4 = WNS_AROW / 4, atotal = H * A4;      // 2304 groups: three per thread
      const float* ab = p.ap + (long long)b * p.ap_bs;
      const bool avec = ((reinterpret_cast<uintptr_t>(p.ap) & 15) == 0) && (p.ap_ld & 3) == 0 && (p.ap_bs & 3) == 0;
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int it = tid + 768 * u;
        if (it < atotal) {
          const int c = it / A4, g4 = it - c * A4;
          const int t = t0 - 8 + 4 * g4;
          const float* row = ab + (long long)c * p.ap_ld;
          float4 q;
          if (avec && t >= 0 && t + 3 < T) q = *reinterpret_cast<const float4*>(row + t);
          else {
            q.x = (t >= 0 && t < T) ? row[t] : 0.f;
            q.y = (t + 1 >= 0 && t + 1 < T) ? row[t + 1] : 0.f;
            q.z = (t + 2 >= 0 && t + 2 < T) ? row[t + 2] : 0.f;
            q.w = (t + 3 >= 0 && t + 3 < T) ? row[t + 3] : 0.f;
          }
          *reinterpret_cast<float4*>(AT + c * WNS_AROW + 4 * g4) = q;
        }
      }
      if (tid < WNS_AROW) {
        const int t = t0 - 8 + tid;
        MK[tid] = (t >= 0 && t < T) ? p.mask[(long long)b * p.mask_bs + t] : 0.f;
      }
    }
  }
  BI[tid] = bias_v;                                          // (BR directly behind BI)
  __syncthreads();
  stamp(1);

  if (has_prev) {
    const float* rsbias = BR;                               // natural order: [0, H) residual part, [H, 2H) skip part
    // ---- skip part of res_skip_{i-1}: rows H + 32 pi .. + 31 on the 32 centre columns: waves 0..3 (+ 4 j: pair pi0 + j) = (row tile, column tile)
    if (wave < 4 * PP) {
      const int pi = pi0 + (wave >> 2);
      const int rt2 = wave & 1, nt2 = (wave >> 1) & 1;
      wns_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const float* bp = AT + k4 * WNS_AROW + 8 + 16 * nt2 + col;
#pragma unroll
      for (int ks4 = 0; ks4 < 12; ++ks4) {
        const float4 a = aws[ks4];
        const float* bq = bp + 16 * ks4 * WNS_AROW;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bq[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bq[4 * WNS_AROW], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bq[8 * WNS_AROW], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bq[12 * WNS_AROW], acc, 0, 0, 0);
      }
      const int t = t0 + 16 * nt2 + col;
      if (t < T) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 32 * pi + 16 * rt2 + 4 * k4 + i;
          float* o = p.out + (long long)b * p.out_bs + (long long)r * p.out_ld + t;
          const float v = acc[i] + rsbias[H + r];
          *o = p.skip_first ? v : *o + v;
        }
      }
    }
    // ---- residual part on all 48 columns: wave w = row tile w (rows 16 w ..), three column tiles; x_i -> the LDS tile in place
    {
      wns_f32x4 acc[3];
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) acc[nt] = (wns_f32x4){0.f, 0.f, 0.f, 0.f};
      const float* bp = AT + k4 * WNS_AROW + col;
#pragma unroll
      for (int ks4 = 0; ks4 < 12; ++ks4) {
        const float4 a = awx[ks4];
        const float* bq = bp + 16 * ks4 * WNS_AROW;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float av = wino_pick(a, j);
#pragma unroll
          for (int nt = 0; nt < 3; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bq[4 * j * WNS_AROW + 16 * nt], acc[nt], 0, 0, 0);
        }
      }
      const bool store = (wave >> 1) >= pi0 && (wave >> 1) < pi0 + PP;      // the workgroup of pair pi writes rows 32 pi .. of x_i
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) {
        const int ac = 16 * nt + col;                       // column of the acts tile; x tile column = ac - 4
        const int xc = ac - 4;
        const float mk = MK[ac];
        if (xc >= 0 && xc < WNS_XROW) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 16 * wave + 4 * k4 + i;
            float* xp = XT + r * WNS_XROW + xc;
            const float v = (*xp + (acc[nt][i] + rsbias[r])) * mk;
            *xp = v;
            const int t = t0 - 8 + ac;
            if (store && ac >= 8 && ac < 40 && t < T) p.xo[(long long)b * p.xo_bs + (long long)r * p.xo_ld + t] = v;
          }
        }
      }
    }
    __syncthreads();
  }
  stamp(2);

#pragma unroll 1
  for (int pi = pi0; pi < pi0 + PP; ++pi) {
  const int sb = 3 + 4 * (pi - pi0);
  if (pi > pi0) __syncthreads();                             // the reduction of the pair before is done with the area the planes live in
  // ---- input transform of x_i (wn_fused.hip): window q of channel c reads tile columns 2q + 2 .. 2q + 7
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = (tid >> 4) + 48 * u, q = tid & 15;
    const float* r = XT + c * WNS_XROW + 2 * q + 2;
    const float2 f0 = *reinterpret_cast<const float2*>(r), f1 = *reinterpret_cast<const float2*>(r + 2), f2 = *reinterpret_cast<const float2*>(r + 4);
    const float d0 = f0.x, d1 = f0.y, d2 = f1.x, d3 = f1.y, d4 = f2.x, d5 = f2.y;
    const float a_ = __builtin_fmaf(-4.f, d2, d4), b_ = __builtin_fmaf(-4.f, d1, d3);
    const float c_ = d4 - d2, e_ = 2.f * (d3 - d1);
    float* o = PLN + c * WNS_NQ + q;
    o[0] = __builtin_fmaf(4.f, d0, __builtin_fmaf(-5.f, d2, d4));
    o[WNS_PLANE] = a_ + b_;
    o[2 * WNS_PLANE] = a_ - b_;
    o[3 * WNS_PLANE] = c_ + e_;
    o[4 * WNS_PLANE] = c_ - e_;
    o[5 * WNS_PLANE] = __builtin_fmaf(4.f, d1, __builtin_fmaf(-5.f, d3, d5));
  }
  __syncthreads();
  stamp(sb);

  // ---- in_layer of pair pi, K split over the twelve waves: wave w = k-steps 4 w .. 4 w + 3 of the 48 (channels 16 w .. 16 w + 15),
  // all four 16-row tiles and six products: 96 MFMAs from wn_fused.hip's image [pair][half][k-step 24][product][lane][tile]
  wns_f32x4 M[4][6];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int q = 0; q < 6; ++q) M[rt][q] = (wns_f32x4){0.f, 0.f, 0.f, 0.f};
  {
    constexpr int NST = 24;
    const __amdgpu_buffer_rsrc_t rsf = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpf), 0, 0x7fffffff, 0x00020000);
    const int kh = wave / 6, ks0 = 4 * (wave - 6 * kh);
    const int w0 = __builtin_amdgcn_readfirstlane(((pi * 2 + kh) * 144 + ks0 * 6) * 1024);
    const unsigned wlane = (unsigned)lane * 16u;
    const unsigned baddr0 = (unsigned)(size_t)PLN + (unsigned)(((16 * wave + k4) * WNS_NQ + col) * 4);
    const unsigned baddr1 = baddr0 + 3u * WNS_PLANE * 4u;
    float4 a[4];
    float fb[2];
    auto wload = [&](float4& d, int soff) {
      const wns_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsf, (int)wlane, soff, 0);
      d = *reinterpret_cast<const float4*>(&t);
    };
    auto rdb = [&](auto ic) {
      constexpr int I = decltype(ic)::value;
      if constexpr (I < NST) {
        constexpr int KS_ = I / 6, P_ = I % 6;
        constexpr int O = ((P_ % 3) * WNS_PLANE + KS_ * 4 * WNS_NQ) * 4;
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
    rqw(std::integral_constant<int, 0>{}); rqw(std::integral_constant<int, 1>{}); rqw(std::integral_constant<int, 2>{});
    rdb(std::integral_constant<int, 0>{}); rdb(std::integral_constant<int, 1>{});
    wino_static_for<0, NST>(step);
  }
  stamp(sb + 1);
  __syncthreads();                                           // every wave is done with the planes: the reduction area takes their place
  // ---- output transform of the partial sums -> RED[wave][j][lane], j = 8 rt + 2 i + o (rt: tanh lo, tanh hi, sigmoid lo, sigmoid hi)
  {
    float* rm = RED + (wave * 32) * 64 + lane;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float s12 = M[rt][1][i] + M[rt][2][i], d12 = M[rt][1][i] - M[rt][2][i];
        const float s34 = M[rt][3][i] + M[rt][4][i], d34 = M[rt][3][i] - M[rt][4][i];
        rm[(8 * rt + 2 * i) * 64] = M[rt][0][i] + (s12 + s34);
        rm[(8 * rt + 2 * i + 1) * 64] = __builtin_fmaf(2.f, d34, d12) + M[rt][5][i];
      }
  }
  __syncthreads();
  stamp(sb + 2);
  // ---- reduction over the twelve K parts + bias + gate: waves 0..7 take two (tanh, sigmoid) pairs per lane each
  if (wave < 8) {
    const float* gb = p.gadd ? p.gadd + (long long)b * p.gadd_bs : nullptr;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = 2 * wave + jj;                          // 0..15: rt = j / 8 (tanh lo / hi), i = (j % 8) / 2, o = j % 2
      float vA = 0.f, vB = 0.f;
#pragma unroll
      for (int w = 0; w < 12; ++w) {
        vA += RED[(w * 32 + j) * 64 + lane];
        vB += RED[(w * 32 + 16 + j) * 64 + lane];
      }
      const int rr = 16 * (j >> 3) + 4 * k4 + ((j & 7) >> 1);
      const int chn = 32 * pi + rr;
      const int m = 2 * col + (j & 1);
      const int t = t0 + m;
      vA += BI[(2 * pi) * 32 + rr];
      vB += BI[(2 * pi + 1) * 32 + rr];
      if (gb) {
        const int tc = min(t, T - 1);
        vA += gb[(long long)chn * p.gadd_ld + (long long)tc * p.gadd_ts];
        vB += gb[(long long)(H + chn) * p.gadd_ld + (long long)tc * p.gadd_ts];
      }
      if (t < T) p.ao[(long long)b * p.ao_bs + (long long)chn * p.ao_ld + t] = gate_tanh_sigmoid(vA, vB);
    }
  }
  stamp(sb + 3);
  }
}

// res_skip of a non-last layer (2H x H x 1) as A operands of v_mfma_f32_16x16x4_f32: [row tile 24][k-step group 12][lane][4]; lane =
// (k4 = lane / 16, r = lane % 16) holds W[16 rt + r][16 ks4 + 4 j + k4] in component j; behind it the bias in natural order [2H].
__global__ void pack_wn_rs16_kernel(const float* __restrict__ src, const float* __restrict__ scale, const float* __restrict__ bias,
                                    float* __restrict__ img, int total, int rows) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total + rows) return;
  if (e >= total) { img[e] = bias ? bias[e - total] : 0.f; return; }
  const int j = e & 3, lane = (e >> 2) & 63, ks4 = (e >> 8) % 12, rt = (e >> 8) / 12;
  const int row = 16 * rt + (lane & 15), ch = 16 * ks4 + 4 * j + (lane >> 4);
  img[e] = src[(long long)row * WNS_H + ch] * (scale ? scale[row] : 1.0f);
}
__global__ void wns_scale_kernel(const float* __restrict__ v, const float* __restrict__ g, float* __restrict__ scale, int inner) {
  __shared__ float red[256];
  const int i = blockIdx.x;
  float s = 0.f;
  for (int k = threadIdx.x; k < inner; k += blockDim.x) s += v[(long long)i * inner + k] * v[(long long)i * inner + k];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) scale[i] = g[i] / sqrtf(red[0]);
}

bool wn_small_enabled() {
  static const bool on = wn_f25_enabled() && !(getenv("SVOC_WN_SMALL_F25") && atoi(getenv("SVOC_WN_SMALL_F25")) == 0);
  return on;
}
// Image of `prefix` (a res_skip layer with 2H output rows of a WN with H = 192, or the last layer's with H rows: wn_mesh.hip); leaves `img`
// empty when the kernel does not apply.
int pack_wn_rs16_named(DevBuf& img, int H, int Cout, const TensorTable& tab, const std::string& prefix, hipStream_t st) {
  if (!wn_small_enabled() || H != WNS_H || (Cout != 2 * H && Cout != H)) return SVOC_OK;
  const svoc_tensor* w = tab.find(prefix + ".weight");
  const svoc_tensor* v = tab.find(prefix + ".weight_v");
  const svoc_tensor* g = tab.find(prefix + ".weight_g");
  const svoc_tensor* bs = tab.find(prefix + ".bias");
  const svoc_tensor* src = w ? w : v;
  if (!src || (!w && !g)) SVOC_FAIL(SVOC_ERR_MISSING_TENSOR, "missing tensor %s.weight / .weight_v / .weight_g", prefix.c_str());
  if (src->ndim != 3 || src->shape[0] != Cout || src->shape[1] != H || src->shape[2] != 1) SVOC_FAIL(SVOC_ERR_SHAPE, "tensor %s has the wrong shape", src->name);
  const int total = (Cout / 16) * 12 * 256;
  SVOC_TRY(img.ensure((size_t)(total + Cout + 1024) * sizeof(float)));
  DevBuf scale;
  if (!w) {
    SVOC_TRY(scale.ensure((size_t)Cout * sizeof(float)));
    hipLaunchKernelGGL(wns_scale_kernel, dim3((unsigned)Cout), dim3(256), 0, st, (const float*)src->data, (const float*)g->data, scale.f(), H);
  }
  hipLaunchKernelGGL(pack_wn_rs16_kernel, dim3((unsigned)((total + Cout + 255) / 256)), dim3(256), 0, st, (const float*)src->data,
                     w ? nullptr : scale.f(), bs ? (const float*)bs->data : nullptr, img.f(), total, Cout);
  SVOC_HIP(hipGetLastError());
  SVOC_HIP(hipStreamSynchronize(st));
  return SVOC_OK;
}

static bool wn_small_two_pairs() {
  static const bool on = !(getenv("SVOC_WN_SMALL_PP2") && atoi(getenv("SVOC_WN_SMALL_PP2")) == 0);      // SVOC_WN_SMALL_PP2=0: always six workgroups per tile (A/B)
  return on;
}
// One layer of the short-input chain.  ap / wrs null on the stack's first layer.  Returns 1 when the kernel does not apply.
int launch_wn_small_layer(const PackedConv& in_l, const float* wpf, const float* wrs, double rs_flops_per_col, const float* x, long long x_bs,
                          int x_ld, const float* ap, long long ap_bs, int ap_ld, float* xo, long long xo_bs, int xo_ld, float* out, long long out_bs,
                          int out_ld, float* ao, long long ao_bs, int ao_ld, const float* mask, long long mask_bs, const float* gadd,
                          long long gadd_bs, int gadd_ld, int gadd_ts, int skip_first, int B, int T, hipStream_t st) {
  if (!wn_small_enabled() || !wpf || in_l.Cin != WNS_H || in_l.Cout != 2 * WNS_H || in_l.ktaps != 5 || in_l.dil != 1 || !in_l.paired) return 1;
  if ((ap != nullptr) != (wrs != nullptr)) return 1;
  WnSmallArgs a;
  a.x = x; a.x_bs = x_bs; a.x_ld = x_ld;
  a.ap = ap; a.ap_bs = ap_bs; a.ap_ld = ap_ld;
  a.xo = xo; a.xo_bs = xo_bs; a.xo_ld = xo_ld;
  a.out = out; a.out_bs = out_bs; a.out_ld = out_ld;
  a.ao = ao; a.ao_bs = ao_bs; a.ao_ld = ao_ld;
  a.mask = mask; a.mask_bs = mask_bs;
  a.gadd = gadd; a.gadd_bs = gadd_bs; a.gadd_ld = gadd_ld; a.gadd_ts = gadd_ts;
  a.wpf = wpf; a.bias1 = in_l.bias.f(); a.wrs = wrs;
  a.T = T; a.skip_first = skip_first;
  a.dbg = debug_stamp_buffer();
  // algorithmic work: the in_layer and, from the second layer on, the previous layer's 1 x 1; executed: 3/5 of the in_layer, the residual
  // half of the 1 x 1 six times on 48 of 32 columns, its skip half once
  const double fin = in_l.flops_per_col * (double)B * (double)T, frs = ap ? rs_flops_per_col * (double)B * (double)T : 0.0;
  stats_add_conv(fin + frs, ap ? 2 : 1, 0.6 * fin + (0.5 * 6.0 * 1.5 + 0.5) * frs);
  int prof_idx = -1;
  if (prof_enabled()) {
    char d[160];
    snprintf(d, sizeof(d), "smallWN H192  k5  d1  N%-7d B%-3d F(2,5)%s", T, B, ap ? " + res_skip" : "");
    prof_idx = prof_begin(st, d, fin + frs);
  }
  // two row pairs per workgroup when that makes ONE round of workgroups out of two (43 .. 85 tiles on 256 CUs; bit-identical to one pair each)
  const long long tiles = (long long)variant_batch(B) * ((T + 31) / 32);
  const int ncu = device_cu_count();
  if (wn_small_two_pairs() && tiles * 6 > ncu && tiles * 3 <= ncu) {
    auto kern = wn_small_f25_kernel<2>;
    SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
    hipLaunchKernelGGL(kern, dim3((T + 31) / 32, 3, B), dim3(768), (size_t)WNS_LDS_FLOATS * sizeof(float), st, a);
  } else {
    auto kern = wn_small_f25_kernel<1>;
    SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
    hipLaunchKernelGGL(kern, dim3((T + 31) / 32, 6, B), dim3(768), (size_t)WNS_LDS_FLOATS * sizeof(float), st, a);
  }
  prof_end(st, prof_idx);
  SVOC_HIP(hipGetLastError());
  return SVOC_OK;
}

}  // namespace svoc
