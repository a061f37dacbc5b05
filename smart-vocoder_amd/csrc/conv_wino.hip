// Winograd F(2,3) implicit-GEMM convolution for the decoder's big undilated convolutions
// (reference modules.py:190-207: convs2 and the d = 1 member of convs1, C = 128 / 256, k = 3 / 7 / 11).
//
// The fp32 matrix pipe is the roofline of this path (157.3 TFLOP/s, no TF32 on gfx950) and the direct kernels sit at
// 82 % of it, so the remaining lever is to issue fewer MFMAs.  A k-tap convolution is split into groups of three taps
// at tap offsets 0, 4, 8 (k = 3: one group; 7: two + tap 3; 11: three + taps 3 and 7).  Each group is a minimal
// F(2,3) filtering: two outputs y[2q], y[2q+1] from the four inputs d_j = x[2q - pad + 4g + j] with four products
// instead of six,
//     V0 = d0 - d2   V1 = d1 + d2   V2 = d2 - d1   V3 = d1 - d3          (input transform, additions only)
//     U0 = w0        U1 = (w0+w1+w2)/2   U2 = (w0-w1+w2)/2   U3 = w2    (weight transform, at load time)
//     y[2q] = M0 + M1 + M2     y[2q+1] = M1 - M2 - M3      M_p = sum_c U_p[c] * V_p[c]
// Because the tap offsets of the groups are multiples of four and pad is odd, every group reads the SAME transformed
// planes V_p[c][q'] (q' = q + 2g), so all groups accumulate into one set of four transform-domain accumulators; the
// channel reduction M_p is the GEMM the MFMAs do.  The left-over taps (3, 7) are ordinary taps on the de-interleaved
// planes E[q] = x[2q], O[q] = x[2q+1] with two more accumulators (their tap offset minus pad is even, so y[2q] reads
// only E and y[2q+1] only O).  MFMAs per output and channel pair: k=3: 2 (direct 3), k=7: 5 (7), k=11: 8 (11).
// fp32 throughout; the additions of the transforms round once more than the direct form (measured <= 1e-6 relative).
//
// One workgroup = 4 waves = 64 rows x 64 q (128 output columns); a wave owns 32 rows x 32 q: four M tiles + two
// direct tiles (96 accumulator registers).  Per 32-channel chunk: raw tile (leaky-relu'd, zero padded) -> LDS ->
// transform pass -> six planes in LDS -> 16 * (4 G + 2 ND) MFMAs per wave with fragment reads at immediate offsets.
#include "svoc_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace svoc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WinoArgs {
  const float* x; long long x_bs; int x_ld; int Cin; int L;      // input [B][Cin][x_ld], valid columns [0, L)
  float pre_slope;                                                  // leaky-relu applied while staging (1 = none)
  const float* wp; const float* bias; int nchunks; int mtiles;      // transformed weights (pack_wino), bias [32 * mtiles]
  float* y; long long y_bs; int y_ld;                               // output [B][Cout][y_ld]
  const float* res; long long res_bs; int res_ld;                   // F_RES
  unsigned flags; float div;                                        // F_RES | F_ACC | F_DIV
  int ntn; int gy; int xcd;                                         // column tiles per row, row blocks, XCD-aware order
};
struct WinoGroup { WinoArgs a[3]; int end[3]; int k[3]; };

template <int K>
struct WinoGeo {
  static constexpr int G = (K + 1) / 4;                   // three-tap groups at tap offsets 0, 4, 8
  static constexpr int ND = G - 1;                        // left-over single taps (3, 7)
  static constexpr int PAD = (K - 1) / 2;
  static constexpr int SLOTS = 4 * G + ND;                // weight slots per 32-channel chunk (a direct tap feeds E and O)
  static constexpr int NQV = 64 + 2 * (G - 1);            // q' range of the V planes
  static constexpr int NQE = 66;                          // q range of the E / O planes, origin q0 - 1
  static constexpr int PQ = 72;                           // plane row stride (floats)
  static constexpr int NPL = ND > 0 ? 6 : 4;              // planes: V0..V3 (+ E, O)
  static constexpr int XOFF = -8;                         // raw tile starts at n0 + XOFF
  static constexpr int RAW = 144;                         // raw tile columns (multiple of 4)
  static constexpr int RAW_FLOATS = KC * RAW;
  static constexpr int LDS_BYTES = (RAW_FLOATS + NPL * KC * PQ) * 4;
  static constexpr int MFMA_PER_CHUNK = 16 * (4 * G + 2 * ND);
};

template <int OFF>
__device__ __forceinline__ float wino_lds_rd(unsigned addr) {
  float v;
  static_assert(OFF >= 0 && OFF < 65536, "ds_read_b32 offset field");
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ void wino_wait4(float (&b)[4]) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])); }
__device__ __forceinline__ float wino_pick(const float4& v, int s) { return s == 0 ? v.x : (s == 1 ? v.y : (s == 2 ? v.z : v.w)); }

// B fragments of one k-group (4 k-steps = channels 8*KG .. 8*KG+7) of plane PL at column offset COL: lane (l31, hi) reads
// row 8*KG + 2*s + hi.  `baddr` = LDS byte address of planes + (hi * PQ + first q of the wave + l31) * 4.
template <int PQ, int PL, int KG, int COL>
__device__ __forceinline__ void wino_frag(float (&b)[4], unsigned baddr) {
  constexpr int BASE = (PL * KC + 8 * KG) * PQ + COL;
  b[0] = wino_lds_rd<(BASE) * 4>(baddr);
  b[1] = wino_lds_rd<(BASE + 2 * PQ) * 4>(baddr);
  b[2] = wino_lds_rd<(BASE + 4 * PQ) * 4>(baddr);
  b[3] = wino_lds_rd<(BASE + 6 * PQ) * 4>(baddr);
}

template <int K>
__device__ __forceinline__ void wino_tile(const WinoArgs& p, const int bx, const int by, const int bz) {
  using Geo = WinoGeo<K>;
  constexpr int G = Geo::G, ND = Geo::ND, PAD = Geo::PAD, PQ = Geo::PQ, RAW = Geo::RAW, SLOTS = Geo::SLOTS;
  extern __shared__ __attribute__((aligned(16))) float wl[];
  float* const raw = wl;                                   // [KC][RAW]  lrelu(x), zero outside [0, L)
  float* const pl = wl + Geo::RAW_FLOATS;                  // [NPL][KC][PQ]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;
  const int n0 = bx * 128;                                 // first output column of the workgroup
  const int mt = by * 2 + wm;                              // this wave's 32-row tile
  const bool row_ok = mt < p.mtiles;
  const int mtc = row_ok ? mt : p.mtiles - 1;
  const int L = p.L;

  f32x16 M[4], D[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { M[0][i] = 0.f; M[1][i] = 0.f; M[2][i] = 0.f; M[3][i] = 0.f; D[0][i] = 0.f; D[1][i] = 0.f; }

  // ---- raw staging: 32 channels x RAW columns = 36 float4 per channel, 1152 per chunk, 4.5 per thread
  constexpr int R4 = RAW / 4, SU = (KC * R4 + 255) / 256;
  const float* xb = p.x + (long long)bz * p.x_bs;
  const int xs_start = n0 + Geo::XOFF;
  const float slope = p.pre_slope;
  float4 v[SU];
  auto issue = [&](int ch) {
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int idx = tid + u * 256;
      const int c = min(idx / R4, KC - 1), g4 = idx - (idx / R4) * R4;
      const int gc = min(ch * KC + c, p.Cin - 1);
      int t = xs_start + 4 * g4;
      t = (t >= 0 && t + 3 < L) ? t : 0;                   // clamped address; edges are fixed up in publish()
      v[u] = *reinterpret_cast<const float4*>(xb + (long long)gc * p.x_ld + t);
    }
  };
  auto publish = [&](int ch) {
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int idx = tid + u * 256;
      if (idx < KC * R4) {
        const int c = idx / R4, g4 = idx - c * R4;
        const int t = xs_start + 4 * g4;
        float4 q = v[u];
        if (!(t >= 0 && t + 3 < L)) {                      // a group that straddles an end (rare): element-wise
          const float* row = xb + (long long)min(ch * KC + c, p.Cin - 1) * p.x_ld;
          q.x = (t >= 0 && t < L) ? row[t] : 0.f;
          q.y = (t + 1 >= 0 && t + 1 < L) ? row[t + 1] : 0.f;
          q.z = (t + 2 >= 0 && t + 2 < L) ? row[t + 2] : 0.f;
          q.w = (t + 3 >= 0 && t + 3 < L) ? row[t + 3] : 0.f;
        }
        if (ch * KC + c >= p.Cin) q = make_float4(0.f, 0.f, 0.f, 0.f);
        q.x = fmaxf(q.x, q.x * slope); q.y = fmaxf(q.y, q.y * slope);
        q.z = fmaxf(q.z, q.z * slope); q.w = fmaxf(q.w, q.w * slope);
        *reinterpret_cast<float4*>(raw + c * RAW + 4 * g4) = q;
      }
    }
  };
  // ---- transform pass: raw -> V0..V3 (q' = q0 + i, window raw[2i + 8 - PAD .. +3]) and E / O (q = q0 - 1 + i: raw[2i + 6], [2i + 7])
  auto transform = [&]() {
    constexpr int NV2 = Geo::NQV / 2;                      // pairs of q' per channel
    for (int it = tid; it < KC * NV2; it += 256) {
      const int c = it / NV2, i = 2 * (it - c * NV2);
      const float* r = raw + c * RAW + 2 * i + 8 - PAD;
      const float d0 = r[0], d1 = r[1], d2 = r[2], d3 = r[3], d4 = r[4], d5 = r[5];
      float* o = pl + c * PQ + i;
      *reinterpret_cast<float2*>(o) = make_float2(d0 - d2, d2 - d4);
      *reinterpret_cast<float2*>(o + KC * PQ) = make_float2(d1 + d2, d3 + d4);
      *reinterpret_cast<float2*>(o + 2 * KC * PQ) = make_float2(d2 - d1, d4 - d3);
      *reinterpret_cast<float2*>(o + 3 * KC * PQ) = make_float2(d1 - d3, d3 - d5);
    }
    if constexpr (ND > 0) {
      constexpr int NE2 = Geo::NQE / 2;
      for (int it = tid; it < KC * NE2; it += 256) {
        const int c = it / NE2, i = 2 * (it - c * NE2);
        const float* r = raw + c * RAW + 2 * i + 6;
        const float2 a = *reinterpret_cast<const float2*>(r), b = *reinterpret_cast<const float2*>(r + 2);
        float* o = pl + 4 * KC * PQ + c * PQ + i;
        *reinterpret_cast<float2*>(o) = make_float2(a.x, b.x);
        *reinterpret_cast<float2*>(o + KC * PQ) = make_float2(a.y, b.y);
      }
    }
  };

  // ---- MFMA phase of one chunk.  Weight slots (pack_wino order): sigma = 4 g + p for the groups, then the direct taps.
  // All four weight fragments (16 k-steps) of slot sigma + 1 are requested while slot sigma computes.
  const unsigned baddr = (unsigned)(size_t)pl + (unsigned)(hi * PQ + wn * 32 + l31) * 4u;
  const char* const wrow = reinterpret_cast<const char*>(p.wp) + (size_t)mtc * p.nchunks * SLOTS * 4096 + (size_t)lane * 16;
  auto mfma_chunk = [&](int ch) {
    const char* wa = wrow + (size_t)ch * SLOTS * 4096;
    float4 a0[4], a1[4];
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) a0[kg] = *reinterpret_cast<const float4*>(wa + kg * 1024);
    // one group slot: 16 MFMAs into M[P]; fragments one k-group ahead
    auto vslot = [&](auto gc, auto pc, float4(&ac)[4], float4(&an)[4], bool more) {
      constexpr int GG = decltype(gc)::value, P = decltype(pc)::value;
      wa += 4096;
      if (more) {
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) an[kg] = *reinterpret_cast<const float4*>(wa + kg * 1024);
      }
      float b0[4], b1[4];
      wino_frag<PQ, P, 0, 2 * GG>(b0, baddr);
      wino_frag<PQ, P, 1, 2 * GG>(b1, baddr);
      wino_wait4(b0);
#pragma unroll
      for (int s = 0; s < 4; ++s) M[P] = __builtin_amdgcn_mfma_f32_32x32x2f32(wino_pick(ac[0], s), b0[s], M[P], 0, 0, 0);
      wino_frag<PQ, P, 2, 2 * GG>(b0, baddr);
      wino_wait4(b1);
#pragma unroll
      for (int s = 0; s < 4; ++s) M[P] = __builtin_amdgcn_mfma_f32_32x32x2f32(wino_pick(ac[1], s), b1[s], M[P], 0, 0, 0);
      wino_frag<PQ, P, 3, 2 * GG>(b1, baddr);
      wino_wait4(b0);
#pragma unroll
      for (int s = 0; s < 4; ++s) M[P] = __builtin_amdgcn_mfma_f32_32x32x2f32(wino_pick(ac[2], s), b0[s], M[P], 0, 0, 0);
      wino_wait4(b1);
#pragma unroll
      for (int s = 0; s < 4; ++s) M[P] = __builtin_amdgcn_mfma_f32_32x32x2f32(wino_pick(ac[3], s), b1[s], M[P], 0, 0, 0);
    };
    // one direct tap: the same weights feed E -> D[0] (even outputs) and O -> D[1] (odd outputs); COL = 1 + (tap - PAD) / 2
    auto dslot = [&](auto colc, float4(&ac)[4], float4(&an)[4], bool more) {
      constexpr int COL = decltype(colc)::value;
      wa += 4096;
      if (more) {
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) an[kg] = *reinterpret_cast<const float4*>(wa + kg * 1024);
      }
      float e0[4], o0[4], e1[4], o1[4];
      wino_frag<PQ, 4, 0, COL>(e0, baddr); wino_frag<PQ, 5, 0, COL>(o0, baddr);
      wino_frag<PQ, 4, 1, COL>(e1, baddr); wino_frag<PQ, 5, 1, COL>(o1, baddr);
      wino_wait4(e0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        D[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wino_pick(ac[0], s), e0[s], D[0], 0, 0, 0);
        D[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wino_pick(ac[0], s), o0[s], D[1], 0, 0, 0);
      }
      wino_frag<PQ, 4, 2, COL>(e0, baddr); wino_frag<PQ, 5, 2, COL>(o0, baddr);
      wino_wait4(e1);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        D[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wino_pick(ac[1], s), e1[s], D[0], 0, 0, 0);
        D[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wino_pick(ac[1], s), o1[s], D[1], 0, 0, 0);
      }
      wino_frag<PQ, 4, 3, COL>(e1, baddr); wino_frag<PQ, 5, 3, COL>(o1, baddr);
      wino_wait4(e0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        D[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wino_pick(ac[2], s), e0[s], D[0], 0, 0, 0);
        D[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wino_pick(ac[2], s), o0[s], D[1], 0, 0, 0);
      }
      wino_wait4(e1);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        D[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wino_pick(ac[3], s), e1[s], D[0], 0, 0, 0);
        D[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wino_pick(ac[3], s), o1[s], D[1], 0, 0, 0);
      }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    // SLOTS is odd for K = 7 (9) and K = 3 has 4: the ping-pong of (a0, a1) is written out per K
    vslot(I0{}, I0{}, a0, a1, true); vslot(I0{}, I1{}, a1, a0, true); vslot(I0{}, I2{}, a0, a1, true);
    if constexpr (G == 1) {
      vslot(I0{}, I3{}, a1, a0, false);
    } else {
      vslot(I0{}, I3{}, a1, a0, true);
      vslot(I1{}, I0{}, a0, a1, true); vslot(I1{}, I1{}, a1, a0, true); vslot(I1{}, I2{}, a0, a1, true); vslot(I1{}, I3{}, a1, a0, true);
      if constexpr (G == 2) {
        dslot(I1{}, a0, a1, false);                                   // tap 3, pad 3: q offset 0 -> column 1
      } else {
        vslot(I2{}, I0{}, a0, a1, true); vslot(I2{}, I1{}, a1, a0, true); vslot(I2{}, I2{}, a0, a1, true); vslot(I2{}, I3{}, a1, a0, true);
        dslot(I0{}, a0, a1, true);                                    // tap 3, pad 5: q offset -1 -> column 0
        dslot(I2{}, a1, a0, false);                                   // tap 7: q offset +1 -> column 2
      }
    }
  };

  issue(0);
  for (int ch = 0; ch < p.nchunks; ++ch) {
    publish(ch);
    __syncthreads();                                       // raw complete; every wave has left the previous chunk's MFMA phase
    transform();
    if (ch + 1 < p.nchunks) issue(ch + 1);                 // in flight under the MFMA phase
    __syncthreads();
    if (row_ok) mfma_chunk(ch);
  }
  if (!row_ok) return;

  // ---- output transform + epilogue: lane owns y[row][2q], y[row][2q+1] for its 16 rows
  const int col = n0 + 2 * (wn * 32 + l31);
  if (col >= L) return;
  const float* bias = p.bias + mt * 32 + 4 * hi;
  const long long yrow0 = (long long)bz * p.y_bs + (long long)(mt * 32 + 4 * hi) * p.y_ld + col;
  float2 vo[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float bv = bias[(r & 3) + 8 * (r >> 2)];
    float ye = (M[0][r] + M[1][r]) + M[2][r];
    float yo = (M[1][r] - M[2][r]) - M[3][r];
    if constexpr (ND > 0) { ye += D[0][r]; yo += D[1][r]; }
    vo[r] = make_float2(ye + bv, yo + bv);
  }
  if (p.flags & F_RES) {
    const float* rb = p.res + (long long)bz * p.res_bs + (long long)(mt * 32 + 4 * hi) * p.res_ld + col;
    float2 rv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rv[r] = *reinterpret_cast<const float2*>(rb + (long long)((r & 3) + 8 * (r >> 2)) * p.res_ld);
#pragma unroll
    for (int r = 0; r < 16; ++r) { vo[r].x += rv[r].x; vo[r].y += rv[r].y; }
  }
  if (p.flags & F_ACC) {
    float2 yv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) yv[r] = *reinterpret_cast<const float2*>(p.y + yrow0 + (long long)((r & 3) + 8 * (r >> 2)) * p.y_ld);
#pragma unroll
    for (int r = 0; r < 16; ++r) { vo[r].x = yv[r].x + vo[r].x; vo[r].y = yv[r].y + vo[r].y; }
  }
  if (p.flags & F_DIV) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int r = 0; r < 16; ++r) { vo[r].x = vo[r].x / p.div; vo[r].y = vo[r].y / p.div; }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) *reinterpret_cast<float2*>(p.y + yrow0 + (long long)((r & 3) + 8 * (r >> 2)) * p.y_ld) = vo[r];
}

template <int K>
__global__ void __launch_bounds__(256, 2) conv_wino_kernel(const WinoArgs p) {
  const int lin = blockIdx.x;
  const int total = gridDim.x;
  const int tl = xcd_linear(lin, total, p.xcd);
  const int t = tl / p.ntn;
  const int bz = t / p.gy;
  wino_tile<K>(p, tl - t * p.ntn, t - bz * p.gy, bz);
}

// up to three problems (the MRF chains' convolutions of one step, k = 11 / 7 / 3) in one launch, longest first
__global__ void __launch_bounds__(256, 2) conv_wino_group_kernel(const WinoGroup g) {
  const int lin = blockIdx.x;
  int pi = 0;
  if (lin >= g.end[0]) pi = 1;
  if (lin >= g.end[1]) pi = 2;
  const int first = pi == 0 ? 0 : g.end[pi - 1];
  const WinoArgs& p = g.a[pi];
  const int tl = xcd_linear(lin - first, g.end[pi] - first, p.xcd);
  const int t = tl / p.ntn;
  const int bz = t / p.gy;
  const int k = g.k[pi];
  if (k == 11) wino_tile<11>(p, tl - t * p.ntn, t - bz * p.gy, bz);
  else if (k == 7) wino_tile<7>(p, tl - t * p.ntn, t - bz * p.gy, bz);
  else wino_tile<3>(p, tl - t * p.ntn, t - bz * p.gy, bz);
}

// ------------------------------------------------------------------ weight transform + packing
// wp[m-tile][chunk][slot][k-group][lane][4]: lane l of k-step 4*kg + s holds the slot's weight for
// row 32*mt + (l & 31), channel 32*chunk + 2*(4*kg + s) + (l >> 5).
__global__ void pack_wino_kernel(const float* __restrict__ src, const float* __restrict__ scale, float* __restrict__ wp, int Cin,
                                 int Cout, int K, int nchunks, int slots, long long total) {
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
    if (slot < 4 * G) {
      const int g = slot >> 2, pp = slot & 3;
      const float w0 = w[4 * g] * sc, w1 = w[4 * g + 1] * sc, w2 = w[4 * g + 2] * sc;
      val = pp == 0 ? w0 : (pp == 1 ? 0.5f * ((w0 + w1) + w2) : (pp == 2 ? 0.5f * ((w0 - w1) + w2) : w2));
    } else {
      val = w[4 * (slot - 4 * G) + 3] * sc;               // taps 3, 7
    }
  }
  wp[e] = val;
}
__global__ void wino_scale_kernel(const float* __restrict__ v, const float* __restrict__ g, float* __restrict__ scale, long long inner) {
  __shared__ float red[256];
  const int i = blockIdx.x;
  const float* p = v + (long long)i * inner;
  float s = 0.f;
  for (long long k = threadIdx.x; k < inner; k += blockDim.x) s += p[k] * p[k];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) scale[i] = g[i] / sqrtf(red[0]);
}
__global__ void wino_bias_kernel(const float* __restrict__ bias, float* __restrict__ bp, int Cout, int rowsP) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rowsP) bp[r] = (bias && r < Cout) ? bias[r] : 0.f;
}

bool wino_supported(int Cin, int Cout, int K, int dil) {
  static const bool on = !(getenv("SVOC_WINO") && atoi(getenv("SVOC_WINO")) == 0);
  return on && dil == 1 && (K == 3 || K == 7 || K == 11) && Cin >= 64 && (Cin % KC) == 0 && (Cout % 32) == 0;
}

int pack_wino(PackedWino& pw, int Cin, int Cout, int K, const float* w_or_v, const float* g, const float* bias, hipStream_t st) {
  if (!((K == 3 || K == 7 || K == 11) && Cin > 0 && Cout > 0)) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "pack_wino: unsupported shape");
  pw.Cin = Cin; pw.Cout = Cout; pw.K = K;
  pw.nchunks = (Cin + KC - 1) / KC;
  pw.mtiles = (Cout + 31) / 32;
  pw.slots = 4 * ((K + 1) / 4) + ((K + 1) / 4 - 1);
  pw.flops_per_col = 2.0 * Cin * Cout * K;               // ALGORITHMIC work of the convolution it computes
  const long long total = (long long)pw.mtiles * pw.nchunks * pw.slots * 4 * 256;
  SVOC_TRY(pw.wp.ensure((size_t)(total + 1024) * sizeof(float)));
  SVOC_HIP(hipMemsetAsync(pw.wp.f() + total, 0, 1024 * sizeof(float), st));   // the last slot's prefetch reads one slot past the end
  SVOC_TRY(pw.bias.ensure((size_t)pw.mtiles * 32 * sizeof(float)));
  DevBuf scale;
  if (g) {
    SVOC_TRY(scale.ensure((size_t)Cout * sizeof(float)));
    hipLaunchKernelGGL(wino_scale_kernel, dim3((unsigned)Cout), dim3(256), 0, st, w_or_v, g, scale.f(), (long long)Cin * K);
  }
  hipLaunchKernelGGL(pack_wino_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w_or_v, g ? scale.f() : nullptr,
                     pw.wp.f(), Cin, Cout, K, pw.nchunks, pw.slots, total);
  hipLaunchKernelGGL(wino_bias_kernel, dim3((pw.mtiles * 32 + 255) / 256), dim3(256), 0, st, bias, pw.bias.f(), Cout, pw.mtiles * 32);
  SVOC_HIP(hipGetLastError());
  SVOC_HIP(hipStreamSynchronize(st));                      // `scale` is freed on return
  return SVOC_OK;
}

int pack_wino_named(PackedWino& pw, int Cin, int Cout, int K, const TensorTable& tab, const std::string& prefix, hipStream_t st) {
  const svoc_tensor* w = tab.find(prefix + ".weight");
  const svoc_tensor* v = tab.find(prefix + ".weight_v");
  const svoc_tensor* g = tab.find(prefix + ".weight_g");
  const svoc_tensor* b = tab.find(prefix + ".bias");
  const svoc_tensor* src = w ? w : v;
  if (!src || (!w && !g)) SVOC_FAIL(SVOC_ERR_MISSING_TENSOR, "missing tensor %s.weight / .weight_v / .weight_g", prefix.c_str());
  if (src->ndim != 3 || src->shape[0] != Cout || src->shape[1] != Cin || src->shape[2] != K)
    SVOC_FAIL(SVOC_ERR_SHAPE, "tensor %s has the wrong shape for a %d->%d k=%d convolution", src->name, Cin, Cout, K);
  return pack_wino(pw, Cin, Cout, K, src->data, w ? nullptr : g->data, b ? b->data : nullptr, st);
}

// ------------------------------------------------------------------ launches
static bool wino_args(const PackedWino& pw, const ConvArgs& a, int B, WinoArgs& w) {
  // the decoder's plain epilogue only: out[0], flags within RES | ACC | DIV, full rows, 8-byte aligned even-length rows
  const EpiOut& o = a.out[0];
  if (a.mode != EPI_PLAIN || a.in_mask || a.mask || a.gadd || (o.flags & ~(unsigned)(F_RES | F_ACC | F_DIV)) || a.split_row < pw.mtiles * 32) return false;
  if (o.nrows < pw.Cout || a.Ncols != a.Lin || (a.Ncols & 1)) return false;
  if ((reinterpret_cast<uintptr_t>(a.x) & 15) || (a.x_ld & 3) || (a.x_bs & 3)) return false;
  if ((reinterpret_cast<uintptr_t>(o.y) & 7) || (o.y_ld & 1) || (o.y_bs & 1)) return false;
  if ((o.flags & F_RES) && ((reinterpret_cast<uintptr_t>(o.res) & 7) || (o.res_ld & 1) || (o.res_bs & 1))) return false;
  w.x = a.x; w.x_bs = a.x_bs; w.x_ld = a.x_ld; w.Cin = pw.Cin; w.L = a.Lin; w.pre_slope = a.pre_slope;
  w.wp = pw.wp.f(); w.bias = pw.bias.f(); w.nchunks = pw.nchunks; w.mtiles = pw.mtiles;
  w.y = o.y; w.y_bs = o.y_bs; w.y_ld = o.y_ld;
  w.res = o.res; w.res_bs = o.res_bs; w.res_ld = o.res_ld;
  w.flags = o.flags; w.div = o.div;
  w.ntn = (a.Ncols + 127) / 128; w.gy = (pw.mtiles + 1) / 2; w.xcd = xcd_mapping_enabled();
  (void)B;
  return true;
}
template <int K> static size_t wino_lds() { return (size_t)WinoGeo<K>::LDS_BYTES; }

// 1 = not eligible (caller uses the direct kernel)
int launch_conv_wino(const PackedWino& pw, const ConvArgs& a, int B, hipStream_t st, long long min_tiles) {
  WinoArgs w;
  if (B <= 0 || !wino_args(pw, a, B, w)) return 1;
  const long long total = (long long)w.ntn * w.gy * B;
  if (min_tiles < 0) min_tiles = 2LL * device_cu_count();
  if (total < min_tiles || total > 0x7fffffffLL) return 1;    // short inputs: the direct / K-split kernels
  const double flops = pw.flops_per_col * (double)B * (double)a.Ncols;
  stats_add_conv(flops);
  int prof_idx = -1;
  if (prof_enabled()) {
    char d[160];
    snprintf(d, sizeof(d), "wino  Ci%-4d Co%-4d k%-2d d1  N%-7d B%-3d", pw.Cin, pw.Cout, pw.K, a.Ncols, B);
    prof_idx = prof_begin(st, d, flops);
  }
  if (pw.K == 11) { auto k = conv_wino_kernel<11>; SVOC_TRY(ensure_max_dyn_lds((const void*)k)); hipLaunchKernelGGL(k, dim3((unsigned)total), dim3(256), wino_lds<11>(), st, w); }
  else if (pw.K == 7) { auto k = conv_wino_kernel<7>; SVOC_TRY(ensure_max_dyn_lds((const void*)k)); hipLaunchKernelGGL(k, dim3((unsigned)total), dim3(256), wino_lds<7>(), st, w); }
  else { auto k = conv_wino_kernel<3>; SVOC_TRY(ensure_max_dyn_lds((const void*)k)); hipLaunchKernelGGL(k, dim3((unsigned)total), dim3(256), wino_lds<3>(), st, w); }
  prof_end(st, prof_idx);
  SVOC_HIP(hipGetLastError());
  return SVOC_OK;
}

int launch_conv_wino_group(const PackedWino* const* pws, const ConvArgs* as, int n, int B, hipStream_t st) {
  if (n < 2 || n > 3 || B <= 0) return 1;
  WinoGroup g{};
  long long total = 0;
  double flops = 0;
  size_t lds = 0;
  for (int i = 0; i < n; ++i) {
    if (!wino_args(*pws[i], as[i], B, g.a[i])) return 1;
    total += (long long)g.a[i].ntn * g.a[i].gy * B;
    if (total > 0x7fffffffLL) return 1;
    g.end[i] = (int)total;
    g.k[i] = pws[i]->K;
    flops += pws[i]->flops_per_col * (double)B * (double)as[i].Ncols;
    lds = std::max(lds, pws[i]->K == 11 ? wino_lds<11>() : (pws[i]->K == 7 ? wino_lds<7>() : wino_lds<3>()));
  }
  if (total < 2LL * device_cu_count()) return 1;
  for (int i = n; i < 3; ++i) { g.end[i] = 0x7fffffff; g.k[i] = 3; }
  stats_add_conv(flops, n);
  int prof_idx = -1;
  if (prof_enabled()) {
    char d[160];
    snprintf(d, sizeof(d), "winoG Ci%-4d Co%-4d k%d/%d/%d N%-7d B%-3d", pws[0]->Cin, pws[0]->Cout, pws[0]->K, pws[1]->K, n > 2 ? pws[2]->K : 0, as[0].Ncols, B);
    prof_idx = prof_begin(st, d, flops);
  }
  auto kern = conv_wino_group_kernel;
  SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
  hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), lds, st, g);
  prof_end(st, prof_idx);
  SVOC_HIP(hipGetLastError());
  return SVOC_OK;
}

}  // namespace svoc
