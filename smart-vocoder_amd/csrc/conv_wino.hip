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

// Dilation D (1, 3, 5) is handled through the polyphase view: a convolution with dilation D is D interleaved undilated
// convolutions, so the pair of outputs that shares a group's four products is (n, n + D) instead of (n, n + 1).  A tile
// holds PU = 64 - 64 % D pairs u = qq * D + phase (outputs n0 + 2 qq D + phase and that + D: contiguous blocks of 2 D
// columns, tile width 2 PU = 128 / 126 / 120); group g then reads the V planes at column u + 2 g D and a direct tap
// with q offset delta reads E / O at u + (delta + 1) D - everything else is unchanged.
// WM = row tiles per workgroup (waves along the rows): 4 x 1 (128 rows x 32 pairs) where the convolution has at least
// four row tiles - the staged / transformed input is then shared by twice as many MFMAs - else 2 x 2 (64 rows x 64 pairs).
template <int K, int D, int WM>
struct WinoGeo {
  static constexpr int G = (K + 1) / 4;                   // three-tap groups at tap offsets 0, 4, 8
  static constexpr int ND = G - 1;                        // left-over single taps (3, 7)
  static constexpr int PADT = (K - 1) / 2;                // padding in taps (columns: PADT * D)
  static constexpr int SLOTS = 4 * G + ND;                // weight slots per 32-channel chunk (a direct tap feeds E and O)
  static constexpr int WN = 4 / WM;                       // waves along the pairs
  static constexpr int PU = ((32 * WN) / D) * D;          // output pairs per tile
  static constexpr int W = 2 * PU;                        // output columns per tile
  static constexpr int NUV = PU + 2 * (G - 1) * D;        // entries of a V plane row
  static constexpr int NUE = PU + 2 * D;                  // entries of an E / O plane row (origin one q block before the tile)
  static constexpr int PQV = (NUV + 3) & ~3;              // plane row strides (floats)
  static constexpr int PQE = (NUE + 3) & ~3;
  static constexpr int NPL = ND > 0 ? 6 : 4;              // planes: V0..V3 (+ E, O)
  static constexpr int XOFF = -((PADT * D + 3) & ~3);     // raw tile starts at n0 + XOFF (multiple of 4)
  static constexpr int VMAX = (2 * (PU / D - 1 + 2 * (G - 1)) - PADT + 3) * D + D - 1;   // last position a V window reads
  static constexpr int EMAX = ND > 0 ? 2 * PU + 2 * D - 1 : 0;                           // last position of the O plane
  static constexpr int RAW = (((VMAX > EMAX ? VMAX : EMAX) + 1 - XOFF) + 3) & ~3;        // raw tile columns
  static constexpr int RAW_FLOATS = KC * RAW;
  static constexpr int PL_FLOATS = 4 * KC * PQV + (ND > 0 ? 2 * KC * PQE : 0);
  static constexpr int LDS_BYTES = (RAW_FLOATS + PL_FLOATS) * 4;
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

// B fragments of one k-group (4 k-steps = channels 8*KG .. 8*KG+7) of a plane whose rows start BASE floats into the
// plane area, row stride PQ, at column offset COL: lane (l31, hi) reads row 8*KG + 2*s + hi.
// `baddr` = LDS byte address of the plane area + (hi * PQ + this lane's pair index u) * 4.
template <int PQ, int BASE, int KG, int COL>
__device__ __forceinline__ void wino_frag(float (&b)[4], unsigned baddr) {
  constexpr int O = BASE + 8 * KG * PQ + COL;
  b[0] = wino_lds_rd<(O) * 4>(baddr);
  b[1] = wino_lds_rd<(O + 2 * PQ) * 4>(baddr);
  b[2] = wino_lds_rd<(O + 4 * PQ) * 4>(baddr);
  b[3] = wino_lds_rd<(O + 6 * PQ) * 4>(baddr);
}

template <int T, int N, class F>
__device__ __forceinline__ void wino_static_for(F&& f) {
  if constexpr (T < N) {
    f(std::integral_constant<int, T>{});
    wino_static_for<T + 1, N>(f);
  }
}
// step t of a chunk (see mfma_chunk): direct-tap steps request 8 fragment values (E and O), group steps 4
template <int K> constexpr bool wino_step_direct(int t) { return t / 4 >= 4 * ((K + 1) / 4); }
template <int K> constexpr int wino_step_reads(int t) {
  return t >= 4 * (4 * ((K + 1) / 4) + (K + 1) / 4 - 1) ? 0 : (wino_step_direct<K>(t) ? 8 : 4);
}

template <int K, int D, int WM>
__device__ __forceinline__ void wino_tile(const WinoArgs& p, const int bx, const int by, const int bz) {
  using Geo = WinoGeo<K, D, WM>;
  constexpr int G = Geo::G, ND = Geo::ND, PADT = Geo::PADT, PQV = Geo::PQV, PQE = Geo::PQE, RAW = Geo::RAW, SLOTS = Geo::SLOTS;
  constexpr int PU = Geo::PU, XOFF = Geo::XOFF;
  extern __shared__ __attribute__((aligned(16))) float wl[];
  float* const raw = wl;                                   // [KC][RAW]  lrelu(x), zero outside [0, L)
  float* const pl = wl + Geo::RAW_FLOATS;                  // V0..V3 [KC][PQV] each, then E, O [KC][PQE]
  constexpr int EBASE = 4 * KC * PQV;                      // E plane offset inside the plane area (O follows)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = WM == 4 ? wave : wave >> 1, wn = WM == 4 ? 0 : wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;
  const int n0 = bx * Geo::W;                              // first output column of the workgroup
  const int mt = by * WM + wm;                             // this wave's 32-row tile
  const bool row_ok = mt < p.mtiles;
  const int mtc = row_ok ? mt : p.mtiles - 1;
  const int L = p.L;

  f32x16 M[4], Dd[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { M[0][i] = 0.f; M[1][i] = 0.f; M[2][i] = 0.f; M[3][i] = 0.f; Dd[0][i] = 0.f; Dd[1][i] = 0.f; }

  // ---- raw staging: 32 channels x RAW columns.  Thread (r0, g4) of the first RPP * R4 threads owns float4 group g4 of rows
  // r0, r0 + RPP, ...: one per-thread offset + a wave-uniform base per pass on the global side, one per-thread LDS address +
  // an immediate per pass on the LDS side - no per-slot index arithmetic (every vector instruction of these phases takes
  // matrix-pipe time from the other workgroup's MFMA phase).
  constexpr int R4 = RAW / 4, RPP = 256 / R4, NPASS = (KC + RPP - 1) / RPP;
  const char* const xb = reinterpret_cast<const char*>(p.x + (long long)bz * p.x_bs);
  const long long ldb = (long long)p.x_ld * 4;
  const int xs_start = n0 + XOFF;
  const float slope = p.pre_slope;
  const int r0 = tid / R4, g4 = tid - r0 * R4;
  const bool mine = tid < RPP * R4;
  const int tg = xs_start + 4 * g4;                        // first column of this thread's groups
  const bool inside = tg >= 0 && tg + 3 < L;               // else: a group that straddles an end of the sequence
  const unsigned goff = (unsigned)((mine ? r0 : 0) * p.x_ld + (inside ? tg : 0)) * 4u;
  float4 v[NPASS];
  auto issue = [&](int ch) {
    const char* cb = xb + (long long)ch * KC * ldb;
#pragma unroll
    for (int u = 0; u < NPASS; ++u) {
      if (u * RPP + RPP - 1 < KC) {                        // compile time: every row of this pass exists
        v[u] = *reinterpret_cast<const float4*>(cb + (long long)(u * RPP) * ldb + goff);
      } else {                                             // last, partial pass: rows beyond the chunk re-read its last row (not written)
        const int c = min((mine ? r0 : 0) + u * RPP, KC - 1);
        v[u] = *reinterpret_cast<const float4*>(cb + (long long)c * ldb + (long long)(inside ? tg : 0) * 4);
      }
    }
  };
  float* const rdst = raw + r0 * RAW + 4 * g4;
  auto publish = [&](int ch) {
#pragma unroll
    for (int u = 0; u < NPASS; ++u) {
      if (mine && (u * RPP + RPP - 1 < KC || r0 + u * RPP < KC)) {
        float4 q = v[u];
        if (!inside) {                                     // rare: element-wise with zero padding
          const float* row = reinterpret_cast<const float*>(xb + (long long)(ch * KC + r0 + u * RPP) * ldb);
          q.x = (tg >= 0 && tg < L) ? row[tg] : 0.f;
          q.y = (tg + 1 >= 0 && tg + 1 < L) ? row[tg + 1] : 0.f;
          q.z = (tg + 2 >= 0 && tg + 2 < L) ? row[tg + 2] : 0.f;
          q.w = (tg + 3 >= 0 && tg + 3 < L) ? row[tg + 3] : 0.f;
        }
        q.x = fmaxf(q.x, q.x * slope); q.y = fmaxf(q.y, q.y * slope);
        q.z = fmaxf(q.z, q.z * slope); q.w = fmaxf(q.w, q.w * slope);
        *reinterpret_cast<float4*>(rdst + u * RPP * RAW) = q;
      }
    }
  };
  // ---- transform pass: raw -> V0..V3 and E / O.  V entry u' = q' D + phase: window d_j = raw[(2 q' - PADT + j) D + phase - XOFF];
  // E / O entry e = (q + 1) D + phase: raw[2 q D + phase - XOFF], raw[(2 q + 1) D + phase - XOFF].  Same regular mapping:
  // an item (two q' of one phase: they share two window samples) per thread and pass, rows TPP apart.
  constexpr int NQ = Geo::NUV / D, NQ2 = (NQ + 1) / 2, IPR = NQ2 * D;   // q' per phase, pairs of q', items per row
  constexpr int TPP = 256 / IPR, TPASS = (KC + TPP - 1) / TPP;
  const int tr0 = tid / IPR, trem = tid - tr0 * IPR;
  const int tq2 = trem / D, tph = trem - tq2 * D;
  const bool tmine = tid < TPP * IPR;
  const bool tsecond = 2 * tq2 + 1 < NQ;                   // the item's second q' exists (NQ may be odd)
  const float* const tsrc = raw + tr0 * RAW + (4 * tq2 - PADT) * D + tph - XOFF;
  float* const tdst = pl + tr0 * PQV + 2 * tq2 * D + tph;
  constexpr int NQE = Geo::NUE / D, EIPR = ((NQE + 1) / 2) * D;         // E / O: pairs of q per phase
  constexpr int EPP = 256 / EIPR, EPASS = (KC + EPP - 1) / EPP;
  const int er0 = tid / EIPR, erem = tid - er0 * EIPR;
  const int eq2 = erem / D, eph = erem - eq2 * D;
  const bool emine = tid < EPP * EIPR;
  const bool esecond = 2 * eq2 + 1 < NQE;
  const float* const esrc = raw + er0 * RAW + (4 * eq2 - 2) * D + eph - XOFF;
  float* const edst = pl + EBASE + er0 * PQE + 2 * eq2 * D + eph;
  auto transform = [&]() {
#pragma unroll
    for (int u = 0; u < TPASS; ++u) {
      if (tmine && (u * TPP + TPP - 1 < KC || tr0 + u * TPP < KC)) {
        const float* r = tsrc + u * TPP * RAW;
        float* o = tdst + u * TPP * PQV;
        const float d0 = r[0], d1 = r[D], d2 = r[2 * D], d3 = r[3 * D];
        if constexpr (D == 1) {                            // the two q' are neighbours in the plane: 8-byte stores
          const float d4 = r[4], d5 = r[5];                // (NUV is even for D = 1)
          *reinterpret_cast<float2*>(o) = make_float2(d0 - d2, d2 - d4);
          *reinterpret_cast<float2*>(o + KC * PQV) = make_float2(d1 + d2, d3 + d4);
          *reinterpret_cast<float2*>(o + 2 * KC * PQV) = make_float2(d2 - d1, d4 - d3);
          *reinterpret_cast<float2*>(o + 3 * KC * PQV) = make_float2(d1 - d3, d3 - d5);
        } else {
          o[0] = d0 - d2; o[KC * PQV] = d1 + d2; o[2 * KC * PQV] = d2 - d1; o[3 * KC * PQV] = d1 - d3;
          if (tsecond) {
            const float d4 = r[4 * D], d5 = r[5 * D];
            o[D] = d2 - d4; o[KC * PQV + D] = d3 + d4; o[2 * KC * PQV + D] = d4 - d3; o[3 * KC * PQV + D] = d3 - d5;
          }
        }
      }
    }
    if constexpr (ND > 0) {
#pragma unroll
      for (int u = 0; u < EPASS; ++u) {
        if (emine && (u * EPP + EPP - 1 < KC || er0 + u * EPP < KC)) {
          const float* r = esrc + u * EPP * RAW;           // q = 2 eq2 - 1: E = r[0], O = r[D]; q + 1: r[2 D], r[3 D]
          float* o = edst + u * EPP * PQE;
          if constexpr (D == 1) {
            const float2 a = *reinterpret_cast<const float2*>(r), b = *reinterpret_cast<const float2*>(r + 2);
            *reinterpret_cast<float2*>(o) = make_float2(a.x, b.x);
            *reinterpret_cast<float2*>(o + KC * PQE) = make_float2(a.y, b.y);
          } else {
            o[0] = r[0]; o[KC * PQE] = r[D];
            if (esecond) { o[D] = r[2 * D]; o[KC * PQE + D] = r[3 * D]; }
          }
        }
      }
    }
  };

  // ---- MFMA phase of one chunk.  Weight slots (pack_wino order): sigma = 4 g + p for the groups, then the direct taps.
  // All four weight fragments (16 k-steps) of slot sigma + 1 are requested while slot sigma computes.
  const int uu = wn * 32 + l31;                            // this lane's pair index (>= PU: idle lane of a dilated tile)
  const unsigned pbase = (unsigned)(size_t)pl;
  const unsigned baddrV = pbase + (unsigned)(hi * PQV + uu) * 4u;
  const unsigned baddrE = pbase + (unsigned)(hi * PQE + uu) * 4u;
  const char* const wrow = reinterpret_cast<const char*>(p.wp) + (size_t)mtc * p.nchunks * SLOTS * 4096 + (size_t)lane * 16;
  // The chunk is a static list of NS = 4 * SLOTS steps; step t = (slot t / 4, k-group t % 4) issues 4 MFMAs (group slot:
  // plane P at column 2 g D into M[P]) or 8 (direct tap: E and O at column DQ * D into Dd[0], Dd[1]).  Fragment reads run
  // TWO steps ahead in two register sets: step t waits for its own reads only (lgkmcnt = size of step t + 1's request),
  // issues its MFMAs, then requests step t + 2 into the set it has just consumed - an LDS round trip is never exposed.
  auto mfma_chunk = [&](int ch) {
    constexpr int NS = 4 * SLOTS;
    const char* wa = wrow + (size_t)ch * SLOTS * 4096;
    float4 a[2][4];
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) a[0][kg] = *reinterpret_cast<const float4*>(wa + kg * 1024);
    float fb[2][4], fo[2][4];
    auto request = [&](auto tc) {
      constexpr int T = decltype(tc)::value;
      if constexpr (T < NS) {
        constexpr int SG = T / 4, KG = T % 4;
        if constexpr (SG < 4 * G) {
          constexpr int GG = SG / 4, P = SG % 4;
          wino_frag<PQV, P * KC * PQV, KG, 2 * GG * D>(fb[T & 1], baddrV);
        } else {
          constexpr int DI = SG - 4 * G;
          constexpr int DQ = (G == 2) ? 1 : (DI == 0 ? 0 : 2);              // 1 + (tap - PADT) / 2 for taps 3, 7
          wino_frag<PQE, EBASE, KG, DQ * D>(fb[T & 1], baddrE);
          wino_frag<PQE, EBASE + KC * PQE, KG, DQ * D>(fo[T & 1], baddrE);
        }
      }
    };
    auto wait_for = [&](auto tc) {
      constexpr int T = decltype(tc)::value;
      constexpr int N = wino_step_reads<K>(T + 1);                          // requests younger than step T's
      float(&b)[4] = fb[T & 1];
      if constexpr (N == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
      else if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
      else asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
      if constexpr (wino_step_direct<K>(T)) {
        float(&o)[4] = fo[T & 1];
        asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));   // same wait covers the O fragments
      }
    };
    auto step = [&](auto tc) {
      constexpr int T = decltype(tc)::value;
      constexpr int SG = T / 4, KG = T % 4;
      // next slot's weights a whole slot ahead (two sets of four registers).  Measured alternative: one float4 per step two
      // steps ahead in a ring of three - 40 registers fewer, three waves per SIMD - is 9 % SLOWER (k=11: 1101 -> 1205 us)
      if constexpr (KG == 0 && SG + 1 < SLOTS) {
        const char* wn_ = wa + (SG + 1) * 4096;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) a[(SG + 1) & 1][kg] = *reinterpret_cast<const float4*>(wn_ + kg * 1024);
      }
      wait_for(tc);
      const float4 av = a[SG & 1][KG];
      if constexpr (SG < 4 * G) {
        constexpr int P = SG % 4;
#pragma unroll
        for (int s = 0; s < 4; ++s) M[P] = __builtin_amdgcn_mfma_f32_32x32x2f32(wino_pick(av, s), fb[T & 1][s], M[P], 0, 0, 0);
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          Dd[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wino_pick(av, s), fb[T & 1][s], Dd[0], 0, 0, 0);
          Dd[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wino_pick(av, s), fo[T & 1][s], Dd[1], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      request(std::integral_constant<int, T + 2>{});
      __builtin_amdgcn_sched_barrier(0);
    };
    request(std::integral_constant<int, 0>{});
    request(std::integral_constant<int, 1>{});
    wino_static_for<0, NS>(step);
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
  if (!row_ok || uu >= PU) return;

  // ---- output transform + epilogue: lane owns y[row][ne] and y[row][ne + D] for its 16 rows
  const int ne = n0 + 2 * (uu / D) * D + (uu % D);
  if (ne >= L) return;
  const bool odd_ok = ne + D < L;
  const float* bias = p.bias + mt * 32 + 4 * hi;
  const long long yrow0 = (long long)bz * p.y_bs + (long long)(mt * 32 + 4 * hi) * p.y_ld + ne;
  float2 vo[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float bv = bias[(r & 3) + 8 * (r >> 2)];
    float ye = (M[0][r] + M[1][r]) + M[2][r];
    float yo = (M[1][r] - M[2][r]) - M[3][r];
    if constexpr (ND > 0) { ye += Dd[0][r]; yo += Dd[1][r]; }
    vo[r] = make_float2(ye + bv, yo + bv);
  }
  // D == 1 with an even L: the pair is one aligned 8-byte access; otherwise two 4-byte accesses D columns apart
  auto ld2 = [&](const float* q) -> float2 {
    if constexpr (D == 1) { if (odd_ok) return *reinterpret_cast<const float2*>(q); return make_float2(q[0], 0.f); }
    else return make_float2(q[0], odd_ok ? q[D] : 0.f);
  };
  if (p.flags & F_RES) {
    const float* rb = p.res + (long long)bz * p.res_bs + (long long)(mt * 32 + 4 * hi) * p.res_ld + ne;
    float2 rv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rv[r] = ld2(rb + (long long)((r & 3) + 8 * (r >> 2)) * p.res_ld);
#pragma unroll
    for (int r = 0; r < 16; ++r) { vo[r].x += rv[r].x; vo[r].y += rv[r].y; }
  }
  if (p.flags & F_ACC) {
    float2 yv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) yv[r] = ld2(p.y + yrow0 + (long long)((r & 3) + 8 * (r >> 2)) * p.y_ld);
#pragma unroll
    for (int r = 0; r < 16; ++r) { vo[r].x = yv[r].x + vo[r].x; vo[r].y = yv[r].y + vo[r].y; }
  }
  if (p.flags & F_DIV) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int r = 0; r < 16; ++r) { vo[r].x = vo[r].x / p.div; vo[r].y = vo[r].y / p.div; }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float* q = p.y + yrow0 + (long long)((r & 3) + 8 * (r >> 2)) * p.y_ld;
    if constexpr (D == 1) {
      if (odd_ok) *reinterpret_cast<float2*>(q) = vo[r]; else q[0] = vo[r].x;
    } else {
      q[0] = vo[r].x;
      if (odd_ok) q[D] = vo[r].y;
    }
  }
}

template <int K, int D, int WM>
__global__ void __launch_bounds__(256, 2) conv_wino_kernel(const WinoArgs p) {
  const int lin = blockIdx.x;
  const int tl = xcd_linear(lin, gridDim.x, p.xcd);
  const int t = tl / p.ntn;
  const int bz = t / p.gy;
  wino_tile<K, D, WM>(p, tl - t * p.ntn, t - bz * p.gy, bz);
}

// up to three problems of one dilation (the MRF chains' convolutions of one step, k = 11 / 7 / 3) in one launch, longest first
template <int D, int WM>
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
  if (k == 11) wino_tile<11, D, WM>(p, tl - t * p.ntn, t - bz * p.gy, bz);
  else if (k == 7) wino_tile<7, D, WM>(p, tl - t * p.ntn, t - bz * p.gy, bz);
  else wino_tile<3, D, WM>(p, tl - t * p.ntn, t - bz * p.gy, bz);
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
  return on && (dil == 1 || dil == 3 || dil == 5) && (K == 3 || K == 7 || K == 11) && Cin >= 64 && (Cin % KC) == 0 && (Cout % 32) == 0;
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
static int wino_wm(const PackedWino& pw) {
  static const int force = getenv("SVOC_WINO_WM") ? atoi(getenv("SVOC_WINO_WM")) : 0;
  if (force == 2 || force == 4) return force;
  return pw.mtiles >= 4 && pw.mtiles % 4 == 0 ? 4 : 2;
}
static int wino_tile_w(int D, int WM) { return 2 * (((32 * (4 / WM)) / D) * D); }
static bool wino_args(const PackedWino& pw, const ConvArgs& a, int B, int D, int WM, WinoArgs& w) {
  // the decoder's plain epilogue only: out[0], flags within RES | ACC | DIV, full rows, 8-byte aligned rows
  const EpiOut& o = a.out[0];
  if (a.mode != EPI_PLAIN || a.in_mask || a.mask || a.gadd || (o.flags & ~(unsigned)(F_RES | F_ACC | F_DIV)) || a.split_row < pw.mtiles * 32) return false;
  if (!(D == 1 || D == 3 || D == 5) || o.nrows < pw.Cout || a.Ncols != a.Lin || a.Ncols < 4) return false;
  if ((reinterpret_cast<uintptr_t>(a.x) & 15) || (a.x_ld & 3) || (a.x_bs & 3)) return false;
  if ((long long)KC * a.x_ld * 4 >= (1LL << 31)) return false;                   // 32-bit per-thread offsets within a chunk
  if ((reinterpret_cast<uintptr_t>(o.y) & 7) || (o.y_ld & 1) || (o.y_bs & 1)) return false;
  if ((o.flags & F_RES) && ((reinterpret_cast<uintptr_t>(o.res) & 7) || (o.res_ld & 1) || (o.res_bs & 1))) return false;
  w.x = a.x; w.x_bs = a.x_bs; w.x_ld = a.x_ld; w.Cin = pw.Cin; w.L = a.Lin; w.pre_slope = a.pre_slope;
  w.wp = pw.wp.f(); w.bias = pw.bias.f(); w.nchunks = pw.nchunks; w.mtiles = pw.mtiles;
  w.y = o.y; w.y_bs = o.y_bs; w.y_ld = o.y_ld;
  w.res = o.res; w.res_bs = o.res_bs; w.res_ld = o.res_ld;
  w.flags = o.flags; w.div = o.div;
  const int W = wino_tile_w(D, WM);
  w.ntn = (a.Ncols + W - 1) / W; w.gy = (pw.mtiles + WM - 1) / WM; w.xcd = xcd_mapping_enabled();
  (void)B;
  return true;
}
template <int K, int D, int WM>
static int wino_launch_one(const WinoArgs& w, long long total, hipStream_t st) {
  using Geo = WinoGeo<K, D, WM>;
  static_assert(Geo::LDS_BYTES <= 160 * 1024, "tile does not fit");
  auto kern = conv_wino_kernel<K, D, WM>;
  SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
  const size_t lds = (size_t)Geo::LDS_BYTES;
  hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), lds, st, w);
  return SVOC_OK;
}
template <int D, int WM>
static size_t wino_lds(int K) {
  return K == 11 ? (size_t)WinoGeo<11, D, WM>::LDS_BYTES : (K == 7 ? (size_t)WinoGeo<7, D, WM>::LDS_BYTES : (size_t)WinoGeo<3, D, WM>::LDS_BYTES);
}
template <int D, int WM>
static int wino_launch_group(const WinoGroup& g, long long total, size_t lds, hipStream_t st) {
  auto kern = conv_wino_group_kernel<D, WM>;
  SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
  hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), lds, st, g);
  return SVOC_OK;
}

// 1 = not eligible (caller uses the direct kernel)
int launch_conv_wino(const PackedWino& pw, const ConvArgs& a, int B, int dil, hipStream_t st, long long min_tiles) {
  WinoArgs w;
  const int WM = wino_wm(pw);
  if (B <= 0 || !wino_args(pw, a, B, dil, WM, w)) return 1;
  const long long total = (long long)w.ntn * w.gy * B;
  if (min_tiles < 0) min_tiles = 2LL * device_cu_count();
  if (total < min_tiles || total > 0x7fffffffLL) return 1;    // short inputs: the direct / K-split kernels
  const double flops = pw.flops_per_col * (double)B * (double)a.Ncols;
  stats_add_conv(flops);
  int prof_idx = -1;
  if (prof_enabled()) {
    char d[160];
    snprintf(d, sizeof(d), "wino  Ci%-4d Co%-4d k%-2d d%-2d N%-7d B%-3d %dx%d", pw.Cin, pw.Cout, pw.K, dil, a.Ncols, B, WM, 4 / WM);
    prof_idx = prof_begin(st, d, flops);
  }
  int rc = SVOC_OK;
#define SVOC_W(KK, DD) if (pw.K == KK && dil == DD) rc = WM == 4 ? wino_launch_one<KK, DD, 4>(w, total, st) : wino_launch_one<KK, DD, 2>(w, total, st);
  SVOC_W(3, 1) SVOC_W(7, 1) SVOC_W(11, 1) SVOC_W(3, 3) SVOC_W(7, 3) SVOC_W(11, 3) SVOC_W(3, 5) SVOC_W(7, 5) SVOC_W(11, 5)
#undef SVOC_W
  prof_end(st, prof_idx);
  if (rc != SVOC_OK) return rc;
  SVOC_HIP(hipGetLastError());
  return SVOC_OK;
}

int launch_conv_wino_group(const PackedWino* const* pws, const ConvArgs* as, int n, int B, int dil, hipStream_t st) {
  if (n < 2 || n > 3 || B <= 0) return 1;
  WinoGroup g{};
  long long total = 0;
  double flops = 0;
  size_t lds = 0;
  const int WM = wino_wm(*pws[0]);
  for (int i = 0; i < n; ++i) {
    if (wino_wm(*pws[i]) != WM || !wino_args(*pws[i], as[i], B, dil, WM, g.a[i])) return 1;
    total += (long long)g.a[i].ntn * g.a[i].gy * B;
    if (total > 0x7fffffffLL) return 1;
    g.end[i] = (int)total;
    g.k[i] = pws[i]->K;
    flops += pws[i]->flops_per_col * (double)B * (double)as[i].Ncols;
    const int K = pws[i]->K;
    size_t l = 0;
    if (WM == 4) l = dil == 1 ? wino_lds<1, 4>(K) : (dil == 3 ? wino_lds<3, 4>(K) : wino_lds<5, 4>(K));
    else l = dil == 1 ? wino_lds<1, 2>(K) : (dil == 3 ? wino_lds<3, 2>(K) : wino_lds<5, 2>(K));
    lds = std::max(lds, l);
  }
  if (total < 2LL * device_cu_count()) return 1;
  for (int i = n; i < 3; ++i) { g.end[i] = 0x7fffffff; g.k[i] = 3; }
  stats_add_conv(flops, n);
  int prof_idx = -1;
  if (prof_enabled()) {
    char d[160];
    snprintf(d, sizeof(d), "winoG Ci%-4d Co%-4d k%d/%d/%d d%d N%-7d B%-3d %dx%d", pws[0]->Cin, pws[0]->Cout, pws[0]->K, pws[1]->K, n > 2 ? pws[2]->K : 0, dil,
             as[0].Ncols, B, WM, 4 / WM);
    prof_idx = prof_begin(st, d, flops);
  }
  int rc = SVOC_OK;
  if (WM == 4) rc = dil == 1 ? wino_launch_group<1, 4>(g, total, lds, st) : (dil == 3 ? wino_launch_group<3, 4>(g, total, lds, st) : wino_launch_group<5, 4>(g, total, lds, st));
  else rc = dil == 1 ? wino_launch_group<1, 2>(g, total, lds, st) : (dil == 3 ? wino_launch_group<3, 2>(g, total, lds, st) : wino_launch_group<5, 2>(g, total, lds, st));
  prof_end(st, prof_idx);
  if (rc != SVOC_OK) return rc;
  SVOC_HIP(hipGetLastError());
  return SVOC_OK;
}

}  // namespace svoc
