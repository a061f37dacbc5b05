// Winograd F(2,3) implicit-GEMM convolution for the ResBlock convolutions of the decoder's C >= 64 stages
// (reference modules.py:190-207: convs1 with dilation 1 / 3 / 5 and convs2, k = 3 / 7 / 11).
//
// The fp32 matrix pipe is the roofline of this path (157.3 TFLOP/s, no TF32 on gfx950) and the direct kernels sit at
// 82 % of it, so the remaining lever is to issue fewer MFMAs.  A k-tap convolution is split into groups of three taps
// at tap offsets 0, 4, 8 (k = 3: one group; 7: two + tap 3; 11: three + taps 3 and 7).  Each group is a minimal
// F(2,3) filtering: two outputs y[2q], y[2q+1] from the four inputs d_j = x[2q - pad + 4g + j] with four products
// instead of six,
//     V0 = d0 - d2   V1 = d1 + d2   V2 = d2 - d1   V3' = d3 - d1         (input transform, additions only)
//     U0 = w0        U1 = (w0+w1+w2)/2   U2 = (w0-w1+w2)/2   U3 = w2    (weight transform, at load time)
//     y[2q] = M0 + M1 + M2     y[2q+1] = M1 - M2 + M3'     M_p = sum_c U_p[c] * V_p[c]
// Because the tap offsets of the groups are multiples of four and pad is odd, every group reads the SAME transformed
// planes V_p[c][q'] (q' = q + 2g), so all groups accumulate into one set of four transform-domain accumulators; the
// channel reduction M_p is the GEMM the MFMAs do.  The left-over taps (3, 7) are ordinary taps on the de-interleaved
// planes E[q] = x[2q], O[q] = x[2q+1] (their tap offset minus pad is even, so y[2q] reads only E and y[2q+1] only O);
// they accumulate into M0 (part of y[2q] only) and M3' (part of y[2q+1] only), the bias starts in M1.
// MFMAs per output and channel pair: k=3: 2 (direct 3), k=7: 5 (7), k=11: 8 (11).
// fp32 throughout; the additions of the transforms round once more than the direct form (measured <= 1e-6 relative).
//
// One workgroup = 4 waves = 128 rows x 32 pairs (C >= 128) or 64 rows x 64 pairs; a wave owns 32 rows x 32 pairs = four
// accumulator tiles.  Per 32-channel chunk: raw tile (leaky-relu'd, zero padded) -> LDS -> transform pass -> six planes
// in LDS -> 16 * (4 G + 2 ND) MFMAs per wave with fragment reads at immediate offsets.
#include "svoc_internal.h"
#include "wino_common.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace svoc {

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
  static constexpr int XOFF = -((PADT * D + 3) & ~3);     // raw tile starts at n0 + XOFF (multiple of 4)
  static constexpr int NQ = PU / D + 2 * (G - 1);         // windows q' per phase and row
  // D = 1: an item of the transform pass is a pair of windows (q'a, q'a + 1) whose six samples sit at f[1..6] of two aligned
  // float4 of the raw row; S shifts the pairing (items start at q' = -S) so that this holds for every k.
  static constexpr int S = (D == 1 && ((-PADT - XOFF) & 3) == 3) ? 1 : 0;
  static constexpr int NT = (NQ + S + 1) / 2;             // D = 1: items per row
  // planes V0, V1, V2, V3' (+ E, O), one geometry: entry (q' + S) * D + phase.  E / O hold d1 / d2 of the same window, i.e.
  // E[q], O[q] with q = q' - (PADT - 1) / 2.
  static constexpr int EO0 = S + (PADT - 1) / 2;          // E / O entry of q in the V geometry: (q + EO0) * D + phase
  static constexpr int NUV = D == 1 ? 2 * NT : (NQ + S) * D;
  static constexpr int PQV = D == 5 ? NUV : ((NUV + 3) & ~3);  // plane row stride (floats); D = 5: unpadded, so that three workgroups fit a CU (53.2 KB each)
  static constexpr int NPL = ND > 0 ? 6 : 4;
  static constexpr int VMAX = (2 * (NQ - 1) - PADT + 3) * D + D - 1;                     // last position a window reads (from n0)
  static constexpr int RAW0 = (VMAX + 1 - XOFF + 3) & ~3;
  static constexpr int RAW = D == 1 ? (RAW0 > 4 * NT + 4 ? RAW0 : 4 * NT + 4) : RAW0;    // raw tile columns
  static constexpr int RAW_FLOATS = KC * RAW;
  // E / O share the V geometry (same entry index, written by the same transform item) unless that costs the second
  // workgroup per CU (k = 11, D = 5, 64-pair tiles: 82 KB): then they are stored one q block lower with their own row
  // stride, and the transform skips the entries that fall outside.
  static constexpr bool ESHIFT = D > 1 && ND > 0 && (RAW_FLOATS + 6 * KC * PQV) * 4 > 80 * 1024;
  static constexpr int ESH = ESHIFT ? 1 : 0;
  static constexpr int EO = EO0 - ESH;
  static constexpr int DQMAX = G == 2 ? 1 : 2;
  static constexpr int NUE = ESHIFT ? PU + (DQMAX - 1 + EO) * D : NUV;
  static constexpr int PQE = ESHIFT ? ((NUE + 3) & ~3) : PQV;
  static constexpr int PL_FLOATS = 4 * KC * PQV + (ND > 0 ? 2 * KC * PQE : 0);
  static constexpr int LDS_BYTES = (RAW_FLOATS + PL_FLOATS) * 4;
};

// step t of a chunk (see mfma_chunk): direct-tap steps request 8 fragment values (E and O), group steps 4
template <int K> constexpr bool wino_step_direct(int t) { return t / 4 >= 4 * ((K + 1) / 4); }
template <int K> constexpr int wino_step_reads(int t) {
  return t >= 4 * (4 * ((K + 1) / 4) + (K + 1) / 4 - 1) ? 0 : (wino_step_direct<K>(t) ? 8 : 4);
}

// Persistent workgroups (one per CU slot walking a strided list of tiles, tile-independent set-up done once, next tile's
// first raw loads requested before the epilogue) were measured too: k = 7 / 11 -1..3 %, k = 3 +7 % (the longer live ranges
// cost the third workgroup per CU), 16 x 512 step 37.15 -> 37.5 ms - not kept.  Lesson kept: loop-invariant work of the rare
// edge-tile paths must not be hoistable (it cost 80 registers there).
// Instruction budget: measured with per-workgroup stamps (tools/wino_timeline.py), the SIMD time of a workgroup is
// 64 cycles per MFMA plus ~20 cycles per OTHER vector / memory instruction its waves issue outside the MFMA stream (the
// partner wave's MFMAs do not hide them) - so staging, transform and epilogue are written for instruction count: aligned
// 16-byte LDS reads, merged E / O writes, branch-free interior tiles, uniform row bases, bias and direct taps folded into
// the transform-domain accumulators.
template <int K, int D, int WM, bool DBG = false>
__device__ __forceinline__ void wino_tile(const WinoArgs& p, const int bx, const int by, const int bz, const int dbg_lin = 0) {
  using Geo = WinoGeo<K, D, WM>;
  constexpr int G = Geo::G, ND = Geo::ND, PADT = Geo::PADT, PQV = Geo::PQV, RAW = Geo::RAW, SLOTS = Geo::SLOTS;
  constexpr int PU = Geo::PU, XOFF = Geo::XOFF, S = Geo::S, EO = Geo::EO, NQ = Geo::NQ;
  extern __shared__ __attribute__((aligned(16))) float wl[];
  float* const raw = wl;                                   // [KC][RAW]  lrelu(x), zero outside [0, L)
  float* const pl = wl + Geo::RAW_FLOATS;                  // V0, V1, V2, V3' (, E, O): [KC][PQV] each
  constexpr int PLANE = KC * PQV;

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

  // transform-domain accumulators: y[even] = M0 + M1 + M2, y[odd] = M1 - M2 + M3' (M3' = -M3: its plane holds d3 - d1).
  // The bias starts in M1 (part of both outputs); a direct tap adds w * E to M0 (even outputs only) and w * O to M3'.
  f32x16 M[4];
  {
    const float* bias = p.bias + mtc * 32 + 4 * hi;
#pragma unroll
    for (int i = 0; i < 16; ++i) { M[0][i] = 0.f; M[1][i] = bias[(i & 3) + 8 * (i >> 2)]; M[2][i] = 0.f; M[3][i] = 0.f; }
  }

  // ---- raw staging: 32 channels x RAW columns = NG float4 groups.  Two item -> thread mappings:
  // LINEAR (4 x 1 tiles, few passes): group g = row * R4 + g4 goes to thread g % 256 in pass g / 256 - the last pass is partial
  // and whole waves skip it; the LDS tile is contiguous in g, the global side keeps one per-thread offset per pass.
  // Regular (2 x 2 tiles, five or six passes): thread (r0, g4) owns group g4 of rows r0, r0 + RPP, ...: one per-thread offset
  // and a uniform base / an immediate per pass, no per-pass registers (and no scalar masks to spill).
  constexpr bool LINEAR = WM == 4 && D < 5;                // D = 5: the per-pass registers would cost the third workgroup per CU
  constexpr int R4 = RAW / 4, NG = KC * R4, RPP = 256 / R4;
  constexpr int NPASS = LINEAR ? (NG + 255) / 256 : (KC + RPP - 1) / RPP;
  const char* const xb = reinterpret_cast<const char*>(p.x + (long long)bz * p.x_bs);
  const long long ldb = (long long)p.x_ld * 4;
  const int xs_start = n0 + XOFF;
  const bool interior = xs_start >= 0 && xs_start + RAW <= L;      // workgroup-uniform: no zero padding anywhere in the tile
  const float slope = p.pre_slope;
  const int r0 = tid / R4, g40 = tid - r0 * R4;
  auto st_row = [&](int u) { return LINEAR ? (tid + 256 * u) / R4 : r0 + u * RPP; };
  auto st_g4 = [&](int u) { return LINEAR ? (tid + 256 * u) % R4 : g40; };
  auto st_valid = [&](int u) -> bool {
    if constexpr (LINEAR) return 256 * (u + 1) <= NG || tid < NG - 256 * u;
    else return tid < RPP * R4 && (u * RPP + RPP - 1 < KC || r0 + u * RPP < KC);
  };
  unsigned goff[LINEAR ? NPASS : 1];                       // byte offset of the pass's group inside the chunk (clamped into the tile)
#pragma unroll
  for (int u = 0; u < (LINEAR ? NPASS : 1); ++u) {
    const int row = LINEAR ? min(st_row(u), KC - 1) : (tid < RPP * R4 ? r0 : 0), tg = xs_start + 4 * st_g4(u);
    goff[u] = (unsigned)(row * p.x_ld + ((tg >= 0 && tg + 3 < L) ? tg : 0)) * 4u;
  }
  float4 v[NPASS];
  auto issue = [&](int ch) {
    const char* cb = xb + (long long)ch * KC * ldb;
#pragma unroll
    for (int u = 0; u < NPASS; ++u) {
      if constexpr (LINEAR) v[u] = *reinterpret_cast<const float4*>(cb + goff[u]);
      else if (u * RPP + RPP - 1 < KC) v[u] = *reinterpret_cast<const float4*>(cb + (long long)(u * RPP) * ldb + goff[0]);
      else {                                               // last, partial pass: rows beyond the chunk re-read its last row (not written)
        const int c = min((tid < RPP * R4 ? r0 : 0) + u * RPP, KC - 1), tg = xs_start + 4 * g40;
        v[u] = *reinterpret_cast<const float4*>(cb + (long long)c * ldb + (long long)((tg >= 0 && tg + 3 < L) ? tg : 0) * 4);
      }
    }
  };
  float* const rdst = LINEAR ? raw + 4 * tid : raw + r0 * RAW + 4 * g40;
  constexpr int RSTEP = LINEAR ? 1024 : RPP * RAW;         // LDS floats between a thread's groups of consecutive passes
  auto publish = [&](int ch) {
    if (interior) {
#pragma unroll
      for (int u = 0; u < NPASS; ++u) {
        if (st_valid(u)) {
          float4 q = v[u];
          wino_lrelu4(q, slope);
          *reinterpret_cast<float4*>(rdst + RSTEP * u) = q;
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < NPASS; ++u) {
        if (st_valid(u)) {
          const int row = st_row(u), tg = xs_start + 4 * st_g4(u);
          float4 q = v[u];
          if (!(tg >= 0 && tg + 3 < L)) {                  // a group that straddles an end of the sequence: element-wise, zero padding
            const float* xr = reinterpret_cast<const float*>(xb + (long long)(ch * KC + row) * ldb);
            q.x = (tg >= 0 && tg < L) ? xr[tg] : 0.f;
            q.y = (tg + 1 >= 0 && tg + 1 < L) ? xr[tg + 1] : 0.f;
            q.z = (tg + 2 >= 0 && tg + 2 < L) ? xr[tg + 2] : 0.f;
            q.w = (tg + 3 >= 0 && tg + 3 < L) ? xr[tg + 3] : 0.f;
          }
          wino_lrelu4(q, slope);
          *reinterpret_cast<float4*>(rdst + RSTEP * u) = q;
        }
      }
    }
  };
  // ---- transform pass: raw -> planes.  Window q' of phase ph: d_j = raw[(2 q' - PADT + j) D + ph - XOFF], j = 0..3;
  // plane entry (q' + S) D + ph:  V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3' = d3 - d1, E = d1, O = d2.
  // An item is two consecutive windows of one phase (they share two samples), IPR items per row; same two mappings
  // (LINEAR: LDS addresses per pass computed once per workgroup; regular: rows TPP apart, immediates).
  constexpr int NQ2 = D == 1 ? Geo::NT : (NQ + 1) / 2, IPR = NQ2 * D;      // items per row
  constexpr int NI = KC * IPR, TPP = 256 / IPR;
  constexpr int TPASS = LINEAR ? (NI + 255) / 256 : (KC + TPP - 1) / TPP;
  constexpr int TN = LINEAR ? TPASS : 1;                   // per-pass address sets kept in registers
  constexpr int PQE = Geo::PQE, ESH = Geo::ESH;
  constexpr int EBASE = 4 * PLANE;                         // E plane offset inside the plane area (O follows, KC * PQE further)
  const float* tsrc[TN];                                   // LDS addresses of the item's samples / V0 entry / E entry
  float* tdst[TN];
  float* tedst[TN];
  unsigned tflags = 0;                                     // per set: bit 3u = second window exists, 3u+1 / 3u+2 = E/O of window 1 / 2 stored
  const int tr0 = tid / IPR;
#pragma unroll
  for (int u = 0; u < TN; ++u) {
    const int it = LINEAR ? min(tid + 256 * u, NI - 1) : tid;
    const int row = LINEAR ? it / IPR : tr0, rem = it - row * IPR;
    const int tq2 = rem / D, tph = rem - tq2 * D;
    // D = 1: item t covers q' = 2 t - S, 2 t - S + 1; its samples are f[1..6] of raw[4 t .. 4 t + 7] (see WinoGeo::S)
    tsrc[u] = D == 1 ? raw + row * RAW + 4 * tq2 : raw + row * RAW + (4 * tq2 - PADT) * D + tph - XOFF;
    tdst[u] = pl + row * PQV + 2 * tq2 * D + tph;
    const int e1 = (2 * tq2 - ESH) * D + tph;              // E / O entry of the first window (second: + D)
    tedst[u] = pl + EBASE + row * PQE + e1;
    const bool second = D == 1 || 2 * tq2 + 1 < NQ;
    if (second) tflags |= 1u << (3 * u);
    if (e1 >= 0 && e1 < Geo::NUE) tflags |= 2u << (3 * u);
    if (second && e1 + D < Geo::NUE) tflags |= 4u << (3 * u);
  }
  auto tr_valid = [&](int u) -> bool {
    if constexpr (LINEAR) return 256 * (u + 1) <= NI || tid < NI - 256 * u;
    else return tid < TPP * IPR && (u * TPP + TPP - 1 < KC || tr0 + u * TPP < KC);
  };
  auto transform = [&]() {
#pragma unroll
    for (int u = 0; u < TPASS; ++u) {
      if (tr_valid(u)) {
        constexpr int FS = LINEAR ? 3 : 0;                 // flag set of pass u: 3 u (LINEAR) or the one shared set
        const float* r = LINEAR ? tsrc[LINEAR ? u : 0] : tsrc[0] + u * TPP * RAW;
        float* o = LINEAR ? tdst[LINEAR ? u : 0] : tdst[0] + u * TPP * PQV;
        if constexpr (D == 1) {
          const float4 fa = *reinterpret_cast<const float4*>(r), fb = *reinterpret_cast<const float4*>(r + 4);
          const float d0 = fa.y, d1 = fa.z, d2 = fa.w, d3 = fb.x, d4 = fb.y, d5 = fb.z;
          *reinterpret_cast<float2*>(o) = make_float2(d0 - d2, d2 - d4);
          *reinterpret_cast<float2*>(o + PLANE) = make_float2(d1 + d2, d3 + d4);
          *reinterpret_cast<float2*>(o + 2 * PLANE) = make_float2(d2 - d1, d4 - d3);
          *reinterpret_cast<float2*>(o + 3 * PLANE) = make_float2(d3 - d1, d5 - d3);
          if constexpr (ND > 0) {
            *reinterpret_cast<float2*>(o + 4 * PLANE) = make_float2(d1, d3);
            *reinterpret_cast<float2*>(o + 5 * PLANE) = make_float2(d2, d4);
          }
        } else {
          const float d0 = r[0], d1 = r[D], d2 = r[2 * D], d3 = r[3 * D];
          o[0] = d0 - d2; o[PLANE] = d1 + d2; o[2 * PLANE] = d2 - d1; o[3 * PLANE] = d3 - d1;
          float* oe = LINEAR ? tedst[LINEAR ? u : 0] : tedst[0] + u * TPP * PQE;
          if constexpr (ND > 0) {
            if (!Geo::ESHIFT || (tflags & (2u << (FS * u)))) { oe[0] = d1; oe[KC * PQE] = d2; }
          }
          if ((NQ & 1) == 0 || (tflags & (1u << (FS * u)))) {
            const float d4 = r[4 * D], d5 = r[5 * D];
            o[D] = d2 - d4; o[PLANE + D] = d3 + d4; o[2 * PLANE + D] = d4 - d3; o[3 * PLANE + D] = d5 - d3;
            if constexpr (ND > 0) {
              if (!Geo::ESHIFT || (tflags & (4u << (FS * u)))) { oe[D] = d3; oe[KC * PQE + D] = d4; }
            }
          }
        }
      }
    }
  };

  // ---- MFMA phase of one chunk.  Weight slots (pack_wino order): sigma = 4 g + p for the groups, then the direct taps.
  const int uu = wn * 32 + l31;                            // this lane's pair index (>= PU: idle lane of a dilated tile)
  const unsigned pbase = (unsigned)(size_t)pl;
  const unsigned baddr = pbase + (unsigned)(hi * PQV + uu) * 4u;
  const unsigned baddrE = pbase + (unsigned)(hi * PQE + uu) * 4u;
  const char* const wrow = reinterpret_cast<const char*>(p.wp) + (size_t)mtc * p.nchunks * SLOTS * 4096 + (size_t)lane * 16;
  // The chunk is a static list of NS = 4 * SLOTS steps; step t = (slot t / 4, k-group t % 4) issues 4 MFMAs (group slot:
  // plane P at column (2 g + S) D into M[P]) or 8 (direct tap: E and O at column (DQ - 1 + EO) D into M[0], M[3]).  Fragment
  // reads run TWO steps ahead in two register sets: step t waits for its own reads only (lgkmcnt = size of step t + 1's
  // request), issues its MFMAs, then requests step t + 2 into the set it has just consumed.
  // Weight registers: two sets of four float4 (16 k-steps of one slot each); slot sigma of chunk ch lives in set
  // (PAR + sigma) & 1 where PAR is the chunk's starting set.  The slot-ahead request runs across chunk boundaries, and the
  // next chunk's raw-tile loads are issued before the transform pass (vmcnt retires in order: a weight wait in the MFMA
  // phase must not sit behind a fresh HBM request).  Both measured neutral, kept for the simpler wait pattern.
  float4 a[2][4];
  auto mfma_chunk = [&](int ch, auto par) {
    constexpr int PAR = decltype(par)::value;
    constexpr int NS = 4 * SLOTS;
    const char* wa = wrow + (size_t)ch * SLOTS * 4096;
    const bool more = ch + 1 < p.nchunks;
    float fb[2][4], fo[2][4];
    auto request = [&](auto tc) {
      constexpr int T = decltype(tc)::value;
      if constexpr (T < NS) {
        constexpr int SG = T / 4, KG = T % 4;
        if constexpr (SG < 4 * G) {
          constexpr int GG = SG / 4, P = SG % 4;
          wino_frag<PQV, P * PLANE, KG, (2 * GG + S) * D>(fb[T & 1], baddr);
        } else {
          constexpr int DI = SG - 4 * G;
          constexpr int DQ = (G == 2) ? 1 : (DI == 0 ? 0 : 2);              // 1 + (tap - PADT) / 2 for taps 3, 7
          wino_frag<PQE, EBASE, KG, (DQ - 1 + EO) * D>(fb[T & 1], baddrE);
          wino_frag<PQE, EBASE + KC * PQE, KG, (DQ - 1 + EO) * D>(fo[T & 1], baddrE);
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
      // next slot's weights a whole slot ahead.  Measured alternative: one float4 per step two steps ahead in a ring of
      // three - 40 registers fewer - is 9 % SLOWER (k=11: 1101 -> 1205 us)
      if constexpr (KG == 0) {
        if (SG + 1 < SLOTS || more) {                      // slot SLOTS of this chunk = slot 0 of the next one
          const char* wn_ = wa + (SG + 1) * 4096;
#pragma unroll
          for (int kg = 0; kg < 4; ++kg) a[(PAR + SG + 1) & 1][kg] = *reinterpret_cast<const float4*>(wn_ + kg * 1024);
        }
      }
      wait_for(tc);
      const float4 av = a[(PAR + SG) & 1][KG];
      if constexpr (SG < 4 * G) {
        constexpr int P = SG % 4;
#pragma unroll
        for (int s = 0; s < 4; ++s) M[P] = __builtin_amdgcn_mfma_f32_32x32x2f32(wino_pick(av, s), fb[T & 1][s], M[P], 0, 0, 0);
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          M[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wino_pick(av, s), fb[T & 1][s], M[0], 0, 0, 0);
          M[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(wino_pick(av, s), fo[T & 1][s], M[3], 0, 0, 0);
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

  // diagnostics (tools/wino_timeline.py): wall-clock start / end of the workgroup, shader cycles wave 0 spent inside MFMA
  // phases and in the parts of the staging / transform section of the chunk loop
  long long ts_start = 0, cyc_mfma = 0, cyc_prep = 0, cyc_pub = 0, cyc_b1 = 0, cyc_tr = 0;
  if constexpr (DBG) ts_start = (long long)wall_clock64();
#pragma unroll
  for (int kg = 0; kg < 4; ++kg) a[0][kg] = *reinterpret_cast<const float4*>(wrow + kg * 1024);
  issue(0);
  auto chunk = [&](int ch, auto par) {
    long long c0 = 0, c1 = 0, q0 = 0, q1 = 0, q2 = 0;
    if constexpr (DBG) c0 = (long long)__builtin_readcyclecounter();
    publish(ch);
    if (ch + 1 < p.nchunks) issue(ch + 1);                 // in flight under the transform pass and the MFMA phase
    if constexpr (DBG) q0 = (long long)__builtin_readcyclecounter();
    __syncthreads();                                       // raw complete; every wave has left the previous chunk's MFMA phase
    if constexpr (DBG) q1 = (long long)__builtin_readcyclecounter();
    transform();
    if constexpr (DBG) { q2 = (long long)__builtin_readcyclecounter(); cyc_pub += q0 - c0; cyc_b1 += q1 - q0; cyc_tr += q2 - q1; }
    __syncthreads();
    if constexpr (DBG) c1 = (long long)__builtin_readcyclecounter();
    if (row_ok) mfma_chunk(ch, par);
    if constexpr (DBG) { cyc_prep += c1 - c0; cyc_mfma += (long long)__builtin_readcyclecounter() - c1; }
  };
  if constexpr ((SLOTS & 1) == 0) {
    for (int ch = 0; ch < p.nchunks; ++ch) chunk(ch, std::integral_constant<int, 0>{});
  } else {                                                 // odd slot count: the starting register set alternates
    for (int ch = 0; ch < p.nchunks; ch += 2) {
      chunk(ch, std::integral_constant<int, 0>{});
      if (ch + 1 < p.nchunks) chunk(ch + 1, std::integral_constant<int, 1>{});
    }
  }
  long long c_epi = 0;
  if constexpr (DBG) c_epi = (long long)__builtin_readcyclecounter();
  auto dbg_out = [&]() {
    if constexpr (DBG) if (tid == 0) {
      long long* d = p.dbg + 16 * (long long)(p.dbg_base + dbg_lin);
      d[8] = cyc_pub; d[9] = cyc_b1; d[10] = cyc_tr;
      d[0] = ts_start; d[1] = (long long)wall_clock64(); d[2] = cyc_mfma; d[3] = cyc_prep;
      d[4] = (long long)__builtin_readcyclecounter() - c_epi; d[5] = 1;
      d[6] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_ID: wave, simd, cu, sh, se
      d[7] = (long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);      // XCC_ID
    }
  };

  // ---- output transform + epilogue: lane owns y[row][ne] and y[row][ne + D] for its 16 rows.  Row r of the lane is at
  // (uniform base of row (r & 3) + 8 (r >> 2)) + (per-lane byte offset): no per-row vector address arithmetic.
  const int ne = n0 + 2 * (uu / D) * D + (uu % D);
  if (!row_ok || uu >= PU || ne >= L) { dbg_out(); return; }
  const bool odd_ok = ne + D < L;
  float2 vo[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float t1 = M[1][r] + M[2][r], t2 = M[1][r] - M[2][r];
    vo[r] = make_float2(M[0][r] + t1, M[3][r] + t2);
  }
  // address of row r = (uniform base of row block r >> 2) + (per-lane offset of row r & 3): 4 + 4 registers, no per-row math
  const unsigned ylb = (unsigned)p.y_ld * 4u, rlb = (unsigned)p.res_ld * 4u;
  unsigned yo4[4], ro4[4];
  char* yb4[4];
  const char* rb4[4];
  {
    const unsigned yoff = (unsigned)(4 * hi * p.y_ld + ne) * 4u, roff = (unsigned)(4 * hi * p.res_ld + ne) * 4u;
    char* const ybase = reinterpret_cast<char*>(p.y + (long long)bz * p.y_bs + (long long)(mt * 32) * p.y_ld);
    const char* const rbase = reinterpret_cast<const char*>(p.res + (long long)bz * p.res_bs + (long long)(mt * 32) * p.res_ld);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      yo4[i] = yoff + i * ylb; ro4[i] = roff + i * rlb;
      yb4[i] = ybase + (size_t)(8 * i) * ylb; rb4[i] = rbase + (size_t)(8 * i) * rlb;
    }
  }
  auto finish = [&](auto pair_c) {
    constexpr bool PAIR = decltype(pair_c)::value;         // the lane's second output exists
    auto ld2 = [&](const char* q) -> float2 {
      if constexpr (D == 1 && PAIR) return *reinterpret_cast<const float2*>(q);
      else if constexpr (PAIR) return make_float2(*reinterpret_cast<const float*>(q), *reinterpret_cast<const float*>(q + 4 * D));
      else return make_float2(*reinterpret_cast<const float*>(q), 0.f);
    };
    if (p.flags & F_RES) {
      float2 rv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) rv[r] = ld2(rb4[r >> 2] + ro4[r & 3]);
#pragma unroll
      for (int r = 0; r < 16; ++r) { vo[r].x += rv[r].x; vo[r].y += rv[r].y; }
    }
    if (p.flags & F_ACC) {
      float2 yv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) yv[r] = ld2(yb4[r >> 2] + yo4[r & 3]);
#pragma unroll
      for (int r = 0; r < 16; ++r) { vo[r].x = yv[r].x + vo[r].x; vo[r].y = yv[r].y + vo[r].y; }
    }
    if (p.flags & F_DIV) {                                 // x / div as x * (1 / div) with one residual correction: 3 instructions
      const float dv = p.div, rc = 1.0f / dv;              // per value instead of the ~10 of the IEEE sequence (same result up to
#pragma unroll                                             // rare last-bit differences in halfway cases)
      for (int r = 0; r < 16; ++r) {
        const float qx = vo[r].x * rc, qy = vo[r].y * rc;
        vo[r].x = __builtin_fmaf(__builtin_fmaf(-qx, dv, vo[r].x), rc, qx);
        vo[r].y = __builtin_fmaf(__builtin_fmaf(-qy, dv, vo[r].y), rc, qy);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      char* q = yb4[r >> 2] + yo4[r & 3];
      if constexpr (D == 1 && PAIR) *reinterpret_cast<float2*>(q) = vo[r];
      else {
        *reinterpret_cast<float*>(q) = vo[r].x;
        if constexpr (PAIR) *reinterpret_cast<float*>(q + 4 * D) = vo[r].y;
      }
    }
  };
  if (odd_ok) finish(std::true_type{}); else finish(std::false_type{});
  dbg_out();
}

template <int K, int D, int WM, bool DBG>
__global__ void __launch_bounds__(256, 2) conv_wino_kernel(const WinoArgs p) {
  const int lin = blockIdx.x;
  const int tl = xcd_linear(lin, gridDim.x, p.xcd);
  const int t = tl / p.ntn;
  const int bz = t / p.gy;
  wino_tile<K, D, WM, DBG>(p, tl - t * p.ntn, t - bz * p.gy, bz, lin);
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
  if (k == 11) wino_tile<11, D, WM>(p, tl - t * p.ntn, t - bz * p.gy, bz, lin);
  else if (k == 7) wino_tile<7, D, WM>(p, tl - t * p.ntn, t - bz * p.gy, bz, lin);
  else wino_tile<3, D, WM>(p, tl - t * p.ntn, t - bz * p.gy, bz, lin);
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

// conv_wino4.hip: the F(4,3) form
bool wino4_enabled();
bool wino4_c32_enabled();
bool wino44_enabled();
int wino4_slots(int K, bool f44);
int wino4_ntn(int L, int D, int NRT);
int pack_wino4_image(float* wp4, int Cin, int Cout, int K, bool f44, const float* w_or_v, const float* scale, hipStream_t st);
int pack_wino4_raw(float* dst, int Cin, int Cout, int K, const float* w_or_v, const float* scale, hipStream_t st);
int wino4_tail_plan(int L, int D, int NRT, int* w_first, int* nwin);
int wino4_launch(const WinoArgs& w, int K, int D, int NC, bool f44, long long total, hipStream_t st);
int wino4_launch_group(const WinoGroup& g, int D, int NC, int in_perm, int out_perm, bool f44, long long total, hipStream_t st, const float* const* wraw,
                       const int* Cout, int tail_w0, int tail_nw, int B);
int wino4_launch_acc3(const WinoArgs* a, int NRT, int in_perm, bool f44, long long total, hipStream_t st);      // conv_wino4_acc.hip

bool wino_supported(int Cin, int Cout, int K, int dil) {
  static const bool on = !(getenv("SVOC_WINO") && atoi(getenv("SVOC_WINO")) == 0);
  // Cin = Cout = 32 (round 4): only the F(4,3) form with one row tile per workgroup (conv_wino4.hip, NRT = 1)
  if (Cin == 32 && Cout == 32) return on && wino4_c32_enabled() && (dil == 1 || dil == 3 || dil == 5) && (K == 3 || K == 7 || K == 11);
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
  // F(4,3) image: 64- or 128-row blocks with even chunk counts, or the single 32 x 32 block of the last MRF stage
  if (wino4_enabled() && ((pw.mtiles % 2 == 0 && (pw.nchunks & 1) == 0) || (pw.mtiles == 1 && pw.nchunks == 1))) {
    // k = 7 / 11 in F(4,4) form (conv_wino4.h); k = 3 keeps F(4,3) and gets a second image for the merged accumulate launch
    pw.f44 = wino44_enabled() && K >= 7;
    const long long total4 = (long long)pw.mtiles * pw.nchunks * wino4_slots(K, pw.f44) * 4 * 256;
    SVOC_TRY(pw.wp4.ensure((size_t)(total4 + 1024) * sizeof(float)));
    SVOC_HIP(hipMemsetAsync(pw.wp4.f() + total4, 0, 1024 * sizeof(float), st));
    SVOC_TRY(pack_wino4_image(pw.wp4.f(), Cin, Cout, K, pw.f44, w_or_v, g ? scale.f() : nullptr, st));
    if (Cin % 32 == 0 && Cout % 32 == 0) {                  // (the tail kernel's 16-byte weight loads and four rows per pass)
      SVOC_TRY(pw.wraw.ensure((size_t)Cout * Cin * K * sizeof(float)));
      SVOC_TRY(pack_wino4_raw(pw.wraw.f(), Cin, Cout, K, w_or_v, g ? scale.f() : nullptr, st));
    }
    if (wino44_enabled() && K == 3) {
      const long long total44 = (long long)pw.mtiles * pw.nchunks * wino4_slots(K, true) * 4 * 256;
      SVOC_TRY(pw.wp44.ensure((size_t)(total44 + 1024) * sizeof(float)));
      SVOC_HIP(hipMemsetAsync(pw.wp44.f() + total44, 0, 1024 * sizeof(float), st));
      SVOC_TRY(pack_wino4_image(pw.wp44.f(), Cin, Cout, K, true, w_or_v, g ? scale.f() : nullptr, st));
    }
  }
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
  w.dbg = debug_stamp_buffer(); w.dbg_base = 0;
  static const unsigned abl_env = getenv("SVOC_DBG_ABL") ? (unsigned)atoi(getenv("SVOC_DBG_ABL")) : 0u;      // diagnostics (tools/wino4_timeline.py): results are WRONG with it
  w.abl = w.dbg ? abl_env : 0u;
  w.out_perm = 0;
  (void)B;
  return true;
}
template <int K, int D, int WM>
static int wino_launch_one(const WinoArgs& w, long long total, hipStream_t st) {
  using Geo = WinoGeo<K, D, WM>;
  static_assert(Geo::LDS_BYTES <= 160 * 1024, "tile does not fit");
  const size_t lds = (size_t)Geo::LDS_BYTES;
  if (w.dbg) {                                             // stamped build of the same kernel (tools/wino_timeline.py)
    auto kern = conv_wino_kernel<K, D, WM, true>;
    SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), lds, st, w);
  } else {
    auto kern = conv_wino_kernel<K, D, WM, false>;
    SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), lds, st, w);
  }
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

// share of the direct form's multiply-adds that the F(2,3) grouping issues: two products per output and three-tap group plus
// one per left-over tap (k=3: 2/3, k=7: 5/7, k=11: 8/11); tile padding is not counted
static double wino_exec_ratio(int K) { const int G = (K + 1) / 4; return (2.0 * G + (G - 1)) / (double)K; }

// F(4,3) form (conv_wino4.hip): four-row-tile blocks; dilation 1 additionally needs 16-byte aligned rows of a length that is a multiple of four
// (F(4,4) form, conv_wino4.h: seven products per four outputs and four-tap group, no left-over taps: k = 7: 3.5/7, k = 11: 5.25/11)
static double wino4_exec_ratio(int K, bool f44) { const int G = (K + 1) / 4; return (f44 ? 1.75 * G : 1.5 * G + (G - 1)) / (double)K; }
static int wino4_nc(const PackedWino& pw) { return pw.mtiles % 4 == 0 ? 4 : (pw.mtiles == 1 ? 1 : 2); }      // row tiles per workgroup: 128-, 64- or 32-row blocks
static bool wino4_args(const PackedWino& pw, const ConvArgs& a, int dil, const WinoArgs& w, WinoArgs& w4) {
  if (!wino4_enabled() || !pw.wp4.p || !(dil == 1 || dil == 3 || dil == 5) || (dil == 1 && (a.Ncols & 3))) return false;
  if ((long long)pw.Cin * a.x_ld * 4 >= (1LL << 31)) return false;                 // the producers' staging loads: 32-bit offsets within one batch element
  const EpiOut& o = a.out[0];
  if (dil == 1) {                                           // contiguous outputs: 16-byte stores
    if ((reinterpret_cast<uintptr_t>(o.y) & 15) || (o.y_ld & 3) || (o.y_bs & 3)) return false;
    if ((o.flags & F_RES) && ((reinterpret_cast<uintptr_t>(o.res) & 15) || (o.res_ld & 3) || (o.res_bs & 3))) return false;
  }
  // window-major rows: written by a dilated convolution with a plain epilogue, read by an undilated one; the row must hold whole q blocks
  if (a.wperm_out && (dil == 1 || a.wperm_out != dil || (o.flags & (F_RES | F_ACC | F_DIV)) || o.y_ld < 4 * dil * ((a.Ncols + 4 * dil - 1) / (4 * dil)))) return false;
  if (a.wperm_in && (dil != 1 || !(a.wperm_in == 3 || a.wperm_in == 5) || a.x_ld < 4 * a.wperm_in * ((a.Ncols + 4 * a.wperm_in - 1) / (4 * a.wperm_in)))) return false;
  // both sides of the window-major hand-over move whole float4 groups per lane (conv_wino4_kernels.h: the dilated epilogue's 16-byte store, the
  // undilated producers' 16-byte q-block loads): base, row stride and batch stride must keep them aligned
  if (a.wperm_out && ((reinterpret_cast<uintptr_t>(o.y) & 15) || (o.y_ld & 3) || (o.y_bs & 3))) return false;
  if (a.wperm_in && ((reinterpret_cast<uintptr_t>(a.x) & 15) || (a.x_ld & 3) || (a.x_bs & 3))) return false;
  w4 = w;
  w4.out_perm = a.wperm_out;
  w4.wp = pw.wp4.f();
  w4.ntn = wino4_ntn(a.Ncols, dil, wino4_nc(pw));
  w4.gy = pw.mtiles / wino4_nc(pw);
  return true;
}

// 1 = not eligible (caller uses the direct kernel)
int launch_conv_wino(const PackedWino& pw, const ConvArgs& a, int B, int dil, hipStream_t st, long long min_tiles) {
  WinoArgs w;
  const int WM = wino_wm(pw);
  if (a.wperm_in || a.wperm_out) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "window-major rows exist in the grouped F(4,3) launches only");
  if (B <= 0 || !wino_args(pw, a, B, dil, WM, w)) return 1;
  const long long total = (long long)w.ntn * w.gy * B;
  if (min_tiles < 0) min_tiles = 2LL * device_cu_count();
  if ((long long)w.ntn * w.gy * variant_batch(B) < min_tiles || total > 0x7fffffffLL) return 1;    // short inputs: the direct / K-split kernels
  const double flops = pw.flops_per_col * (double)B * (double)a.Ncols;
  WinoArgs w4;
  const bool f4 = wino4_args(pw, a, dil, w, w4);
  if (!f4 && pw.mtiles < 2) return 1;                       // C = 32 exists in F(4,3) form only
  stats_add_conv(flops, 1, flops * (f4 ? wino4_exec_ratio(pw.K, pw.f44) : wino_exec_ratio(pw.K)));
  int prof_idx = -1;
  if (prof_enabled()) {
    char d[160];
    snprintf(d, sizeof(d), "%s Ci%-4d Co%-4d k%-2d d%-2d N%-7d B%-3d %dx%d%s", f4 ? "wino4" : "wino ", pw.Cin, pw.Cout, pw.K, dil, a.Ncols, B, WM, 4 / WM,
             f4 && pw.f44 ? " F(4,4)" : "");
    prof_idx = prof_begin(st, d, flops);
  }
  int rc = SVOC_OK;
  if (f4) {
    rc = wino4_launch(w4, pw.K, dil, wino4_nc(pw), pw.f44, (long long)w4.ntn * w4.gy * B, st);
    prof_end(st, prof_idx);
    if (rc != SVOC_OK) return rc < 0 ? rc : SVOC_ERR_UNSUPPORTED;
    SVOC_HIP(hipGetLastError());
    return SVOC_OK;
  }
#define SVOC_W(KK, DD) if (pw.K == KK && dil == DD) rc = WM == 4 ? wino_launch_one<KK, DD, 4>(w, total, st) : wino_launch_one<KK, DD, 2>(w, total, st);
  SVOC_W(3, 1) SVOC_W(7, 1) SVOC_W(11, 1) SVOC_W(3, 3) SVOC_W(7, 3) SVOC_W(11, 3) SVOC_W(3, 5) SVOC_W(7, 5) SVOC_W(11, 5)
#undef SVOC_W
  prof_end(st, prof_idx);
  if (rc != SVOC_OK) return rc;
  SVOC_HIP(hipGetLastError());
  return SVOC_OK;
}

// The three chains' last convolutions (k = 3, 7, 11 in chain order, dilation 1), each with its own epilogue flags (residual;
// + accumulate; + accumulate and divide) into one output: ONE launch of conv_wino4_accum_kernel.  1 = not eligible (the caller
// runs them one by one).
int launch_conv_wino4_accum(const PackedWino* const* pws, const ConvArgs* as, int B, hipStream_t st, bool query) {
  if (B <= 0 || !pws[0] || !pws[1] || !pws[2] || pws[0]->K != 3 || pws[1]->K != 7 || pws[2]->K != 11) return 1;
  WinoGroup g4{};
  double flops = 0, exec4 = 0;
  long long total = 0;
  const int WM = wino_wm(*pws[0]);
  for (int i = 0; i < 3; ++i) {
    WinoArgs w;
    if (wino_wm(*pws[i]) != WM || wino4_nc(*pws[i]) != wino4_nc(*pws[0]) || !wino_args(*pws[i], as[i], B, 1, WM, w)) return 1;
    if (!wino4_args(*pws[i], as[i], 1, w, g4.a[i])) return 1;
    const long long t = (long long)g4.a[i].ntn * g4.a[i].gy * B;
    if (i == 0) total = t;
    if (t != total || total > 0x7fffffffLL || as[i].out[0].y != as[0].out[0].y) return 1;      // one tile space, one output tensor
    g4.end[i] = (int)total; g4.k[i] = pws[i]->K;
    flops += pws[i]->flops_per_col * (double)B * (double)as[i].Ncols;
  }
  // F(4,4) members (128-row layout): k = 7 / 11 carry that image; the merged launch needs k = 3 in the same form (its second image)
  const bool f44 = pws[1]->f44 && pws[2]->f44;
  if (pws[1]->f44 != pws[2]->f44 || pws[0]->f44 || !f44) return 1;
  if (total * 3 / B * variant_batch(B) < mrf_min_tiles()) return 1;
  const int in_perm = as[0].wperm_in;
  if (as[1].wperm_in != in_perm || as[2].wperm_in != in_perm) return 1;
  // one set of accumulators for the three members (conv_wino4_acc.hip): every member a plain residual convolution, accumulated in chain order
  // (round 3's three read-modify-write members in one launch were removed in round 5: otherwise the caller runs them one by one)
  if (!((g4.a[0].flags & (F_RES | F_ACC | F_DIV)) == F_RES && (!f44 || pws[0]->wp44.p) &&
        (g4.a[1].flags & (F_RES | F_ACC)) == (F_RES | F_ACC) && (g4.a[2].flags & (F_RES | F_ACC)) == (F_RES | F_ACC))) return 1;
  if (query) return 0;
  int prof_idx = -1;
  if (prof_enabled()) {
    char d[160];
    snprintf(d, sizeof(d), "wino4A Ci%-4d Co%-4d k3+7+11 accumulate N%-7d B%-3d%s", pws[0]->Cin, pws[0]->Cout, as[0].Ncols, B, f44 ? " F(4,4)" : "");
    prof_idx = prof_begin(st, d, flops);
  }
  if (f44) g4.a[0].wp = pws[0]->wp44.f();                   // one set of accumulators: the k = 3 member in F(4,4) form as well
  for (int i = 0; i < 3; ++i)
    exec4 += pws[i]->flops_per_col * (double)B * (double)as[i].Ncols * wino4_exec_ratio(pws[i]->K, f44);
  stats_add_conv(flops, 3, exec4);
  const int rc = wino4_launch_acc3(g4.a, wino4_nc(*pws[0]), in_perm, f44, total, st);
  prof_end(st, prof_idx);
  if (rc != SVOC_OK) return rc < 0 ? rc : SVOC_ERR_UNSUPPORTED;
  SVOC_HIP(hipGetLastError());
  return SVOC_OK;
}

int launch_conv_wino_group(const PackedWino* const* pws, const ConvArgs* as, int n, int B, int dil, hipStream_t st, bool query) {
  if (n < 2 || n > 3 || B <= 0) return 1;
  WinoGroup g{}, g4{};
  long long total = 0, total4 = 0;
  bool f4 = n == 3 && pws[0]->K == 11 && pws[1]->K == 7 && pws[2]->K == 3;
  double flops = 0, exec_flops = 0, exec4 = 0;
  size_t lds = 0;
  const int WM = wino_wm(*pws[0]);
  for (int i = 0; i < n; ++i) {
    if (wino_wm(*pws[i]) != WM || !wino_args(*pws[i], as[i], B, dil, WM, g.a[i])) return 1;
    total += (long long)g.a[i].ntn * g.a[i].gy * B;
    if (total > 0x7fffffffLL) return 1;
    g.end[i] = (int)total;
    g.k[i] = pws[i]->K;
    flops += pws[i]->flops_per_col * (double)B * (double)as[i].Ncols;
    exec_flops += pws[i]->flops_per_col * (double)B * (double)as[i].Ncols * wino_exec_ratio(pws[i]->K);
    exec4 += pws[i]->flops_per_col * (double)B * (double)as[i].Ncols * wino4_exec_ratio(pws[i]->K, pws[i]->f44);
    f4 = f4 && wino4_nc(*pws[i]) == wino4_nc(*pws[0]) && wino4_args(*pws[i], as[i], dil, g.a[i], g4.a[i]);
    if (f4) { total4 += (long long)g4.a[i].ntn * g4.a[i].gy * B; g4.end[i] = (int)total4; g4.k[i] = pws[i]->K; }
    const int K = pws[i]->K;
    size_t l = 0;
    if (WM == 4) l = dil == 1 ? wino_lds<1, 4>(K) : (dil == 3 ? wino_lds<3, 4>(K) : wino_lds<5, 4>(K));
    else l = dil == 1 ? wino_lds<1, 2>(K) : (dil == 3 ? wino_lds<3, 2>(K) : wino_lds<5, 2>(K));
    lds = std::max(lds, l);
  }
  if (!f4 && pws[0]->mtiles < 2) return 1;                  // C = 32 exists in F(4,3) form only
  const bool f44 = f4 && pws[0]->f44;                       // k = 11 and k = 7 in F(4,4) form, k = 3 in F(4,3) (pack_wino)
  if (f4 && (!pws[0]->f44 || !pws[1]->f44 || pws[2]->f44)) return 1;
  // (the threshold counts the F(2,3) kernels' 64- / 128-column tiles whichever form runs: round 3's measured break-even for the
  // grouped launches - counting the F(4,3) tiles instead sent the 1 x 200 C = 128 / 64 stages to the direct kernels, +0.5 ms)
  if (total / B * variant_batch(B) < mrf_min_tiles()) return 1;
  const int in_perm = as[0].wperm_in, out_perm = as[0].wperm_out;
  for (int i = 1; i < n; ++i) if (as[i].wperm_in != in_perm || as[i].wperm_out != out_perm) return 1;
  if (query) return f4 ? 0 : 1;
  if ((in_perm || out_perm) && !f4) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "window-major rows exist in the grouped F(4,3) launches only");
  // the rows' partial last tiles go to the tail kernel where that saves a round of the persistent workgroups (conv_wino4.hip; the rounds are counted for
  // the variant batch, so that a shard of a job takes the same path as the job)
  int tail_w0 = 0, tail_nw = 0;
  if (f4 && dil > 1) {
    int w0 = 0, nw = 0;
    const int full = wino4_tail_plan(as[0].Ncols, dil, wino4_nc(*pws[0]), &w0, &nw);
    bool ok = full > 0;
    for (int i = 0; i < n; ++i) ok = ok && pws[i]->wraw.p && as[i].Ncols == as[0].Ncols && !(g4.a[i].flags & (F_RES | F_ACC | F_DIV)) && pws[i]->Cin * pws[i]->K * 32 <= 160 * 1024;
    if (ok) {
      const long long G = device_cu_count(), bv = variant_batch(B);
      long long per_b = 0, per_b_drop = 0;
      for (int i = 0; i < n; ++i) { per_b += (long long)g4.a[i].ntn * g4.a[i].gy; per_b_drop += (long long)full * g4.a[i].gy; }
      if ((per_b_drop * bv + G - 1) / G < (per_b * bv + G - 1) / G) {
        total4 = 0;
        for (int i = 0; i < n; ++i) { g4.a[i].ntn = full; total4 += (long long)full * g4.a[i].gy * B; g4.end[i] = (int)total4; }
        tail_w0 = w0; tail_nw = nw;
      }
    }
  }
  for (int i = n; i < 3; ++i) { g.end[i] = 0x7fffffff; g.k[i] = 3; }
  stats_add_conv(flops, n, f4 ? exec4 : exec_flops);
  int prof_idx = -1;
  if (prof_enabled()) {
    char d[160];
    snprintf(d, sizeof(d), "%s Ci%-4d Co%-4d k%d/%d/%d d%d N%-7d B%-3d %dx%d%s", f4 ? "wino4G" : "winoG", pws[0]->Cin, pws[0]->Cout, pws[0]->K, pws[1]->K, n > 2 ? pws[2]->K : 0, dil,
             as[0].Ncols, B, WM, 4 / WM, f44 ? " F(4,4)" : "");
    prof_idx = prof_begin(st, d, flops);
  }
  int rc = SVOC_OK;
  if (f4) {
    const float* wr[3]; int co[3];
    for (int i = 0; i < 3; ++i) { wr[i] = pws[i]->wraw.f(); co[i] = pws[i]->Cout; }
    rc = wino4_launch_group(g4, dil, wino4_nc(*pws[0]), in_perm, out_perm, f44, total4, st, wr, co, tail_w0, tail_nw, B);
  }
  else if (WM == 4) rc = dil == 1 ? wino_launch_group<1, 4>(g, total, lds, st) : (dil == 3 ? wino_launch_group<3, 4>(g, total, lds, st) : wino_launch_group<5, 4>(g, total, lds, st));
  else rc = dil == 1 ? wino_launch_group<1, 2>(g, total, lds, st) : (dil == 3 ? wino_launch_group<3, 2>(g, total, lds, st) : wino_launch_group<5, 2>(g, total, lds, st));
  prof_end(st, prof_idx);
  if (rc != SVOC_OK) return rc;
  SVOC_HIP(hipGetLastError());
  return SVOC_OK;
}

}  // namespace svoc
