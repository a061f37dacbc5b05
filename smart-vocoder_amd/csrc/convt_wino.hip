// Winograd F(4,2) form of the decoder's upsamplers (reference models.py:123-127, 147-148: lrelu(0.1) -> weight-normed
// ConvTranspose1d(C, C/2, k, stride s, padding (k - s)/2) with k = 2 s).  Round 3: the upsamplers ran on the generic kernel at
// 76-114 TFLOP/s (conv_mfma<2,2,2,2>, 2.6 VALU instructions per MFMA); this is the kernel form of conv_wino4.hip applied to them.
//
// Polyphase view: output n = s q + r - pad (q = 0 .. Lin, r = 0 .. s-1) of channel o is a TWO-tap filter along q,
//     y[o][s q + r - pad] = sum_c ( w[c][o][r] x[c][q] + w[c][o][r + s] x[c][q - 1] ),
// i.e. a GEMM with M = Cout * s rows (row = o s + r), K = Cin x 2.  Along q it is a minimal F(4,2) filtering: the four outputs
// q = 4w .. 4w+3 of a row from the five inputs d_k = x[4w - 1 + k] with five products instead of eight (points 0, 1, -1, 2, inf):
//     V0 = 2 d0 - d1 - 2 d2 + d3     U0 = g0 / 2               y0 = M0 + M1 + M2 + M3          (g0 = w[c][o][r + s], g1 = w[c][o][r])
//     V1 = -2 d1 - d2 + d3           U1 = -(g0 + g1) / 2       y1 = M1 - M2 + 2 M3
//     V2 = 2 d1 - 3 d2 + d3          U2 = (g1 - g0) / 6        y2 = M1 + M2 + 4 M3
//     V3 = d3 - d1                   U3 = g0 / 6 + g1 / 3      y3 = M1 - M2 + 8 M3 + M4
//     V4 = 2 d1 - d2 - 2 d3 + d4     U4 = g1                   M_p = sum_c U_p[row][c] V_p[c][w]     (the GEMM)
// The bias starts in M1 (part of all four outputs).  fp32 throughout.
//
// Kernel form (as conv_wino4.hip): persistent eight-wave workgroups, one per CU: four consumers (one 32-row tile x 32 windows =
// 128 columns q each; VALU-free MFMA stream: fragment reads two steps ahead at immediate LDS offsets, weights one slot ahead
// through buffer loads with the slot offset in an SGPR) and four producers (stage + leaky relu + input transform of their channel
// rows one stage ahead into the other of two plane sets; one workgroup barrier per stage; s_setprio 3).  A stage is 64 input
// channels (two weight chunks): 160 MFMAs per consumer.  Epilogue (s = 8, pad = 4): the four rows of an accumulator quad are four
// consecutive output samples of one channel, so a lane stores sixteen 16-byte groups.
#include "svoc_internal.h"
#include "wino_common.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace svoc {

typedef unsigned int ct_u32x4 __attribute__((ext_vector_type(4)));

// shader-clock stamp kept in scalar registers (the stamped build; __builtin_readcyclecounter() inside the consumer loop made
// the register allocator spill 800 bytes per lane)
__device__ __forceinline__ long long ct_clock() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  return (long long)t;
}

struct CtArgs {
  const float* x; long long x_bs; int x_ld; int Lin;      // input [B][Cin][x_ld], valid columns [0, Lin)
  float pre_slope;
  const float* wp; const float* bias; int nchunks; int mtiles;   // packed U (pack_ct), bias per OUTPUT CHANNEL
  float* y; long long y_bs; int y_ld; int Lout;           // output [B][Cout][y_ld], Lout = Lin * s
  int ntn; int gy; int B; int xcd;                        // tile id = (row block * B + batch) * ntn + column tile: tiles that share a
  unsigned flags;                                         // row block's weights are neighbours.  flags bit 8: producers at s_setprio 3
  long long* dbg;                                         // optional [workgroups][16] stamps (svoc_debug_set_stamp_buffer)
};

// NRT = row tiles per workgroup: 4 (128 rows x 32 windows; a stage = 64 channels), 2 (64 rows x 64 windows, the consumers as
// 2 row tiles x 2 column halves; a stage = 32 channels - the planes of 64 windows x 64 channels x 2 sets would not fit), or 8
// (256 rows x 32 windows: EIGHT consumer waves, two per SIMD, on the same planes - twelve waves of <= 168 registers.  The
// producers only get VALU issue slots in the gaps of their SIMD's MFMA stream; what they have not finished when the stream
// ends is what the consumers wait for at the stage barrier (tools/ct_timeline.py: 2.4-2.9k of 13k cycles per stage with one
// consumer per SIMD).  Twice the MFMAs over the same planes halve that share.)
template <int NRT>
struct CtGeo {
  static constexpr int NCT = NRT == 2 ? 2 : 1;
  static constexpr int NCW = NRT == 8 ? 8 : 4;            // consumer waves
  static constexpr int CPS = NRT == 2 ? 1 : 2;            // weight chunks per stage
  static constexpr int KS = CPS * KC;                     // channels per stage
  static constexpr int NWT = 32 * NCT;                    // windows per tile (= 4 NWT columns q)
  static constexpr int PQ = NWT;                          // plane row stride
  static constexpr int RAW = 4 * NWT + 4;                 // raw tile columns: from 4 w0 - 4 (d0 of window w0 is raw[3])
  static constexpr int NPL = 5, NACC = 5, WSLOTS = 5, NSTEP = 5 * 4;
  static constexpr int PLANE = KS * PQ;
  static constexpr int PLF = NPL * PLANE;
  static constexpr int RAW_FLOATS = KS * RAW;
  static constexpr int LDS_BYTES = (RAW_FLOATS + 2 * PLF) * 4;
};

// S = stride (8: k = 16, pad 4; 2: k = 4, pad 1)
template <int S, int NRT, bool DBG>
__global__ void __launch_bounds__(NRT == 8 ? 768 : 512) convt_wino_kernel(const CtArgs p, const int total) {
  using Geo = CtGeo<NRT>;
  constexpr int CPS = Geo::CPS;
  constexpr int KS = Geo::KS, PQ = Geo::PQ, RAW = Geo::RAW, PLANE = Geo::PLANE, PLF = Geo::PLF, NSTEP = Geo::NSTEP, WSLOTS = Geo::WSLOTS;
  constexpr int RPW = KS / 4;
  extern __shared__ __attribute__((aligned(16))) float wl[];
  float* const raw = wl;
  float* const pl = wl + Geo::RAW_FLOATS;
  const int v0 = blockIdx.x, stride = gridDim.x;
  if (v0 >= total) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Lin = p.Lin;
  const int nst = p.nchunks / CPS;                         // stages per tile (host: nchunks even)
  const int my_tiles = (total - v0 + stride - 1) / stride;
  const int nstages = my_tiles * nst;
  auto locate = [&](int v, int& w0_, int& bz_, int& by_) {
    const int tl = xcd_linear(v, total, p.xcd);
    const int t = tl / p.ntn;
    by_ = t / p.B;                                         // tiles that share a row block's weights (up to 1.3 MB) are neighbours
    w0_ = (tl - t * p.ntn) * Geo::NWT;
    bz_ = t - by_ * p.B;
    // a row's column tiles in an order rotated by t: with a power-of-two tile count a workgroup would otherwise meet the same
    // column tile in every round, and the eight that own the left edge (slower staging: bounds per element) finish last
    w0_ = (int)((unsigned)(w0_ / Geo::NWT + t) % (unsigned)p.ntn) * Geo::NWT;
  };

  if (wave >= Geo::NCW) {
    // ================================================================= producer
    const int pw_ = wave - Geo::NCW;
    if (p.flags & 0x100u) __builtin_amdgcn_s_setprio(3);
    constexpr int R4 = RAW / 4, NGW = RPW * R4, SPW = (NGW + 63) / 64;
    constexpr int NIW = RPW * Geo::NWT, TPW = (NIW + 63) / 64;
    const long long ldb = (long long)p.x_ld * 4;
    const float slope = p.pre_slope;
    unsigned goff[SPW];
    int gcol[SPW];
    float* rdst[SPW];
#pragma unroll
    for (int u = 0; u < SPW; ++u) {
      const int it = min(lane + 64 * u, NGW - 1);
      const int row = RPW * pw_ + it / R4, g4 = it % R4;
      goff[u] = (unsigned)(row * p.x_ld + 4 * g4) * 4u;
      gcol[u] = 4 * g4;
      rdst[u] = raw + row * RAW + 4 * g4;
    }
    const float* tsrc[TPW];
    int tdst[TPW];
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      const int it = min(lane + 64 * u, NIW - 1);
      const int row = RPW * pw_ + it / Geo::NWT, e = it % Geo::NWT;
      tsrc[u] = raw + row * RAW + 4 * e;                   // d0 = tsrc[3], d1..d4 = the aligned group tsrc[4..7]
      tdst[u] = row * PQ + e;
    }
    float4 v[SPW];
    int w0 = 0, bz = 0, by = 0;
    // staging modes: 0 = the raw tile lies inside the row; 1 = it crosses an end of a row whose length is a multiple of four
    // (16-byte groups are then entirely inside or outside: a select per group); 2 = general (bounds per element)
    const bool whole = (Lin & 3) == 0;
    auto mode_of = [&](int xs_) -> int { return (xs_ >= 0 && xs_ + RAW <= Lin) ? 0 : (whole ? 1 : 2); };
    auto issue = [&](const char* xb_, int xs_, int mode_, int st_) {
      const char* cb = xb_ + (long long)st_ * KS * ldb;
      if (mode_ == 0) {
        const char* ct = cb + (long long)xs_ * 4;
#pragma unroll
        for (int u = 0; u < SPW; ++u) v[u] = *reinterpret_cast<const float4*>(ct + goff[u]);
      } else if (mode_ == 1) {
        const unsigned lim = (unsigned)(Lin - 4);
#pragma unroll
        for (int u = 0; u < SPW; ++u) {
          const int tg = xs_ + gcol[u];
          const unsigned off = (unsigned)tg <= lim ? goff[u] + (unsigned)(xs_ * 4) : goff[u] - (unsigned)(gcol[u] * 4);      // outside: column 0 of the row (dropped on arrival)
          v[u] = *reinterpret_cast<const float4*>(cb + off);
        }
      } else {
        int l_ = lane;
        asm volatile("" : "+v"(l_));
#pragma unroll
        for (int u = 0; u < SPW; ++u) {
          const int it = min(l_ + 64 * u, NGW - 1);
          const int row = RPW * pw_ + it / R4, tg = xs_ + 4 * (it % R4);
          v[u] = *reinterpret_cast<const float4*>(cb + (long long)row * ldb + (long long)((tg >= 0 && tg + 3 < Lin) ? tg : 0) * 4);
        }
      }
    };
    locate(v0, w0, bz, by);
    {
      const int xs = 4 * w0 - 4;
      issue(reinterpret_cast<const char*>(p.x + (long long)bz * p.x_bs), xs, mode_of(xs), 0);
    }
    int ti = 0, ch = 0;
    long long pc_all0 = 0, pc_bar = 0;
    if constexpr (DBG) pc_all0 = ct_clock();
    long long pc_ld = 0, pc_st = 0, pc_tr = 0;
    for (int s_ = 0; s_ < nstages; ++s_) {
      long long q0 = 0, q1 = 0, q2 = 0;
      if constexpr (DBG) { q0 = ct_clock(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); q1 = ct_clock(); pc_ld += q1 - q0; }
      const int xs_start = 4 * w0 - 4;
      const int mode = mode_of(xs_start);
      const char* const xb = reinterpret_cast<const char*>(p.x + (long long)bz * p.x_bs);
      if (mode == 0) {
#pragma unroll
        for (int u = 0; u < SPW; ++u) {
          if (64 * (u + 1) <= NGW || lane < NGW - 64 * u) {
            float4 q = v[u];
            wino_lrelu4(q, slope);
            *reinterpret_cast<float4*>(rdst[u]) = q;
          }
        }
      } else if (mode == 1) {
        const unsigned lim = (unsigned)(Lin - 4);
#pragma unroll
        for (int u = 0; u < SPW; ++u) {
          if (64 * (u + 1) <= NGW || lane < NGW - 64 * u) {
            float4 q = v[u];
            if ((unsigned)(xs_start + gcol[u]) > lim) q = make_float4(0.f, 0.f, 0.f, 0.f);
            wino_lrelu4(q, slope);
            *reinterpret_cast<float4*>(rdst[u]) = q;
          }
        }
      } else {
        int l_ = lane;
        asm volatile("" : "+v"(l_));
#pragma unroll
        for (int u = 0; u < SPW; ++u) {
          if (64 * (u + 1) <= NGW || lane < NGW - 64 * u) {
            const int it = l_ + 64 * u;
            const int row = RPW * pw_ + it / R4, tg = xs_start + 4 * (it % R4);
            float4 q = v[u];
            if (!(tg >= 0 && tg + 3 < Lin)) {
              const float* xr = reinterpret_cast<const float*>(xb + (long long)(ch * KS + row) * ldb);
              q.x = (tg >= 0 && tg < Lin) ? xr[tg] : 0.f;
              q.y = (tg + 1 >= 0 && tg + 1 < Lin) ? xr[tg + 1] : 0.f;
              q.z = (tg + 2 >= 0 && tg + 2 < Lin) ? xr[tg + 2] : 0.f;
              q.w = (tg + 3 >= 0 && tg + 3 < Lin) ? xr[tg + 3] : 0.f;
            }
            wino_lrelu4(q, slope);
            *reinterpret_cast<float4*>(rdst[u]) = q;
          }
        }
      }
      if constexpr (DBG) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); q2 = ct_clock(); pc_st += q2 - q1; }
      int nti = ti, nchn = ch + 1, w0n = w0, bzn = bz, byn = by;
      if (nchn == nst) { nchn = 0; ++nti; if (nti < my_tiles) locate(v0 + nti * stride, w0n, bzn, byn); }
      if (s_ + 1 < nstages) {
        const int xsn = 4 * w0n - 4;
        issue(reinterpret_cast<const char*>(p.x + (long long)bzn * p.x_bs), xsn, mode_of(xsn), nchn);
      }
      float* const pb = pl + (s_ & 1) * PLF;
#pragma unroll
      for (int u = 0; u < TPW; ++u) {
        if (64 * (u + 1) <= NIW || lane < NIW - 64 * u) {
          const float* r = tsrc[u];
          float* o = pb + tdst[u];
          const float d0 = r[3];
          const float4 fm = *reinterpret_cast<const float4*>(r + 4);
          const float d1 = fm.x, d2 = fm.y, d3 = fm.z, d4 = fm.w;
          const float v3 = d3 - d1;
          o[0] = __builtin_fmaf(2.f, d0 - d2, v3);                               // 2 d0 - d1 - 2 d2 + d3
          o[PLANE] = __builtin_fmaf(-2.f, d1, d3 - d2);                          // -2 d1 - d2 + d3
          o[2 * PLANE] = __builtin_fmaf(2.f, d1, __builtin_fmaf(-3.f, d2, d3));  // 2 d1 - 3 d2 + d3
          o[3 * PLANE] = v3;
          o[4 * PLANE] = __builtin_fmaf(-2.f, v3, d4 - d2);                      // 2 d1 - d2 - 2 d3 + d4
        }
      }
      long long pb0 = 0;
      if constexpr (DBG) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); pb0 = ct_clock(); pc_tr += pb0 - q2; }
      __syncthreads();
      if constexpr (DBG) pc_bar += ct_clock() - pb0;
      ti = nti; ch = nchn; w0 = w0n; bz = bzn; by = byn;
    }
    if constexpr (DBG) if (tid == 64 * Geo::NCW) {                   // producer wave 0: total cycles, cycles waiting at the stage barriers
      long long* d = p.dbg + 16 * (long long)v0;
      d[8] = ct_clock() - pc_all0; d[9] = pc_bar; d[12] = pc_ld; d[13] = pc_st; d[14] = pc_tr;      // waiting for the global loads, staging, issue + transform
    }
    return;
  }

  // =================================================================== consumer: row tile rt of the block, column group cg
  const int l31 = lane & 31, hi = lane >> 5;
  const int rt = NRT == 2 ? (wave & 1) : wave, cg = NRT == 2 ? (wave >> 1) : 0;
  const unsigned pbase = (unsigned)(size_t)pl;
  const unsigned baddr0 = pbase + (unsigned)(hi * PQ + 32 * cg + l31) * 4u;
  const unsigned wlane = (unsigned)lane * 16u;
  f32x16 M[5];
  float4 a[2][4];
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wp), 0, 0x7fffffff, 0x00020000);
  auto wload1 = [&](float4& dst, int soff, auto kg_c) {
    const ct_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)wlane + decltype(kg_c)::value * 1024, soff, 0);
    dst = *reinterpret_cast<const float4*>(&t);
  };
  auto wload4 = [&](float4 (&dst)[4], int soff) {
    wload1(dst[0], soff, std::integral_constant<int, 0>{}); wload1(dst[1], soff, std::integral_constant<int, 1>{});
    wload1(dst[2], soff, std::integral_constant<int, 2>{}); wload1(dst[3], soff, std::integral_constant<int, 3>{});
  };
  // one 32-channel chunk: 20 steps of four MFMAs.  CC = parity of the chunk (which weight register set its first slot is in),
  // PC = its position in the stage's planes
  auto mfma_chunk = [&](const unsigned baddr, const int wa, const int wnext_, auto cc, auto pc) {
    const int wnext = wnext_ >= 0 ? wnext_ : 0;          // after the last tile: harmless loads of the image's first slots
    constexpr int CC = decltype(cc)::value, PC = decltype(pc)::value;
    float fb[2][4];
    auto request = [&](auto tc) {
      constexpr int T = decltype(tc)::value;
      if constexpr (T < NSTEP) wino_frag<PQ, (T / 4) * PLANE + PC * KC * PQ, T % 4, 0>(fb[T & 1], baddr);
    };
    auto step = [&](auto tc) {
      constexpr int T = decltype(tc)::value;
      constexpr int WS = T / 4, KG = T % 4;
      {
        float(&b)[4] = fb[T & 1];
        if constexpr (T + 1 < NSTEP) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
      }
      const float4 av = a[(CC + WS) & 1][KG];
#pragma unroll
      for (int s = 0; s < 4; ++s) M[WS] = __builtin_amdgcn_mfma_f32_32x32x2f32(wino_pick(av, s), fb[T & 1][s], M[WS], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // the k-group just used is refilled with the same k-group of the slot TWO ahead (same register set): the load has
      // 32 MFMAs = ~2000 cycles to arrive (one slot ahead, ~1000, left the stream waiting on L2 for ~12 % of its time)
      if constexpr (WS + 2 < WSLOTS) wload1(a[(CC + WS) & 1][KG], wa + (WS + 2) * 4096, std::integral_constant<int, KG>{});
      else wload1(a[(CC + WS) & 1][KG], wnext + (WS + 2 - WSLOTS) * 4096, std::integral_constant<int, KG>{});
      request(std::integral_constant<int, T + 2>{});
      __builtin_amdgcn_sched_barrier(0);
    };
    request(std::integral_constant<int, 0>{});
    request(std::integral_constant<int, 1>{});
    wino_static_for<0, NSTEP>(step);
  };
  // Weight register sets: slot ws of a chunk of parity CC lives in set (CC + ws) & 1 (five slots per chunk: the first slot of the
  // next chunk continues the alternation; chunks go in pairs = ten slots, so every pair starts in set 0).
  auto wtile = [&](int mt_) -> int { return __builtin_amdgcn_readfirstlane(mt_ * p.nchunks * WSLOTS * 4096); };
  const unsigned ylb = (unsigned)p.y_ld * 4u;
  float bv[16];
  int bias_mt = -1;
  long long cyc_bar = 0, cyc_mf = 0, cyc_epi = 0, cyc_all0 = 0, wall0 = 0;      // diagnostics (stamped build): consumer wave 0
  if constexpr (DBG) { cyc_all0 = ct_clock(); wall0 = (long long)wall_clock64(); }
  for (int ti = 0; ti < my_tiles; ++ti) {
    int w0, bz, by;
    locate(v0 + ti * stride, w0, bz, by);
    const int mt = by * NRT + rt;
    const bool row_ok = mt < p.mtiles;
    const int mtc = row_ok ? mt : p.mtiles - 1;
    const int wt = wtile(mtc);
    if (ti == 0) { wload4(a[0], wt); wload4(a[1], wt + 4096); }
    {   // rows 32 mt + 8 Q + 4 hi + j belong to output channel (32 mt) / S + ... : the bias of an accumulator row starts in M1
#pragma unroll
      for (int q = 0; q < 5; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) M[q][i] = 0.f;
      if (mtc != bias_mt) {                                // (kept across the tiles of one row block: the loads' latency is exposed)
        bias_mt = mtc;
#pragma unroll
        for (int i = 0; i < 16; ++i) bv[i] = p.bias[(mtc * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi) / S];
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) M[1][i] = bv[i];
    }
    int wnext_tile = -1;
    if (ti + 1 < my_tiles) {
      int w0n, bzn, byn;
      locate(v0 + (ti + 1) * stride, w0n, bzn, byn);
      const int mtn = byn * NRT + rt;
      wnext_tile = wtile(mtn < p.mtiles ? mtn : p.mtiles - 1);
    }
    long long c_prev = 0;
    for (int cp = 0; cp < p.nchunks; cp += 2) {            // chunk pairs
      const int wa = wt + cp * WSLOTS * 4096;
      const int wnext = cp + 2 < p.nchunks ? wa + 2 * WSLOTS * 4096 : wnext_tile;
      long long c0 = 0, c1 = 0;                            // (stamps: a stream ends where the next barrier wait or the epilogue begins)
      if constexpr (CPS == 2) {
        const int s_ = ti * nst + (cp >> 1);
        if constexpr (DBG) c0 = ct_clock();
        __syncthreads();
        if constexpr (DBG) c1 = ct_clock();
        const unsigned off = (unsigned)((s_ & 1) * PLF) * 4u;
        mfma_chunk(baddr0 + off, wa, wa + WSLOTS * 4096, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        mfma_chunk(baddr0 + off, wa + WSLOTS * 4096, wnext, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        if constexpr (DBG) { cyc_bar += c1 - c0; if (c_prev) cyc_mf += c0 - c_prev; c_prev = c1; }
      } else {
        const int s_ = ti * nst + cp;
        if constexpr (DBG) c0 = ct_clock();
        __syncthreads();
        if constexpr (DBG) c1 = ct_clock();
        mfma_chunk(baddr0 + (unsigned)((s_ & 1) * PLF) * 4u, wa, wa + WSLOTS * 4096, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        if constexpr (DBG) { cyc_bar += c1 - c0; if (c_prev) cyc_mf += c0 - c_prev; c_prev = c1; }      // (the pair's second barrier wait counts as stream time)
        __syncthreads();
        mfma_chunk(baddr0 + (unsigned)(((s_ + 1) & 1) * PLF) * 4u, wa + WSLOTS * 4096, wnext, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
      }
    }
    // ---- output transform + polyphase scatter
    if (!row_ok) continue;
    const int LoutE = p.Lout;
    long long ce0 = 0;
    if constexpr (DBG) { ce0 = ct_clock(); cyc_mf += ce0 - c_prev; }
    char* const yb = reinterpret_cast<char*>(p.y + (long long)bz * p.y_bs);
    const int qc = 4 * (w0 + 32 * cg + l31);               // this lane's first column q (then +1, +2, +3)
    if constexpr (S == 8) {
      // rows of quad Q: output channel 4 mt + Q, phases r = 4 hi .. 4 hi + 3: samples n = 8 q + 4 hi - 4 .. + 3
#pragma unroll
      for (int Q = 0; Q < 4; ++Q) {
        float4 yv[4];                                      // yv[i] = column q = qc + i, its four phases
        float* yf = reinterpret_cast<float*>(yv);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * Q + j;
          const float t1 = M[1][r] + M[2][r], t2 = M[1][r] - M[2][r];
          yf[0 * 4 + j] = M[0][r] + t1 + M[3][r];
          yf[1 * 4 + j] = __builtin_fmaf(2.f, M[3][r], t2);
          yf[2 * 4 + j] = __builtin_fmaf(4.f, M[3][r], t1);
          yf[3 * 4 + j] = __builtin_fmaf(8.f, M[3][r], t2) + M[4][r];
        }
        char* const yrow = yb + (size_t)(4 * mt + Q) * ylb;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int n = 8 * (qc + i) + 4 * hi - 4;
          if (n >= 0 && n + 3 < LoutE) *reinterpret_cast<float4*>(yrow + (long long)n * 4) = yv[i];
        }
      }
    } else {
      // S == 2: rows of quad Q: channels 16 mt + 4 Q + 2 hi + (0, 1), phases r = 0, 1: samples n = 2 q + r - 1
#pragma unroll
      for (int Q = 0; Q < 4; ++Q) {
        float yv[4][4];                                    // [column i][row j]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * Q + j;
          const float t1 = M[1][r] + M[2][r], t2 = M[1][r] - M[2][r];
          yv[0][j] = M[0][r] + t1 + M[3][r];
          yv[1][j] = __builtin_fmaf(2.f, M[3][r], t2);
          yv[2][j] = __builtin_fmaf(4.f, M[3][r], t1);
          yv[3][j] = __builtin_fmaf(8.f, M[3][r], t2) + M[4][r];
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {                      // the quad's two channels: rows 2 c (phase 0), 2 c + 1 (phase 1)
          float* const yrow = reinterpret_cast<float*>(yb + (size_t)(16 * mt + 4 * Q + 2 * hi + c) * ylb);
          const int n0 = 2 * qc - 1;                       // eight consecutive samples n0 .. n0 + 7: (i, phase) = (0,0) (0,1) (1,0) ...
          // (round 6) the lane's samples start one short of a 32-byte boundary: its first one is stored by the lane to its LEFT, which thereby writes two
          // aligned 16-byte groups [n0 + 1, n0 + 9) instead of 4 + 16 + 8 + 4 bytes; the first lane of the 32 and a lane whose left neighbour is off the row
          // store their own first sample, the last lane of the 32 keeps the short form
          const float nxt = __shfl_down(yv[0][2 * c], 1, 64);
          if (n0 >= 0 && n0 + 7 < LoutE) {
            if (l31 == 0 || n0 < 8) yrow[n0] = yv[0][2 * c];
            *reinterpret_cast<float4*>(yrow + n0 + 1) = make_float4(yv[0][2 * c + 1], yv[1][2 * c], yv[1][2 * c + 1], yv[2][2 * c]);
            if (l31 < 31 && n0 + 8 < LoutE) {
              *reinterpret_cast<float4*>(yrow + n0 + 5) = make_float4(yv[2][2 * c + 1], yv[3][2 * c], yv[3][2 * c + 1], nxt);
            } else {
              *reinterpret_cast<float2*>(yrow + n0 + 5) = make_float2(yv[2][2 * c + 1], yv[3][2 * c]);
              yrow[n0 + 7] = yv[3][2 * c + 1];
            }
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int ph = 0; ph < 2; ++ph) {
                const int n = n0 + 2 * i + ph;
                if (n >= 0 && n < LoutE) yrow[n] = yv[i][2 * c + ph];
              }
          }
        }
      }
    }
    if constexpr (DBG) cyc_epi += ct_clock() - ce0;
  }
  if constexpr (DBG) if (tid == 0) {      // [workgroup][16]: 0 tiles, 1 total cycles, 2 barrier waits, 3 MFMA streams, 4 epilogues, 5 marker, 8/9 producer
    long long* d = p.dbg + 16 * (long long)v0;
    d[0] = my_tiles; d[1] = ct_clock() - cyc_all0; d[2] = cyc_bar; d[3] = cyc_mf; d[4] = cyc_epi; d[5] = 5;
    d[10] = wall0; d[11] = (long long)wall_clock64();     // 100 MHz constant clock
  }
}

// ------------------------------------------------------------------ weight transform + packing
// wp[m-tile][chunk][slot p = 0..4][k-group][lane][4]: lane l of k-step 4 kg + s holds U_p of row 32 mt + (l & 31) (= o S + r),
// input channel 32 chunk + 2 (4 kg + s) + (l >> 5).  Source: ConvTranspose1d weight [Cin][Cout][2 S] (weight_v with weight_g per
// INPUT channel, reference weight_norm dim 0, or the folded weight).
__global__ void pack_ct_kernel(const float* __restrict__ src, const float* __restrict__ scale, float* __restrict__ wp, int Cin, int Cout,
                               int S, int nchunks, long long total) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int s = (int)(e & 3);
  const int lane = (int)((e >> 2) & 63);
  long long rest = e >> 8;
  const int kg = (int)(rest & 3); rest >>= 2;
  const int slot = (int)(rest % 5); rest /= 5;
  const int ch = (int)(rest % nchunks);
  const int mt = (int)(rest / nchunks);
  const int row = mt * 32 + (lane & 31);
  const int chan = ch * KC + 2 * (4 * kg + s) + (lane >> 5);
  float val = 0.f;
  if (row < Cout * S && chan < Cin) {
    const int o = row / S, r = row - o * S;
    const float* w = src + ((long long)chan * Cout + o) * (2 * S);
    const float sc = scale ? scale[chan] : 1.0f;
    const float g1 = w[r] * sc, g0 = w[r + S] * sc;         // g1 multiplies x[q], g0 multiplies x[q - 1]
    switch (slot) {
      case 0: val = 0.5f * g0; break;
      case 1: val = -0.5f * (g0 + g1); break;
      case 2: val = (g1 - g0) * (1.0f / 6.0f); break;
      case 3: val = g0 * (1.0f / 6.0f) + g1 * (1.0f / 3.0f); break;
      default: val = g1; break;
    }
  }
  wp[e] = val;
}
__global__ void ct_scale_kernel(const float* __restrict__ v, const float* __restrict__ g, float* __restrict__ scale, long long inner) {
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

// The column q = Lin (the last `pad` samples of every row: only the x[q - 1] tap is inside the input) when Lin % 4 == 0: the
// window tiles then cover q = 0 .. Lin - 1 exactly and one more tile per row for a single column would cost 1/16 .. 1/4 of the
// launch (513 columns at the first upsampler).  y[b][o][S Lin - pad + r] = bias[o] + sum_c g0[c][o][r] lrelu(x[b][c][Lin - 1]),
// r < pad; g0 = 2 U0 is read back from the image.  One workgroup per (four row tiles, utterance); wave k takes the chunks
// k, k + 16, ...; a lane one of the 64 tail rows.
template <int S>
__global__ void __launch_bounds__(1024) convt_tail_kernel(const CtArgs p, const int Cin) {
  __shared__ float xl[1024];
  __shared__ float red[16][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bz = blockIdx.y;
  const float* xb = p.x + (long long)bz * p.x_bs + (p.Lin - 1);
  for (int c = tid; c < Cin; c += 1024) {
    const float v = xb[(long long)c * p.x_ld];
    xl[c] = fmaxf(v, v * p.pre_slope);
  }
  __syncthreads();
  const int mt = blockIdx.x * 4 + (lane >> 4), tr = lane & 15;
  const int rl = S == 8 ? (tr >> 2) * 8 + (tr & 3) : 2 * tr;      // row of the tile: phases r < pad
  float acc = 0.f;
  if (mt < p.mtiles) {
    for (int ch = wave; ch < p.nchunks; ch += 16) {
      const float* w = p.wp + ((long long)(mt * p.nchunks + ch) * 5) * 1024 + rl * 4;      // slot 0: U0 = g0 / 2
      const float* xc = xl + ch * KC;
#pragma unroll
      for (int kg = 0; kg < 4; ++kg)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float4 u = *reinterpret_cast<const float4*>(w + kg * 256 + h * 128);
          acc += u.x * xc[8 * kg + h] + u.y * xc[8 * kg + 2 + h] + u.z * xc[8 * kg + 4 + h] + u.w * xc[8 * kg + 6 + h];
        }
    }
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && mt < p.mtiles) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][lane];
    const int row = mt * 32 + rl, o = row / S, r = row - o * S;
    p.y[(long long)bz * p.y_bs + (long long)o * p.y_ld + (p.Lout - (S == 8 ? 4 : 1) + r)] = 2.f * t + p.bias[o];
  }
}

bool convt_wino_enabled() {
  static const bool on = !(getenv("SVOC_CT_WINO") && atoi(getenv("SVOC_CT_WINO")) == 0);
  return on;
}
bool convt_wino_supported(int Cin, int Cout, int K, int stride, int tpad) {
  if (!convt_wino_enabled()) return false;
  if (!((stride == 8 && K == 16 && tpad == 4) || (stride == 2 && K == 4 && tpad == 1))) return false;
  return Cin % (2 * KC) == 0 && (Cout * stride) % 64 == 0;
}

int pack_convt_wino(PackedCtWino& pw, int Cin, int Cout, int K, int stride, int tpad, const float* w_or_v, const float* g,
                    const float* bias, hipStream_t st) {
  if (!convt_wino_supported(Cin, Cout, K, stride, tpad)) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "pack_convt_wino: unsupported shape");
  pw.Cin = Cin; pw.Cout = Cout; pw.S = stride; pw.tpad = tpad;
  pw.nchunks = Cin / KC;
  pw.mtiles = Cout * stride / 32;
  pw.flops_per_col = 2.0 * Cin * Cout * K;                 // algorithmic 2*MAC per INPUT column (as the direct kernel counts it)
  const long long total = (long long)pw.mtiles * pw.nchunks * 5 * 4 * 256;
  if (total * 4 >= (1LL << 31)) SVOC_FAIL(SVOC_ERR_UNSUPPORTED, "pack_convt_wino: weight image too large for 32-bit offsets");
  SVOC_TRY(pw.wp.ensure((size_t)(total + 1024) * sizeof(float)));
  SVOC_HIP(hipMemsetAsync(pw.wp.f() + total, 0, 1024 * sizeof(float), st));
  SVOC_TRY(pw.bias.ensure((size_t)Cout * sizeof(float)));
  if (bias) SVOC_HIP(hipMemcpyAsync(pw.bias.p, bias, (size_t)Cout * sizeof(float), hipMemcpyDeviceToDevice, st));
  else SVOC_HIP(hipMemsetAsync(pw.bias.p, 0, (size_t)Cout * sizeof(float), st));
  DevBuf scale;
  if (g) {
    SVOC_TRY(scale.ensure((size_t)Cin * sizeof(float)));
    hipLaunchKernelGGL(ct_scale_kernel, dim3((unsigned)Cin), dim3(256), 0, st, w_or_v, g, scale.f(), (long long)Cout * K);
  }
  hipLaunchKernelGGL(pack_ct_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w_or_v, g ? scale.f() : nullptr, pw.wp.f(),
                     Cin, Cout, stride, pw.nchunks, total);
  SVOC_HIP(hipGetLastError());
  SVOC_HIP(hipStreamSynchronize(st));
  return SVOC_OK;
}

int pack_convt_wino_named(PackedCtWino& pw, int Cin, int Cout, int K, int stride, int tpad, const TensorTable& tab,
                          const std::string& prefix, hipStream_t st) {
  const svoc_tensor* w = tab.find(prefix + ".weight");
  const svoc_tensor* v = tab.find(prefix + ".weight_v");
  const svoc_tensor* g = tab.find(prefix + ".weight_g");
  const svoc_tensor* b = tab.find(prefix + ".bias");
  const svoc_tensor* src = w ? w : v;
  if (!src || (!w && !g)) SVOC_FAIL(SVOC_ERR_MISSING_TENSOR, "missing tensor %s.weight / .weight_v / .weight_g", prefix.c_str());
  if (src->ndim != 3 || src->shape[0] != Cin || src->shape[1] != Cout || src->shape[2] != K)
    SVOC_FAIL(SVOC_ERR_SHAPE, "tensor %s has the wrong shape for a %d->%d k=%d transposed convolution", src->name, Cin, Cout, K);
  return pack_convt_wino(pw, Cin, Cout, K, stride, tpad, src->data, w ? nullptr : g->data, b ? b->data : nullptr, st);
}

template <int S, int NRT>
static int ct_launch_one(const CtArgs& a, unsigned grid, int total, hipStream_t st) {
  static_assert(CtGeo<NRT>::LDS_BYTES <= 160 * 1024, "tile does not fit");
  if (a.dbg) {                                             // stamped build (tools/ct_timeline.py)
    auto kern = convt_wino_kernel<S, NRT, true>;
    SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * (CtGeo<NRT>::NCW + 4)), (size_t)CtGeo<NRT>::LDS_BYTES, st, a, total);
    return SVOC_OK;
  }
  auto kern = convt_wino_kernel<S, NRT, false>;
  SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * (CtGeo<NRT>::NCW + 4)), (size_t)CtGeo<NRT>::LDS_BYTES, st, a, total);
  return SVOC_OK;
}

// x [B][Cin][x_ld] (lrelu(pre_slope) applied while staging) -> y [B][Cout][y_ld], Lout = Lin * S.  1 = not eligible.
int launch_convt_wino(const PackedCtWino& pw, const float* x, long long x_bs, int x_ld, float pre_slope, float* y, long long y_bs,
                      int y_ld, int B, int Lin, hipStream_t st) {
  if (B <= 0 || Lin <= 0 || !pw.wp.p) return 1;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (x_ld & 3) || (x_bs & 3)) return 1;
  if ((reinterpret_cast<uintptr_t>(y) & 15) || (y_ld & 3) || (y_bs & 3)) return 1;
  if ((long long)2 * KC * x_ld * 4 >= (1LL << 31)) return 1;
  if (!(pre_slope > 0.f && pre_slope <= 1.f)) return 1;     // the staging's leaky relu is max(x, slope x)
  CtArgs a;
  a.x = x; a.x_bs = x_bs; a.x_ld = x_ld; a.Lin = Lin; a.pre_slope = pre_slope;
  a.wp = pw.wp.f(); a.bias = pw.bias.f(); a.nchunks = pw.nchunks; a.mtiles = pw.mtiles;
  a.y = y; a.y_bs = y_bs; a.y_ld = y_ld; a.Lout = Lin * pw.S;
  const bool tail = (Lin & 3) == 0 && pw.Cin <= 1024;     // the column q = Lin by convt_tail_kernel, the windows cover 0 .. Lin - 1
  const int nw = (Lin + (tail ? 0 : 1) + 3) / 4;          // windows over the columns q = 0 .. Lin
  // 256-row blocks (eight consumers) where they leave two workgroups per CU, else 128-row blocks, or 64 rows x twice the windows
  // fewest workgroups the launch is worth it for: half the CUs (measured per launch, direct -> F(4,2) in us: 256 tiles 156 -> 74 and
  // 91 -> 58, 128 tiles 78 -> 71, 100 tiles 42 -> 51 and 28 -> 35, 32 tiles 58 -> 77); SVOC_CT_MIN_TILES overrides
  static const long long min_cfg = getenv("SVOC_CT_MIN_TILES") ? atoll(getenv("SVOC_CT_MIN_TILES")) : -1;
  const long long min_tiles = min_cfg >= 0 ? min_cfg : device_cu_count() / 2;
  int nrt = pw.mtiles % 4 == 0 ? 4 : 2;
  if (pw.mtiles % 8 == 0 && (long long)((nw + 31) / 32) * (pw.mtiles / 8) * variant_batch(B) >= 2LL * device_cu_count()) nrt = 8;
  const int nwt = nrt == 2 ? 64 : 32;
  a.ntn = (nw + nwt - 1) / nwt;
  a.gy = pw.mtiles / nrt; a.B = B;
  a.xcd = xcd_mapping_enabled();
  a.flags = 0x100u;
  a.dbg = debug_stamp_buffer();
  const long long total = (long long)a.ntn * a.gy * B;
  if (total > 0x7fffffffLL || total * variant_batch(B) / B < min_tiles) return 1;      // short inputs: the generic kernel
  const double flops = pw.flops_per_col * (double)B * (double)Lin;
  stats_add_conv(flops, 1, flops * 5.0 / 8.0);
  int prof_idx = -1;
  if (prof_enabled()) {
    char d[160];
    snprintf(d, sizeof(d), "convTw Ci%-4d Co%-4d k%-2d s%d N%-7d B%-3d F(4,2)", pw.Cin, pw.Cout, 2 * pw.S, pw.S, Lin + 1, B);
    prof_idx = prof_begin(st, d, flops);
  }
  const unsigned grid = (unsigned)std::min<long long>(total, (long long)device_cu_count());
  if (tail) {
    const dim3 tg((unsigned)((pw.mtiles + 3) / 4), (unsigned)B);
    if (pw.S == 8) hipLaunchKernelGGL(convt_tail_kernel<8>, tg, dim3(1024), 0, st, a, pw.Cin);
    else hipLaunchKernelGGL(convt_tail_kernel<2>, tg, dim3(1024), 0, st, a, pw.Cin);
  }
  int rc = SVOC_OK;
  if (pw.S == 8) rc = nrt == 8 ? ct_launch_one<8, 8>(a, grid, (int)total, st) : (nrt == 4 ? ct_launch_one<8, 4>(a, grid, (int)total, st) : ct_launch_one<8, 2>(a, grid, (int)total, st));
  else rc = nrt == 8 ? ct_launch_one<2, 8>(a, grid, (int)total, st) : (nrt == 4 ? ct_launch_one<2, 4>(a, grid, (int)total, st) : ct_launch_one<2, 2>(a, grid, (int)total, st));
  if (rc < 0) { prof_end(st, prof_idx); return rc; }
  prof_end(st, prof_idx);
  SVOC_HIP(hipGetLastError());
  return SVOC_OK;
}

}  // namespace svoc
