// Winograd F(4,3) implicit-GEMM convolution for the ResBlock convolutions of the decoder's C >= 128 stages (reference
// modules.py:190-207: convs1 / convs2, k = 3 / 7 / 11).  Round 3: the F(2,3) kernels (conv_wino.hip) issue 15/21 of the
// direct form's multiply-adds and keep the matrix pipe ~80 % busy, so the next lever is again FEWER MFMAs: F(4,3) issues 12/21.
//
// A k-tap convolution is split into three-tap groups at tap offsets 0, 4, 8 plus the left-over taps 3, 7 (as in conv_wino.hip).
// Each group is a minimal F(4,3) filtering: the four outputs y[4q .. 4q+3] from the six inputs d_j = x[4q - pad + 4g + j]
// with six products instead of twelve (Lavin & Gray's matrices, correlation form):
//     V0 = 4 d0 - 5 d2 + d4            U0 = w0 / 4                      y0 = M0 + M1 + M2 + M3 + M4
//     V1 = -4 d1 - 4 d2 + d3 + d4      U1 = -(w0 + w1 + w2) / 6         y1 = M1 - M2 + 2 M3 - 2 M4
//     V2 =  4 d1 - 4 d2 - d3 + d4      U2 = -(w0 - w1 + w2) / 6         y2 = M1 + M2 + 4 M3 + 4 M4
//     V3 = -2 d1 - d2 + 2 d3 + d4      U3 = (w0 + 2 w1 + 4 w2) / 24     y3 = M1 - M2 + 8 M3 - 8 M4 + M5
//     V4 =  2 d1 - d2 - 2 d3 + d4      U4 = (w0 - 2 w1 + 4 w2) / 24
//     V5 =  4 d1 - 5 d3 + d5           U5 = w2                          M_p = sum_c U_p[c] * V_p[c]   (the GEMM)
// The window step (4) equals the group spacing, so every group reads the SAME six transformed planes V_p[c][q'] at
// q' = q + g and accumulates into ONE set of six transform-domain accumulators.  A left-over tap contributes
// w * x[4q + r + 4t + 3 - pad] to output r of the tile; with X_r'[q'] = d_{1+r'} of window q' (the four samples a window adds to
// its predecessor, stored by the same transform item) that is X_{(r+2)%4}[q + t + (r+2)/4]: r = 0 goes into M0 (part of y0
// only), r = 3 into M5 (part of y3 only), r = 1 / 2 into two accumulators of their own.  MFMAs per output and channel pair:
// k=3: 1.5 (F(2,3): 2, direct 3), k=7: 4 (5, 7), k=11: 6.5 (8, 11).  Dilation D through the polyphase view (as conv_wino.hip):
// the four outputs of a window are n, n + D, n + 2D, n + 3D; a tile holds 32 consecutive windows w = D b + ph (conv_wino4.h: W4Geo::NWT).
// fp32 throughout; the transforms scale by up to 8, measured waveform error below (tests assert <= 1e-4 relative RMS).
//
// Kernel form: wave-specialised persistent workgroups, ONE per CU (eight accumulator tiles = 128 registers per consumer):
// waves 0-3 = consumers, one 32-row tile x 32 windows (= 128 outputs) each: nothing but the MFMA stream (fragment reads two
// steps ahead at immediate LDS offsets, weights one slot ahead) and the epilogue; waves 4-7 = producers: producer p stages
// and transforms channel rows 8p .. 8p+7 of every 32-channel chunk one stage ahead into the other of two plane sets; one
// workgroup barrier per stage.  A stage carries 16 * (6 G + 4 ND) = 96 / 256 / 416 MFMAs per consumer (k = 3 / 7 / 11).
#pragma once
#include "svoc_internal.h"
#include "wino_common.h"
#include "conv_wino4.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace svoc {

// PERM_: D = 1: the input rows are window-major for dilation PERM_ (W4Geo::PERM); D > 1: nonzero = the OUTPUT rows are written
// window-major (compile-time: as a run-time branch the two epilogues together cost the dilated kernels ~100 spilled registers)
// F44: the convolution runs in F(4,4) form (seven planes, no left-over taps; conv_wino4.h) on a weight image packed for that form
template <int K, int D, int NRT = 4, bool DBG = false, int PERM_ = 0, bool F44 = false>
__device__ __forceinline__ void wino4_problem(const WinoArgs& p, const int v0, const int vend, const int first, const int stride) {
  constexpr int PERM = D == 1 ? PERM_ : 0;
  constexpr bool OPERM = D > 1 && PERM_ != 0;
  using Geo = W4Geo<K, D, NRT, PERM, 1, F44>;
  constexpr int RAWS = Geo::RAWS, PORG = Geo::PORG;
  constexpr int PADT = Geo::PADT, PQ = Geo::PQ, RAW = Geo::RAW, WSLOTS = Geo::WSLOTS;
  constexpr int NWT = Geo::NWT, XOFF = Geo::XOFF, NE = Geo::NE, LEAD = Geo::LEAD, PLANE = Geo::PLANE, PLF = Geo::PLF;
  constexpr int NSTEP = Geo::NSTEP, NACC = Geo::NACC, CPS = Geo::CPS, KS = Geo::KS, RPW = KS / 4, HALVES = Geo::HALVES, KGS = Geo::KGS;
  constexpr int NSET = Geo::NSET, PD = Geo::PD, NPS = Geo::NPS;
  extern __shared__ __attribute__((aligned(16))) float wl[];
  float* const raw = wl;                                   // [KC][RAW], producers only
  float* const pl = wl + Geo::RAW_FLOATS;                  // two plane sets of PLF floats
  if (v0 >= vend) return;
  if constexpr (DBG) {                                     // ablation 64: workgroups start 0 .. 3 quarters of ~16k cycles apart (are the store bursts of lock-step tiles the cost?)
    if (p.abl & 64u) {
      const long long t0 = (long long)__builtin_readcyclecounter(), dl = (long long)((blockIdx.x >> 3) & 3) * 4000;
      while ((long long)__builtin_readcyclecounter() - t0 < dl) __builtin_amdgcn_s_sleep(8);
      __syncthreads();
    }
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = p.L;
  // Stamped build only: work REMOVED for the power / clock ablations of round 5 (results are wrong; the stamps give the effective clock):
  //   1 producers: no global loads behind the first    2 producers: no publish / transform behind stage 2 (barriers only)
  //   4 consumers: no epilogue    8 consumers: every weight request re-reads the image's first 4 KB (no L2 -> CU weight traffic)
  //   16 the epilogue without its stores    32 the epilogue without its residual loads    64 staggered starts (round 6: what does the epilogue cost beyond its own cycles?)
  const unsigned abl = DBG ? p.abl : 0u;
  const int ntiles_all = vend - first;
  const int nch = p.nchunks;                               // 32-channel chunks
  const int nst = nch * HALVES / CPS;                      // stages per tile
  const int my_tiles = (vend - v0 + stride - 1) / stride;
  const int nstages = my_tiles * nst;
  // tile v -> (first window w0 of the tile, batch element, row block)
  auto locate = [&](int v, int& w0_, int& bz_, int& by_) {
    const int tl = xcd_linear(v - first, ntiles_all, p.xcd);
    const int t = tl / p.ntn;
    bz_ = t / p.gy;
    w0_ = (tl - t * p.ntn) * NWT;
    by_ = t - bz_ * p.gy;
  };
  // raw tile of the tile that starts at window w0: first column xs (multiple of 4) and, for D > 1, the phase of window w0 and
  // the raw index `lead` of its first sample
  auto origin = [&](int w0_, int& xs_, int& ph0_, int& lead_) {
    if constexpr (D == 1) { xs_ = 4 * w0_ + XOFF; ph0_ = 0; lead_ = LEAD; }
    else {
      const int b0 = w0_ / D;
      ph0_ = w0_ - b0 * D;
      const int f0 = 4 * D * b0 + ph0_ - PADT * D;
      xs_ = f0 & ~3;                                       // two's complement: rounds towards minus infinity
      lead_ = f0 - xs_;
    }
  };

  if (wave >= 4) {
    // ================================================================= producer
    const int pw_ = wave - 4;
    // Producers outrank the consumers for issue: fp32 MFMAs and VALU instructions exclude each other on a SIMD
    // (tools/mfma_valu_probe.hip), and the older consumer wave wins the arbitration by age, so at equal priority the
    // producer only advances in the gaps of the MFMA stream and the consumers then wait for it at the stage barrier.
    // Measured (profiles/r03_i_f43_producer_priority.txt): barrier waits of the consumers halve, k = 11 tile 120.7k -> 118.6k cycles,
    // 16 x 512 step 33.2 -> 33.0 ms.
    __builtin_amdgcn_s_setprio(3);
    constexpr int R4 = RAW / 4, NGW = RPW * R4, SPW = (NGW + 63) / 64;      // producer p owns channel rows RPW p .. RPW p + RPW - 1
    constexpr int IPR = NE, NIW = RPW * IPR, TPW = (NIW + 63) / 64;        // transform items: one window each
    const long long ldb = (long long)p.x_ld * 4;
    const float slope = p.pre_slope;
    unsigned goff[SPW];
    float* rdst[SPW];
#pragma unroll
    for (int u = 0; u < SPW; ++u) {
      const int it = min(lane + 64 * u, NGW - 1);
      const int row = RPW * pw_ + it / R4, g4 = it % R4;
      goff[u] = (unsigned)(row * p.x_ld + 4 * g4) * 4u;
      rdst[u] = raw + row * RAWS + PORG + 4 * g4;
    }
    // window-major input: group g of a row = window P b_first + g of the source row, scattered to columns 4 P (g / P) + g % P + r P
    constexpr int NGWP = PERM > 0 ? RPW * Geo::PNG : 1, SPWP = PERM > 0 ? (NGWP + 63) / 64 : 1;
    unsigned pgoff[SPWP];
    float* pdst[SPWP];
    int pgb[SPWP];                                         // edge tiles: block of the group relative to the first loaded one | its phase << 16
    if constexpr (PERM > 0) {
#pragma unroll
      for (int u = 0; u < SPWP; ++u) {
        const int it = min(lane + 64 * u, NGWP - 1);
        const int row = RPW * pw_ + it / Geo::PNG, g = it % Geo::PNG;
        pgoff[u] = (unsigned)(row * p.x_ld + 4 * g) * 4u;
        pdst[u] = raw + row * RAWS + PORG + 4 * PERM * (g / PERM) + g % PERM;
        pgb[u] = (g / PERM) | ((g % PERM) << 16) | (row << 20);
      }
    }
    const int pnblk_row = PERM > 0 ? (L + 4 * PERM - 1) / (4 * (PERM > 0 ? PERM : 1)) : 0;      // q blocks of a row
    // first q block a tile loads: the one that holds column xs (xs >= -4 PERM: the left halo is at most 8 columns)
    auto pfirst = [&](int xs_) -> int { constexpr int P4 = PERM > 0 ? 4 * PERM : 1; return (xs_ + P4) / P4 - 1; };      // (only called with PERM > 0)
    int tsrc[TPW];                                         // float offset inside `raw`: D = 1: the item's samples; D > 1: the item's raw row (the column follows the tile)
    int tdst[TPW];                                         // float offset of the item's entry inside plane 0 of a set
    int tent[TPW];                                         // D > 1: the item's entry index e
    int toff[TPW];                                         // D > 1: raw column of the item's d0 in the current tile
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      const int it = min(lane + 64 * u, NIW - 1);
      const int row = RPW * pw_ + it / IPR, e = it % IPR;
      // D = 1: the sixteen-byte group that holds d1..d4 (LEAD = 3) or d0..d2 | d3..d5 (LEAD = 1) of the window starts at 4 e (+ 4)
      tsrc[u] = D == 1 ? row * RAWS + PORG + 4 * e : row * RAWS;
      tdst[u] = row * PQ + e;
      tent[u] = e; toff[u] = 0;
      // opaque from here on: left to itself the compiler RE-COMPUTES these per-lane constants (the divisions by the entries per row
      // included) at every stage - ~40 VALU instructions per stage beside the transform's 70, each of which costs the SIMD's matrix
      // pipe its issue time - instead of keeping them in two registers per item
      asm volatile("" : "+v"(tsrc[u]), "+v"(tdst[u]));
    }
    // D > 1: entry e is window w0 + e = (q block b0 + (ph0 + e) / D, phase (ph0 + e) % D): its d0 sits 4 D qe + pe - ph0 samples
    // behind the tile's first sample
    auto retarget = [&](int ph0_, int lead_) {
      if constexpr (D > 1) {
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
          const int t = ph0_ + tent[u];
          const int qe = t / D, pe = t - qe * D;
          toff[u] = lead_ + 4 * D * qe + pe - ph0_;
        }
      }
    };
    float4 v[PERM > 0 ? (SPWP > SPW ? SPWP : SPW) : SPW];
    int w0 = 0, bz = 0, by = 0, xs0 = 0, ph0 = 0, lead0 = 0;
    auto issue = [&](const char* xb_, int xs_, bool interior_, int ch) {
      const char* cb = xb_ + (long long)ch * KS * ldb;
      if constexpr (PERM > 0) {
        if (interior_) {                                   // whole q blocks from the one that holds column xs
          const int so = ch * KS * (int)ldb + (xs_ / (4 * PERM)) * (16 * PERM);
#pragma unroll
          for (int u = 0; u < SPWP; ++u) v[u] = w4_load16(xb_, pgoff[u], so);
        } else {                                           // edge tile: the same groups from blocks clamped into the row (zeroed by publish)
          const int bf = pfirst(xs_);
#pragma unroll
          for (int u = 0; u < SPWP; ++u) {
            const int bb = min(max(bf + (pgb[u] & 0xffff), 0), pnblk_row - 1), ph_ = (pgb[u] >> 16) & 15, row = pgb[u] >> 20;
            v[u] = *reinterpret_cast<const float4*>(cb + (long long)row * ldb + (long long)(PERM * bb + ph_) * 16);
          }
        }
      } else if (interior_) {
        const int so = ch * KS * (int)ldb + xs_ * 4;
#pragma unroll
        for (int u = 0; u < SPW; ++u) v[u] = w4_load16(xb_, goff[u], so);
      } else {
        int l_ = lane;
        asm volatile("" : "+v"(l_));
#pragma unroll
        for (int u = 0; u < SPW; ++u) {
          const int it = min(l_ + 64 * u, NGW - 1);
          const int row = RPW * pw_ + it / R4, tg = xs_ + 4 * (it % R4);
          v[u] = *reinterpret_cast<const float4*>(cb + (long long)row * ldb + (long long)((tg >= 0 && tg + 3 < L) ? tg : 0) * 4);
        }
      }
    };
    locate(v0, w0, bz, by);
    origin(w0, xs0, ph0, lead0);
    retarget(ph0, lead0);
    issue(reinterpret_cast<const char*>(p.x + (long long)bz * p.x_bs), xs0, xs0 >= 0 && xs0 + RAW <= L, 0);
    int ti = 0, ch = 0;                                    // stage s = (tile ti, stage ch of the tile)
    int pset = 0;
    long long pc_all0 = 0, pc_bar = 0;
    if constexpr (DBG) pc_all0 = (long long)__builtin_readcyclecounter();
    for (int s_ = 0; s_ < nstages; ++s_) {
      const int xs_start = xs0;
      const bool interior = xs_start >= 0 && xs_start + RAW <= L;
      const char* const xb = reinterpret_cast<const char*>(p.x + (long long)bz * p.x_bs);
      if (!(DBG && (abl & 2u) && s_ > 2)) {
      // ---- publish own rows (lrelu, zero padding on edge tiles)
      if constexpr (PERM > 0) {
        if (interior) {
          const int delta = (xs_start / (4 * PERM)) * (4 * PERM) - xs_start;      // column of the first loaded block relative to xs
#pragma unroll
          for (int u = 0; u < SPWP; ++u) {
            if (64 * (u + 1) <= NGWP || lane < NGWP - 64 * u) {
              float4 q = v[u];
              wino_lrelu4(q, slope);
              float* d = pdst[u] + delta;
              d[0] = q.x; d[PERM] = q.y; d[2 * PERM] = q.z; d[3 * PERM] = q.w;
            }
          }
        } else {                                           // edge tile: groups of blocks outside the row, and samples beyond L, are zero
          const int bf = pfirst(xs_start);
          const int delta = bf * (4 * PERM) - xs_start;
#pragma unroll
          for (int u = 0; u < SPWP; ++u) {
            if (64 * (u + 1) <= NGWP || lane < NGWP - 64 * u) {
              const int bb = bf + (pgb[u] & 0xffff), ph_ = (pgb[u] >> 16) & 15;
              const int n0 = 4 * PERM * bb + ph_;          // natural column of the group's first sample
              const bool inrow = bb >= 0 && bb < pnblk_row;
              float4 q = v[u];
              q.x = (inrow && n0 < L) ? q.x : 0.f;
              q.y = (inrow && n0 + PERM < L) ? q.y : 0.f;
              q.z = (inrow && n0 + 2 * PERM < L) ? q.z : 0.f;
              q.w = (inrow && n0 + 3 * PERM < L) ? q.w : 0.f;
              wino_lrelu4(q, slope);
              float* d = pdst[u] + delta;
              d[0] = q.x; d[PERM] = q.y; d[2 * PERM] = q.z; d[3 * PERM] = q.w;
            }
          }
        }
      } else if (interior) {
#pragma unroll
        for (int u = 0; u < SPW; ++u) {
          if (64 * (u + 1) <= NGW || lane < NGW - 64 * u) {
            float4 q = v[u];
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
            if (tg + 3 < 0 || tg >= L) q = make_float4(0.f, 0.f, 0.f, 0.f);      // the whole group is padding (xs is a multiple of 4)
            else if (!(tg >= 0 && tg + 3 < L)) {             // a group that straddles the end: only when L is not a multiple of 4
              const float* xr = reinterpret_cast<const float*>(xb + (long long)(ch * KS + row) * ldb);
              q.x = (tg >= 0 && tg < L) ? xr[tg] : 0.f;
              q.y = (tg + 1 >= 0 && tg + 1 < L) ? xr[tg + 1] : 0.f;
              q.z = (tg + 2 >= 0 && tg + 2 < L) ? xr[tg + 2] : 0.f;
              q.w = (tg + 3 >= 0 && tg + 3 < L) ? xr[tg + 3] : 0.f;
            }
            wino_lrelu4(q, slope);
            *reinterpret_cast<float4*>(rdst[u]) = q;
          }
        }
      }
      }
      // ---- request the next stage's raw rows (next chunk, or chunk 0 of this workgroup's next tile)
      int nti = ti, nchn = ch + 1, w0n = w0, bzn = bz, byn = by, xsn = xs0, ph0n = ph0, leadn = lead0;
      if (nchn == nst) { nchn = 0; ++nti; if (nti < my_tiles) { locate(v0 + nti * stride, w0n, bzn, byn); origin(w0n, xsn, ph0n, leadn); } }
      if (s_ + 1 < nstages && !(abl & 1u)) issue(reinterpret_cast<const char*>(p.x + (long long)bzn * p.x_bs), xsn, xsn >= 0 && xsn + RAW <= L, nchn);
      // ---- transform own rows into plane set s & 1 (LDS operations of one wave execute in order: no barrier needed)
      float* const pb = pl + (NPS == 2 ? (s_ & 1) : pset) * PLF;
      if constexpr (NPS == 3) pset = pset == 2 ? 0 : pset + 1;
      if (!(DBG && (abl & 2u) && s_ > 2)) {
#pragma unroll
      for (int u = 0; u < TPW; ++u) {
        if (64 * (u + 1) <= NIW || lane < NIW - 64 * u) {
          w4_transform_window<Geo>(pb + tdst[u], D == 1 ? raw + tsrc[u] : raw + tsrc[u] + toff[u]);
        }
      }
      }
      long long pb0 = 0;
      if constexpr (DBG) pb0 = (long long)__builtin_readcyclecounter();
      // B_s: plane set s & 1 complete.  Three sets: the barrier behind stage s is B_{s-1} (the consumers then start stage s - 1
      // while stage s + 1 is produced into the set they left before B_{s-1}); B_{last} follows the loop
      if (NPS == 2 || s_ > 0) __syncthreads();
      if constexpr (DBG) pc_bar += (long long)__builtin_readcyclecounter() - pb0;
      if (nti != ti) retarget(ph0n, leadn);                // the next stage belongs to another tile
      ti = nti; ch = nchn; w0 = w0n; bz = bzn; by = byn; xs0 = xsn; ph0 = ph0n; lead0 = leadn;
    }
    if constexpr (NPS == 3) __syncthreads();
    if constexpr (DBG) if (tid == 256) {                   // producer wave 0: total cycles, cycles spent waiting at the stage barriers
      long long* d = p.dbg + 16 * (long long)(p.dbg_base + v0);
      d[8] = (long long)__builtin_readcyclecounter() - pc_all0; d[9] = pc_bar;
    }
    return;
  }

  // =================================================================== consumer: row tile rt, column tile ct of the workgroup
  const int l31 = lane & 31, hi = lane >> 5;
  const int rt = NRT == 4 ? wave : (NRT == 2 ? (wave & 1) : 0), ct = NRT == 4 ? 0 : (NRT == 2 ? (wave >> 1) : wave);
  const int uu = ct * 32 + l31;                            // this lane's window inside the workgroup tile
  const unsigned pbase = (unsigned)(size_t)pl;
  const unsigned baddr0 = pbase + (unsigned)(hi * PQ + uu) * 4u;
  const unsigned wlane = (unsigned)lane * 16u;
  f32x16 M[NACC];
  float4 a[NSET][KGS];
  const int wstep = (DBG && (abl & 8u)) ? 0 : 4096;        // ablation 8: the slot stride of the weight requests (0: the same 4 KB over and over, from the L1)
  // MFMA stream of one 32-channel chunk (chunk CC of the stage: plane rows 32 CC ..) or of half HF of a chunk (KS = 16): NSTEP steps of four MFMAs, fragment
  // reads two steps ahead in two register sets, the next weight slot's four float4 requested at the first step of each slot
  // Weights stream through buffer loads: descriptor base = packed image, SGPR offset = (row tile, chunk, slot), VGPR offset =
  // lane * 16, immediate = k-group.  The per-slot address arithmetic is then SALU only: a VALU instruction in the consumer
  // stream costs the matrix pipe its issue time AND breaks the back-to-back forwarding of dependent MFMAs.
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wp), 0, 0x7fffffff, 0x00020000);
  auto wload = [&](float4& dst, int soff, auto kg_c) {
    constexpr int KGO = decltype(kg_c)::value * 1024;
    const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)wlane + KGO, soff, 0);
    dst = *reinterpret_cast<const float4*>(&t);
  };
  auto wload4 = [&](float4 (&dst)[KGS], int soff, auto hf) {       // the KGS k-groups of half HF of a slot
    constexpr int K0 = decltype(hf)::value * KGS;
    wload(dst[0], soff, std::integral_constant<int, K0>{});
    if constexpr (KGS >= 2) wload(dst[1], soff, std::integral_constant<int, K0 + 1>{});
    if constexpr (KGS == 4) { wload(dst[2], soff, std::integral_constant<int, K0 + 2>{}); wload(dst[3], soff, std::integral_constant<int, K0 + 3>{}); }
  };
  // wa / wnext: byte offsets of the chunk's / the following chunk's slot 0 inside the image (wnext < 0: none)
  // wnext belongs to part NHF (the next part of the same chunk while HF + 1 < HALVES, else part 0 of the next chunk)
  auto mfma_chunk = [&](const unsigned baddr, const int wa, const int wnext, auto par, auto cc, auto hf) {
    constexpr int PAR = decltype(par)::value, CC = decltype(cc)::value, HF = decltype(hf)::value;
    constexpr int NHF = HF + 1 < HALVES ? HF + 1 : 0;
    float fb[2][4];
    auto request = [&](auto tc) {
      constexpr int T = decltype(tc)::value;
      if constexpr (T < NSTEP) wino_frag<PQ, Geo::plane(T) * PLANE + CC * KC * PQ, Geo::kgi(T), Geo::colq(T) * D>(fb[T & 1], baddr);
    };
    auto step = [&](auto tc) {
      constexpr int T = decltype(tc)::value;
      constexpr int WS = Geo::wslot(T), KG = Geo::kgi(T);
      if constexpr (Geo::slot_first(T)) {                 // request slot WS + PD (of this stage, else of the stage / tile that follows)
        if constexpr (DBG) {                                // ablation 8, branch-free (a branch per slot stretched the stream from 68 to 85 cycles per MFMA): every request re-reads slot 0
          if constexpr (WS + PD < WSLOTS) wload4(a[(PAR + WS + PD) % NSET], wa + (WS + PD) * wstep, hf);
          else if (wnext >= 0) wload4(a[(PAR + WS + PD) % NSET], wnext + (WS + PD - WSLOTS) * wstep, std::integral_constant<int, NHF>{});
        } else {
          if constexpr (WS + PD < WSLOTS) wload4(a[(PAR + WS + PD) % NSET], wa + (WS + PD) * 4096, hf);
          else if (wnext >= 0) wload4(a[(PAR + WS + PD) % NSET], wnext + (WS + PD - WSLOTS) * 4096, std::integral_constant<int, NHF>{});
        }
      }
      {
        float(&b)[4] = fb[T & 1];
        if constexpr (T + 1 < NSTEP) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
      }
      const float4 av = a[(PAR + WS) % NSET][KG];
      constexpr int AC = Geo::acc(T);
#pragma unroll
      for (int s = 0; s < 4; ++s) M[AC] = __builtin_amdgcn_mfma_f32_32x32x2f32(wino_pick(av, s), fb[T & 1][s], M[AC], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      request(std::integral_constant<int, T + 2>{});
      __builtin_amdgcn_sched_barrier(0);
    };
    request(std::integral_constant<int, 0>{});
    request(std::integral_constant<int, 1>{});
    wino_static_for<0, NSTEP>(step);
  };
  auto wtile = [&](int mt_) -> int { return (DBG && (abl & 8u)) ? 0 : __builtin_amdgcn_readfirstlane(mt_ * nch * WSLOTS * 4096); };
  // byte offsets of this lane's first output (then + D, + 2D, + 3D) in rows 4 hi + i of a 32-row block.  D = 1: relative to the
  // tile's first column (fixed); D > 1: absolute column of window w0 + uu, recomputed per tile
  const unsigned ylb = (unsigned)p.y_ld * 4u, rlb = (unsigned)p.res_ld * 4u;
  unsigned yo4[4], ro4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    yo4[i] = (unsigned)((4 * hi + i) * p.y_ld + 4 * uu) * 4u;
    ro4[i] = (unsigned)((4 * hi + i) * p.res_ld + 4 * uu) * 4u;
  }
  // fast epilogue: residual but no read-modify-write of y, and every lane's four outputs exist (D = 1: L % 4 == 0, host;
  // D > 1: the tile lies inside [0, L), decided per tile)
  const bool res_flags = (p.flags & F_RES) && !(p.flags & F_ACC);
  auto ldq = [&](const char* q) -> float4 {                // the lane's four outputs of one row: 16 bytes (D = 1) or four dwords D apart
    if constexpr (D == 1) return *reinterpret_cast<const float4*>(q);
    else return make_float4(*reinterpret_cast<const float*>(q), *reinterpret_cast<const float*>(q + 4 * D),
                            *reinterpret_cast<const float*>(q + 8 * D), *reinterpret_cast<const float*>(q + 12 * D));
  };
  auto stq = [&](char* q, const float4& v) {
    if constexpr (D == 1) *reinterpret_cast<float4*>(q) = v;
    else {
      *reinterpret_cast<float*>(q) = v.x; *reinterpret_cast<float*>(q + 4 * D) = v.y;
      *reinterpret_cast<float*>(q + 8 * D) = v.z; *reinterpret_cast<float*>(q + 12 * D) = v.w;
    }
  };
  int cset = 0;                                            // three plane sets: set of the next stage
  long long cyc_bar = 0, cyc_mf = 0, cyc_epi = 0, cyc_all0 = 0, cyc_drain = 0;     // diagnostics (stamped build): consumer wave 0 of each workgroup
  long long wall0 = 0;
  if constexpr (DBG) { cyc_all0 = (long long)__builtin_readcyclecounter(); wall0 = (long long)wall_clock64(); }
  for (int ti = 0; ti < my_tiles; ++ti) {
    int w0, bz, by;
    locate(v0 + ti * stride, w0, bz, by);
    int ne, n0;                                            // this lane's first output column; column the row bases point at
    bool tile_full;                                        // every lane of the tile has its four outputs inside [0, L)
    if constexpr (D == 1) { n0 = 4 * w0; ne = n0 + 4 * uu; tile_full = true; }
    else {
      const int w = w0 + uu, b = w / D, ph = w - b * D;
      ne = 4 * D * b + ph; n0 = 0;
      tile_full = 4 * D * ((w0 + NWT - 1) / D + 1) <= L;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        yo4[i] = (unsigned)((4 * hi + i) * p.y_ld + ne) * 4u;
        ro4[i] = (unsigned)((4 * hi + i) * p.res_ld + ne) * 4u;
      }
    }
    const int mt = by * NRT + rt;
    const bool row_ok = mt < p.mtiles;
    const int mtc = row_ok ? mt : p.mtiles - 1;
    const int wt = wtile(mtc);
    if (ti == 0) {
      wload4(a[0], wt, std::integral_constant<int, 0>{});
      if constexpr (PD > 1) { wload4(a[1], wt + 4096, std::integral_constant<int, 0>{}); wload4(a[2], wt + 2 * 4096, std::integral_constant<int, 0>{}); }
    }
    {   // the bias starts in M1: y0 and y2 contain M1 + M2, y1 and y3 contain M1 - M2, so all four outputs receive it once
      const float* bias = p.bias + mtc * 32 + 4 * hi;
#pragma unroll
      for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) M[q][i] = q == 1 ? bias[(i & 3) + 8 * (i >> 2)] : 0.f;
    }
    int wnext_tile = -1;
    if (ti + 1 < my_tiles) {
      int n0n, bzn, byn;
      locate(v0 + (ti + 1) * stride, n0n, bzn, byn);
      const int mtn = byn * NRT + rt;
      wnext_tile = wtile(mtn < p.mtiles ? mtn : p.mtiles - 1);
    }
    const bool lane_ok = row_ok && ne < L;
    const bool res_only = res_flags && tile_full;
    // (ablation 256, stamped build: every tile's stores go to the first columns of batch element 0 - the same 64 KB per row block, L2-resident)
    char* const ybase = (DBG && (abl & 256u)) ? reinterpret_cast<char*>(p.y + (long long)(mt * 32) * p.y_ld)
                                              : reinterpret_cast<char*>(p.y + (long long)bz * p.y_bs + (long long)(mt * 32) * p.y_ld + n0);
    const char* const rbase = reinterpret_cast<const char*>(p.res + (long long)bz * p.res_bs + (long long)(mt * 32) * p.res_ld + n0);
    float4 rvA[8];                                         // residual of accumulator rows 0..7, requested under the tile's last stage
    auto stage = [&](int st_, auto par, auto hf) {
      constexpr int HF = decltype(hf)::value;
      const int s_ = ti * nst + st_;
      long long c0 = 0, c1 = 0;
      if constexpr (DBG) c0 = (long long)__builtin_readcyclecounter();
      __syncthreads();                                     // B_s: plane set s & 1 is complete, set (s - 1) & 1 may be overwritten
      if constexpr (DBG) c1 = (long long)__builtin_readcyclecounter();
      if (st_ == nst - 1 && res_only && lane_ok) {
        if (DBG && (abl & 32u)) {
#pragma unroll
          for (int r = 0; r < 8; ++r) rvA[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else
#pragma unroll
        for (int r = 0; r < 8; ++r) rvA[r] = ldq(rbase + (size_t)(8 * (r >> 2)) * rlb + ro4[r & 3]);
      }
      const unsigned off = (unsigned)((NPS == 2 ? (s_ & 1) : cset) * PLF) * 4u;
      if constexpr (NPS == 3) cset = cset == 2 ? 0 : cset + 1;
      constexpr auto c0_ = std::integral_constant<int, 0>{};
      if constexpr (HALVES > 1) {                          // a part of a chunk per stage: the same slots again for the next part
        const int wa = wt + (st_ / HALVES) * WSLOTS * 4096;
        const int wnext = HF + 1 < HALVES ? wa : (st_ + 1 < nst ? wa + WSLOTS * 4096 : wnext_tile);
        mfma_chunk(baddr0 + off, wa, wnext, par, c0_, hf);
      } else if constexpr (CPS == 1) {
        const int wa = wt + st_ * WSLOTS * 4096;
        const int wnext = st_ + 1 < nst ? wa + WSLOTS * 4096 : wnext_tile;
        mfma_chunk(baddr0 + off, wa, wnext, par, c0_, c0_);
      } else {
        static_assert(CPS == 1 || (WSLOTS & 1) == 0, "two chunks per stage need an even slot count");
        const int wa = wt + (st_ * CPS) * WSLOTS * 4096;
        mfma_chunk(baddr0 + off, wa, wa + WSLOTS * 4096, par, c0_, c0_);
        const int wnext = st_ + 1 < nst ? wa + 2 * WSLOTS * 4096 : wnext_tile;
        mfma_chunk(baddr0 + off, wa + WSLOTS * 4096, wnext, par, std::integral_constant<int, CPS - 1>{}, c0_);
      }
      if constexpr (DBG) { cyc_bar += c1 - c0; cyc_mf += (long long)__builtin_readcyclecounter() - c1; }
    };
    // the register set of a stage's first slot advances by the stage's slot count (mod NSET): the stage loop is unrolled over U
    // stages, U * WSLOTS a multiple of NSET, so that every round (and every tile: nst is a multiple of U, host) starts in set 0
    if constexpr (HALVES > 1) {
      constexpr int U = (HALVES * WSLOTS) % NSET == 0 ? HALVES : 2 * HALVES;
      static_assert((U * WSLOTS) % NSET == 0 && U <= 4, "weight ring does not close");
      for (int st_ = 0; st_ < nst; st_ += U) {
        stage(st_, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        stage(st_ + 1, std::integral_constant<int, WSLOTS % NSET>{}, std::integral_constant<int, 1 % HALVES>{});
        if constexpr (U == 4) {
          stage(st_ + 2, std::integral_constant<int, (2 * WSLOTS) % NSET>{}, std::integral_constant<int, 2 % HALVES>{});
          stage(st_ + 3, std::integral_constant<int, (3 * WSLOTS) % NSET>{}, std::integral_constant<int, 3 % HALVES>{});
        }
      }
    } else if constexpr ((WSLOTS & 1) == 0) {
      for (int st_ = 0; st_ < nst; ++st_) stage(st_, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    } else {
      for (int st_ = 0; st_ < nst; st_ += 2) {
        stage(st_, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        stage(st_ + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
      }
    }
    // ---- output transform + epilogue: the lane owns y[row][ne + r D], r = 0..3, for its 16 accumulator rows, four at a time
    long long ce0 = 0;
    if constexpr (DBG) ce0 = (long long)__builtin_readcyclecounter();
    if (lane_ok && !(DBG && (abl & 4u))) {
      auto ytrans = [&](auto q_c, float4 (&vo)[4]) {
        constexpr int Q = decltype(q_c)::value;
#pragma unroll
        for (int r = 0; r < 4; r += 2) w4_output_transform2<Geo, NACC>(M, 4 * Q + r, vo[r], vo[r + 1]);
      };
      auto divide = [&](float4 (&vo)[4]) {                 // x / div as x * (1 / div) with one residual correction (as conv_wino.hip)
        const float dv = p.div, rc = 1.0f / dv;
        auto dv1 = [&](float x) { const float q = x * rc; return __builtin_fmaf(__builtin_fmaf(-q, dv, x), rc, q); };
#pragma unroll
        for (int r = 0; r < 4; ++r) vo[r] = make_float4(dv1(vo[r].x), dv1(vo[r].y), dv1(vo[r].z), dv1(vo[r].w));
      };
      if constexpr (OPERM) {
        // window-major store (round 4): the lane's four outputs n, n + D, n + 2D, n + 3D go to P[4 w .. 4 w + 3] of the row, w = w0 + uu:
        // one 16-byte store per row instead of four scattered dwords; the undilated convolution that follows reads through the
        // same map (W4Geo::PERM).  Plain epilogue only (a c1 has no residual): host.
        char* const yp = reinterpret_cast<char*>(p.y + (long long)bz * p.y_bs + (long long)(mt * 32) * p.y_ld) + (size_t)(w0 + uu) * 16;
        auto quarter = [&](auto q_c) {
          constexpr int Q = decltype(q_c)::value;
          float4 vo[4];
          ytrans(q_c, vo);
#pragma unroll
          for (int r = 0; r < 4; ++r) *reinterpret_cast<float4*>(yp + (size_t)(8 * Q + 4 * hi + r) * ylb) = vo[r];
        };
        quarter(std::integral_constant<int, 0>{});
        quarter(std::integral_constant<int, 1>{});
        quarter(std::integral_constant<int, 2>{});
        quarter(std::integral_constant<int, 3>{});
      } else if (res_only) {
        {
          auto quarter = [&](auto q_c, const float4* rv) {
            constexpr int Q = decltype(q_c)::value;
            float4 vo[4];
            ytrans(q_c, vo);
#pragma unroll
            for (int r = 0; r < 4; ++r) w4_add4(vo[r], rv[r]);
            if (p.flags & F_DIV) divide(vo);
            if (DBG && ((abl & 16u) || ((abl & 512u) && Q > 0))) {      // ablation 16: the epilogue without its stores (the branch is never taken); 512: the first quarter's only
#pragma unroll
              for (int r = 0; r < 4; ++r) if (vo[r].x == 1.2345678e38f) stq(ybase + (size_t)(8 * Q) * ylb + yo4[r], vo[r]);
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) stq(ybase + (size_t)(8 * Q) * ylb + yo4[r], vo[r]);
            }
          };
          float4 rvB[8];                                   // (experiment: requested AHEAD of the first quarter)
          if (DBG && (abl & 32u)) {                        // ablation 32: the epilogue without its residual loads
#pragma unroll
            for (int r = 0; r < 8; ++r) rvB[r] = make_float4(0.f, 0.f, 0.f, 0.f);
          } else
#pragma unroll
          for (int r = 0; r < 8; ++r) rvB[r] = ldq(rbase + (size_t)(8 * (2 + (r >> 2))) * rlb + ro4[r & 3]);
          quarter(std::integral_constant<int, 0>{}, rvA);
          quarter(std::integral_constant<int, 1>{}, rvA + 4);
          quarter(std::integral_constant<int, 2>{}, rvB);
          quarter(std::integral_constant<int, 3>{}, rvB + 4);
        }
      } else {
        const int nval = D == 1 ? 4 : min(4, (L - ne + D - 1) / D);      // outputs of this lane inside [0, L)
        auto ld4 = [&](const char* q) -> float4 {
          if constexpr (D == 1) return *reinterpret_cast<const float4*>(q);
          else {
            float4 t = make_float4(*reinterpret_cast<const float*>(q), 0.f, 0.f, 0.f);
            if (nval > 1) t.y = *reinterpret_cast<const float*>(q + 4 * D);
            if (nval > 2) t.z = *reinterpret_cast<const float*>(q + 8 * D);
            if (nval > 3) t.w = *reinterpret_cast<const float*>(q + 12 * D);
            return t;
          }
        };
        auto quarter = [&](auto q_c) {
          constexpr int Q = decltype(q_c)::value;
          float4 vo[4];
          ytrans(q_c, vo);
          if (p.flags & F_RES) {
            float4 rv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) rv[r] = ld4(rbase + (size_t)(8 * Q) * rlb + ro4[r]);
#pragma unroll
            for (int r = 0; r < 4; ++r) w4_add4(vo[r], rv[r]);
          }
          if (p.flags & F_ACC) {
            float4 yv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) yv[r] = ld4(ybase + (size_t)(8 * Q) * ylb + yo4[r]);
#pragma unroll
            for (int r = 0; r < 4; ++r) { vo[r].x = yv[r].x + vo[r].x; vo[r].y = yv[r].y + vo[r].y; vo[r].z = yv[r].z + vo[r].z; vo[r].w = yv[r].w + vo[r].w; }
          }
          if (p.flags & F_DIV) divide(vo);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            char* q = ybase + (size_t)(8 * Q) * ylb + yo4[r];
            if constexpr (D == 1) *reinterpret_cast<float4*>(q) = vo[r];
            else {
              *reinterpret_cast<float*>(q) = vo[r].x;
              if (nval > 1) *reinterpret_cast<float*>(q + 4 * D) = vo[r].y;
              if (nval > 2) *reinterpret_cast<float*>(q + 8 * D) = vo[r].z;
              if (nval > 3) *reinterpret_cast<float*>(q + 12 * D) = vo[r].w;
            }
          }
        };
        quarter(std::integral_constant<int, 0>{});
        quarter(std::integral_constant<int, 1>{});
        quarter(std::integral_constant<int, 2>{});
        quarter(std::integral_constant<int, 3>{});
      }
    }
    if constexpr (DBG) cyc_epi += (long long)__builtin_readcyclecounter() - ce0;
    if constexpr (DBG) if (abl & 128u) {                   // ablation 128: drain the vector-memory queue behind the epilogue and time it: how long do the stores' acknowledgements take?
      const long long cd0 = (long long)__builtin_readcyclecounter();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      cyc_drain += (long long)__builtin_readcyclecounter() - cd0;
    }
  }
  if constexpr (DBG) if (tid == 0) {      // [workgroup][16]: 0 tiles, 1 total cycles, 2 barrier waits, 3 MFMA streams, 4 epilogues, 5 marker, 6 HW_ID, 7 XCC_ID
    long long* d = p.dbg + 16 * (long long)(p.dbg_base + v0);
    d[0] = my_tiles; d[1] = (long long)__builtin_readcyclecounter() - cyc_all0; d[2] = cyc_bar; d[3] = cyc_mf; d[4] = cyc_epi; d[5] = 4;
    d[6] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    d[7] = (long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    d[12] = cyc_drain;
    d[10] = wall0; d[11] = (long long)wall_clock64();     // 100 MHz constant clock: effective shader clock = d[1] / (d[11] - d[10]) * 100 MHz
  }
}

template <int K, int D, int NRT, bool DBG, bool F44 = false>
__global__ void __launch_bounds__(512, 2) conv_wino4_kernel(const WinoArgs p, const int total) {
  wino4_problem<K, D, NRT, DBG, 0, F44>(p, blockIdx.x, total, 0, gridDim.x);
}
template <int K, int D, int NRT, int PERM = 0, bool F44 = false>
__device__ __forceinline__ void wino4_member(const WinoArgs& p, const int first, const int vend, const int b, const int G_) {
  if (vend <= first) return;
  int v0 = b - first % G_;
  if (v0 < 0) v0 += G_;
  wino4_problem<K, D, NRT, false, PERM, F44>(p, v0 + first, vend, first, G_);
}
// the MRF chains' three convolutions of one step (k = 11 / 7 / 3) back to back in one persistent launch; PERM: their inputs are
// in the window-major order of a dilation-PERM predecessor (D = 1 only); F44: the k = 11 and k = 7 members in F(4,4) form (the k = 3
// member stays F(4,3): six products against seven)
template <int D, int NRT, int PERM = 0, bool F44 = false>
__global__ void __launch_bounds__(512, 2) conv_wino4_group_kernel(const WinoGroup g) {
  const int b = blockIdx.x, G_ = gridDim.x;
  // (round 6, measured and not kept - profiles/r06_epilogue_store_cost_study.txt: the XCDs starting on different members, so that the k = 3 phase's write
  // traffic is spread over the launch: 25.59 against 25.57 ms; the bias from an LDS copy instead of 16 global loads per tile: -350 cycles of a tile's
  // set-up in the stamps, nothing in the step; non-temporal stores: a launch between profiler events 1 - 2 % faster, the replayed step unchanged)
  wino4_member<11, D, NRT, PERM, F44>(g.a[0], 0, g.end[0], b, G_);
  __syncthreads();
  wino4_member<7, D, NRT, PERM, F44>(g.a[1], g.end[0], g.end[1], b, G_);
  __syncthreads();
  wino4_member<3, D, NRT, PERM>(g.a[2], g.end[1], g.end[2], b, G_);
}

}  // namespace svoc
