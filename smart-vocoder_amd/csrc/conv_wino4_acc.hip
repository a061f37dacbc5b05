// The accumulate launch of an MRF stage in ONE set of accumulators (round 4).
//
// The last convolutions of the three chains of a stage sum into one tensor: x = (rb_0 + rb_1 + rb_2) / 3 with
// rb_j = c2_j(lrelu(t_j)) + cur_j (reference models.py:149-155, modules.py:210-223; k = 3 / 7 / 11, dilation 1).  As three
// members of conv_wino4_accum_kernel the second and third read-modify-write the output: 11 tensor passes (3 inputs, 3 residuals,
// 2 old outputs read; 3 outputs written) where 7 are needed, and at the 268 MB per tensor of the C = 128 / 64 / 32 stages those
// launches are paced by exactly that traffic (+170 / +180 / +200 us against a plain grouped launch; the C = 256 stage, whose 67 MB
// tensors stay in the Infinity Cache, pays +26 us; profiles/r04_accumulate_*_null_*.txt: hiding the LATENCY of the old-output
// loads changes nothing).  The output transform of F(4,3) is linear, so the three convolutions can add their products into the
// SAME transform-domain accumulators: per tile the k = 3 stages, the k = 7 stages and the k = 11 stages run back to back (each
// with its own plane geometry and weight image, conv_wino4.h), the bias of M1 is the sum of the three biases, and ONE epilogue
// adds the three residuals, divides and stores.  The sum is formed in another order than the reference's (products of the three
// chains in one accumulator, then the residuals): fp32 rounding only (tests: the MRF goldens, Generator / infer against the oracle).
//
// Kernel form = conv_wino4.hip's: persistent eight-wave workgroups, ONE per CU, four consumers (nothing but the MFMA stream
// and the epilogue) and four producers one stage ahead in the other of two plane sets, one workgroup barrier per stage; the
// stage counter runs on across the members, the producers request the first raw rows of the next member at a member's last
// stage, the consumers the next member's first weight slots right behind a member's last MFMA.
#include "svoc_internal.h"
#include "wino_common.h"
#include "conv_wino4.h"
#include "conv_wino4_consume.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace svoc {

struct WinoAcc3 { WinoArgs a[3]; int total; };              // members k = 3, 7, 11 (chain order); one tile space; output / flags / div of a[0]

// F44: the three members in F(4,4) form (conv_wino4.h) - one set of SEVEN accumulators; the k = 3 member then issues seven products
// per window instead of six (its second weight image, PackedWino::wp44): 42 products per window against 48.  (One row tile: the k = 3
// member's seven slots of 16-channel stages would not close the four-set weight ring within its two stages; it stages 8 channels.)
template <int NRT, int PERM, bool F44 = false>
struct Acc3Geo {
  using G3 = W4Geo<3, 1, NRT, PERM, (F44 && NRT == 1) ? 2 : 1, F44>;
  using G7 = W4Geo<7, 1, NRT, PERM, 1, F44>;
  using G11 = W4Geo<11, 1, NRT, PERM, 1, F44>;
  static constexpr int NACC = F44 ? 7 : 8;
  static constexpr int cmax(int a, int b) { return a > b ? a : b; }
  static constexpr int RAWMAX = cmax(G3::RAW_FLOATS, cmax(G7::RAW_FLOATS, G11::RAW_FLOATS));
  static constexpr int PLFMAX = cmax(G3::PLF, cmax(G7::PLF, G11::PLF));
  // Two plane sets.  (Three - with the k = 3 member staged half as wide, W4Geo's KSDIV, so that they fit - measured SLOWER: accumulate
  // launches 1769 -> 1800 / 1108 -> 1137 / 795 -> 826 us at C = 128 / 64 / 32: the k = 3 stages get too short for their barriers.)
  // With the seven planes of the F(4,4) form three sets fit as they are: 750 / 1611 / 965 / 684 -> 754 / 1614 / 967 / 706 us, not taken either.
  static constexpr int NPS = 2;
  static constexpr int LDS_BYTES = (RAWMAX + NPS * PLFMAX) * 4;
  static_assert(G3::NWT == G7::NWT && G7::NWT == G11::NWT, "one tile space");
};

// ------------------------------------------------------------------------------------------------ producer side
// Per-lane staging constants of one member (recomputed at every member start: three sets would not fit the register file)
template <class Geo>
struct Acc3Prod {
  static constexpr int KS = Geo::KS, RPW = KS / 4, RAW = Geo::RAW, RAWS = Geo::RAWS, PORG = Geo::PORG, PERM = Geo::PORG / 4;
  static constexpr int R4 = RAW / 4, NGW = RPW * R4, SPW = (NGW + 63) / 64;
  static constexpr int NGWP = PERM > 0 ? RPW * Geo::PNG : 1, SPWP = PERM > 0 ? (NGWP + 63) / 64 : 1;
  static constexpr int NV = PERM > 0 ? (SPWP > SPW ? SPWP : SPW) : SPW;             // float4 registers of one stage's raw rows
  static constexpr int NIW = RPW * Geo::NE, TPW = (NIW + 63) / 64;
};

// requests the raw rows of stage `ch` of the tile whose raw tile starts at column xs (interior tiles: plain 16-byte groups, or
// whole q blocks of a window-major row; edge tiles: clamped addresses, zeroed by acc3_publish)
template <class Geo>
__device__ __forceinline__ void acc3_issue(float4 (&v)[Acc3Prod<Geo>::NV], const WinoArgs& p, const int bz, const int xs, const int ch,
                                           const int lane, const int pw_) {
  using P = Acc3Prod<Geo>;
  constexpr int PERM = P::PERM;
  const long long ldb = (long long)p.x_ld * 4;
  const int L = p.L;
  const bool interior = xs >= 0 && xs + P::RAW <= L;
  // 32-bit per-lane offsets against the batch element's base, the stage's rows as a scalar offset (w4_load16: no 64-bit VALU arithmetic)
  const float* const xb = p.x + (long long)bz * p.x_bs;
  const int so = ch * P::KS * (int)ldb;
  int l_ = lane;
  asm volatile("" : "+v"(l_));                             // keeps the per-lane address arithmetic inside the call (not hoisted and kept live)
  if constexpr (PERM > 0) {
    const int pnblk_row = (L + 4 * PERM - 1) / (4 * PERM);
    const int bf = (xs + 4 * PERM) / (4 * PERM) - 1;
#pragma unroll
    for (int u = 0; u < P::SPWP; ++u) {
      const int it = min(l_ + 64 * u, P::NGWP - 1);
      const int row = P::RPW * pw_ + it / Geo::PNG, g = it % Geo::PNG;
      int bb = bf + g / PERM;
      if (!interior) bb = min(max(bb, 0), pnblk_row - 1);
      v[u] = w4_load16(xb, (unsigned)(row * p.x_ld + 4 * (PERM * bb + g % PERM)) * 4u, so);
    }
  } else {
#pragma unroll
    for (int u = 0; u < P::SPW; ++u) {
      const int it = min(l_ + 64 * u, P::NGW - 1);
      const int row = P::RPW * pw_ + it / P::R4, tg = xs + 4 * (it % P::R4);
      v[u] = w4_load16(xb, (unsigned)(row * p.x_ld + ((interior || (tg >= 0 && tg + 3 < L)) ? tg : 0)) * 4u, so);
    }
  }
}

// One member of one tile on the producer side: publish (lrelu, padding) -> request the next stage -> transform -> barrier, per stage.
// `v` holds the raw rows of the member's stage 0 on entry (requested by the caller's previous member).  next(): requests the raw
// rows of whatever follows this member's last stage.
template <class Geo, int NPS, class Next>
__device__ __forceinline__ void acc3_produce(const WinoArgs& p, float4 (&v)[Acc3Prod<Geo>::NV], float* const raw, float* const pl, const int PLFMAX_,
                                            const int w0, const int bz, int& s_, const int lane, const int pw_, Next&& next) {
  using P = Acc3Prod<Geo>;
  constexpr int PERM = P::PERM, RAWS = P::RAWS, PORG = P::PORG, RPW = P::RPW, R4 = P::R4, NGW = P::NGW, SPW = P::SPW, NGWP = P::NGWP, SPWP = P::SPWP;
  constexpr int NE = Geo::NE, PQ = Geo::PQ, NIW = P::NIW, TPW = P::TPW;
  const int L = p.L;
  const float slope = p.pre_slope;
  const int nst = p.nchunks * Geo::HALVES / Geo::CPS;
  const int xs = 4 * w0 + Geo::XOFF;
  const bool interior = xs >= 0 && xs + Geo::RAW <= L;
  const int l_ = lane;
  // per-lane constants of this member (no barrier against hoisting any more: what the register allocator can keep over the tile loop it keeps - no spills, the
  // C = 64 launch 935 -> 922 us)
  float* rdst[SPW];
#pragma unroll
  for (int u = 0; u < SPW; ++u) {
    const int it = min(l_ + 64 * u, NGW - 1);
    rdst[u] = raw + (RPW * pw_ + it / R4) * RAWS + PORG + 4 * (it % R4);
  }
  float* pdst[SPWP];
  int pgb[SPWP];
  if constexpr (PERM > 0) {
#pragma unroll
    for (int u = 0; u < SPWP; ++u) {
      const int it = min(l_ + 64 * u, NGWP - 1);
      const int row = RPW * pw_ + it / Geo::PNG, g = it % Geo::PNG;
      pdst[u] = raw + row * RAWS + PORG + 4 * PERM * (g / PERM) + g % PERM;
      pgb[u] = (g / PERM) | ((g % PERM) << 16);
    }
  }
  int tsrc[TPW], tdst[TPW];
#pragma unroll
  for (int u = 0; u < TPW; ++u) {
    const int it = min(l_ + 64 * u, NIW - 1);
    const int row = RPW * pw_ + it / NE, e = it % NE;
    tsrc[u] = row * RAWS + PORG + 4 * e;
    tdst[u] = row * PQ + e;
    asm volatile("" : "+v"(tsrc[u]), "+v"(tdst[u]));       // kept in registers over the member's stages, not re-computed at each (conv_wino4_kernels.h)
  }
  // byte offsets of this lane's sixteen-byte groups inside a stage's rows: the same for every stage of the member (round 6, tools/acc3_timeline.py: acc3_issue
  // re-derived them - two divisions and a clamp per group - at EVERY stage, ~100 producer instructions per stage at one instruction per ~100 cycles)
  unsigned voff[P::NV];
  {
    if constexpr (PERM > 0) {
      const int pnblk_row = (L + 4 * PERM - 1) / (4 * PERM);
      const int bf = (xs + 4 * PERM) / (4 * PERM) - 1;
#pragma unroll
      for (int u = 0; u < SPWP; ++u) {
        const int it = min(l_ + 64 * u, NGWP - 1);
        const int row = RPW * pw_ + it / Geo::PNG, g = it % Geo::PNG;
        int bb = bf + g / PERM;
        if (!interior) bb = min(max(bb, 0), pnblk_row - 1);
        voff[u] = (unsigned)(row * p.x_ld + 4 * (PERM * bb + g % PERM)) * 4u;
      }
    } else {
#pragma unroll
      for (int u = 0; u < SPW; ++u) {
        const int it = min(l_ + 64 * u, NGW - 1);
        const int row = RPW * pw_ + it / R4, tg = xs + 4 * (it % R4);
        voff[u] = (unsigned)(row * p.x_ld + ((interior || (tg >= 0 && tg + 3 < L)) ? tg : 0)) * 4u;
      }
    }
  }
  const float* const xb_ = p.x + (long long)bz * p.x_bs;
  const int sstep = P::KS * p.x_ld * 4;
  for (int ch = 0; ch < nst; ++ch) {
    // ---- publish own rows
    if constexpr (PERM > 0) {
      const int pnblk_row = (L + 4 * PERM - 1) / (4 * PERM);
      const int bf = (xs + 4 * PERM) / (4 * PERM) - 1;
      const int delta = bf * (4 * PERM) - xs;
#pragma unroll
      for (int u = 0; u < SPWP; ++u) {
        if (64 * (u + 1) <= NGWP || lane < NGWP - 64 * u) {
          float4 q = v[u];
          if (!interior) {                                 // groups of blocks outside the row, and samples beyond L, are zero
            const int bb = bf + (pgb[u] & 0xffff), n0 = 4 * PERM * bb + ((pgb[u] >> 16) & 15);
            const bool inrow = bb >= 0 && bb < pnblk_row;
            q.x = (inrow && n0 < L) ? q.x : 0.f;
            q.y = (inrow && n0 + PERM < L) ? q.y : 0.f;
            q.z = (inrow && n0 + 2 * PERM < L) ? q.z : 0.f;
            q.w = (inrow && n0 + 3 * PERM < L) ? q.w : 0.f;
          }
          wino_lrelu4(q, slope);
          float* d = pdst[u] + delta;
          d[0] = q.x; d[PERM] = q.y; d[2 * PERM] = q.z; d[3 * PERM] = q.w;
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < SPW; ++u) {
        if (64 * (u + 1) <= NGW || lane < NGW - 64 * u) {
          float4 q = v[u];
          if (!interior) {                                 // L is a multiple of four (host): a group is inside or padding
            const int tg = xs + 4 * ((l_ + 64 * u) % R4);
            if (tg < 0 || tg + 3 >= L) q = make_float4(0.f, 0.f, 0.f, 0.f);
          }
          wino_lrelu4(q, slope);
          *reinterpret_cast<float4*>(rdst[u]) = q;
        }
      }
    }
    // ---- request the next stage's raw rows
    if (ch + 1 < nst) {
      const int so = (ch + 1) * sstep;
#pragma unroll
      for (int u = 0; u < (PERM > 0 ? SPWP : SPW); ++u) v[u] = w4_load16(xb_, voff[u], so);
    } else next();
    // ---- transform own rows into plane set s mod NPS (s_ holds that index; it wraps here)
    float* const pb = pl + (s_ & 0xff) * PLFMAX_;
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      if (64 * (u + 1) <= NIW || lane < NIW - 64 * u) {
        w4_transform_window<Geo>(pb + tdst[u], raw + tsrc[u]);
      }
    }
    // B_s: plane set complete.  Three sets: the barrier behind stage s is B_{s-1} (started = a stage has been produced before), the
    // last one follows the tile loop (conv_wino4.hip)
    if (NPS == 2 || (s_ >> 8)) __syncthreads();
    s_ = ((s_ & 0xff) + 1 == NPS ? 0 : (s_ & 0xff) + 1) | 0x100;
  }
}

// PIPE: the epilogue's residual loads are software-pipelined - quarter 0's twelve 16-byte loads are requested before the last member's
// stages, quarter Q + 1's behind quarter Q's output transform (whose accumulator rows are dead by then).  With the seven accumulators of
// the F(4,4) form there is room for the two sets of twelve registers (254 registers, 0-2 spilled; with F(4,3)'s eight the same code
// spilled 14-19 and gained nothing): accumulate launches 772 / 1666 / 1030 / 776 -> 777 / 1650 / 1004 / 734 us at C = 256 / 128 / 64 / 32,
// 16x512 step 26.49 -> 26.34 ms (profiles/r04_accumulate_pipelined_residuals_{on,off}.txt).  On with F44, off without.
// DBG: stamped instantiation (tools/acc3_timeline.py): consumer wave 0's cycles per member (barrier waits, MFMA streams) and in the epilogue, [workgroup][16] behind
// row 20000 of the stamp buffer
template <int NRT, int PERM, bool F44 = false, bool PIPE = false, bool DBG = false>
__global__ void __launch_bounds__(512, 2) conv_wino4_acc3_kernel(const WinoAcc3 g) {
  using AG = Acc3Geo<NRT, PERM, F44>;
  using G3 = typename AG::G3;
  using G7 = typename AG::G7;
  using G11 = typename AG::G11;
  constexpr int NWT = G11::NWT, NACC = AG::NACC;
  extern __shared__ __attribute__((aligned(16))) float wl[];
  float* const raw = wl;
  float* const pl = wl + AG::RAWMAX;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const WinoArgs& p3 = g.a[0];
  const WinoArgs& p7 = g.a[1];
  const WinoArgs& p11 = g.a[2];
  const int total = g.total, stride = gridDim.x, v0 = blockIdx.x;
  if (v0 >= total) return;
  const int my_tiles = (total - v0 + stride - 1) / stride;
  const int L = p3.L;
  auto locate = [&](int v, int& w0_, int& bz_, int& by_) {
    const int tl = xcd_linear(v, total, p3.xcd);
    const int t = tl / p3.ntn;
    bz_ = t / p3.gy;
    w0_ = (tl - t * p3.ntn) * NWT;
    by_ = t - bz_ * p3.gy;
  };
  int s_ = 0;                                              // plane set of the next stage (the producers keep a "started" flag in bit 8)

  if (wave >= 4) {
    // ================================================================= producers
    const int pw_ = wave - 4;
    __builtin_amdgcn_s_setprio(3);
    float4 v3[Acc3Prod<G3>::NV], v7[Acc3Prod<G7>::NV], v11[Acc3Prod<G11>::NV];
    int w0, bz, by;
    locate(v0, w0, bz, by);
    acc3_issue<G3>(v3, p3, bz, 4 * w0 + G3::XOFF, 0, lane, pw_);
    for (int ti = 0; ti < my_tiles; ++ti) {
      int w0n = w0, bzn = bz, byn = by;
      const bool more = ti + 1 < my_tiles;
      if (more) locate(v0 + (ti + 1) * stride, w0n, bzn, byn);
      acc3_produce<G3, AG::NPS>(p3, v3, raw, pl, AG::PLFMAX, w0, bz, s_, lane, pw_, [&]() { acc3_issue<G7>(v7, p7, bz, 4 * w0 + G7::XOFF, 0, lane, pw_); });
      acc3_produce<G7, AG::NPS>(p7, v7, raw, pl, AG::PLFMAX, w0, bz, s_, lane, pw_, [&]() { acc3_issue<G11>(v11, p11, bz, 4 * w0 + G11::XOFF, 0, lane, pw_); });
      acc3_produce<G11, AG::NPS>(p11, v11, raw, pl, AG::PLFMAX, w0, bz, s_, lane, pw_,
                        [&]() { if (more) acc3_issue<G3>(v3, p3, bzn, 4 * w0n + G3::XOFF, 0, lane, pw_); });
      w0 = w0n; bz = bzn; by = byn;
    }
    if constexpr (AG::NPS == 3) __syncthreads();
    return;
  }

  // =================================================================== consumers: row tile rt, column tile ct of the workgroup
  const int l31 = lane & 31, hi = lane >> 5;
  const int rt = NRT == 4 ? wave : (NRT == 2 ? (wave & 1) : 0), ct = NRT == 4 ? 0 : (NRT == 2 ? (wave >> 1) : wave);
  const int uu = ct * 32 + l31;
  const unsigned plbase = (unsigned)(size_t)pl;
  const unsigned wlane = (unsigned)lane * 16u;
  f32x16 M[NACC];
  const unsigned ylb = (unsigned)p3.y_ld * 4u;
  unsigned yo4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) yo4[i] = (unsigned)((4 * hi + i) * p3.y_ld + 4 * uu) * 4u;
  long long cy3[2] = {0, 0}, cy7[2] = {0, 0}, cy11[2] = {0, 0}, cy_epi = 0, cy_all0 = 0, wall0 = 0;
  if constexpr (DBG) { cy_all0 = (long long)__builtin_readcyclecounter(); wall0 = (long long)wall_clock64(); }
  for (int ti = 0; ti < my_tiles; ++ti) {
    int w0, bz, by;
    locate(v0 + ti * stride, w0, bz, by);
    const int n0 = 4 * w0, ne = n0 + 4 * uu;
    const int mt = by * NRT + rt;
    const bool row_ok = mt < p3.mtiles;
    const int mtc = row_ok ? mt : p3.mtiles - 1;
    {   // the bias starts in M1 (part of all four outputs): the sum of the three members' biases
      const float* b3 = p3.bias + mtc * 32 + 4 * hi;
      const float* b7 = p7.bias + mtc * 32 + 4 * hi;
      const float* b11 = p11.bias + mtc * 32 + 4 * hi;
#pragma unroll
      for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int r = (i & 3) + 8 * (i >> 2);
          M[q][i] = q == 1 ? (b3[r] + b7[r]) + b11[r] : 0.f;
        }
    }
    acc3_consume<G3, NACC, AG::NPS, DBG>(p3, M, plbase, AG::PLFMAX, __builtin_amdgcn_readfirstlane(mtc * p3.nchunks * G3::WSLOTS * 4096), s_, wlane, (unsigned)(hi * G3::PQ + uu) * 4u, cy3);
    acc3_consume<G7, NACC, AG::NPS, DBG>(p7, M, plbase, AG::PLFMAX, __builtin_amdgcn_readfirstlane(mtc * p7.nchunks * G7::WSLOTS * 4096), s_, wlane, (unsigned)(hi * G7::PQ + uu) * 4u, cy7);
    const bool lane_ok = row_ok && ne < L;
    const long long roff3 = (long long)bz * p3.res_bs + (long long)(mt * 32 + 4 * hi) * p3.res_ld + ne;
    const long long roff7 = (long long)bz * p7.res_bs + (long long)(mt * 32 + 4 * hi) * p7.res_ld + ne;
    const long long roff11 = (long long)bz * p11.res_bs + (long long)(mt * 32 + 4 * hi) * p11.res_ld + ne;
    auto rq = [&](float4 (&rr)[12], const int Q) {          // the residuals of quarter Q: rows 8 Q + 4 hi + r of the three chains
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        rr[r] = *reinterpret_cast<const float4*>(p3.res + roff3 + (long long)(8 * Q + r) * p3.res_ld);
        rr[4 + r] = *reinterpret_cast<const float4*>(p7.res + roff7 + (long long)(8 * Q + r) * p7.res_ld);
        rr[8 + r] = *reinterpret_cast<const float4*>(p11.res + roff11 + (long long)(8 * Q + r) * p11.res_ld);
      }
    };
    float4 ra[12], rb[12];
    if constexpr (PIPE) if (lane_ok) rq(ra, 0);
    acc3_consume<G11, NACC, AG::NPS, DBG>(p11, M, plbase, AG::PLFMAX, __builtin_amdgcn_readfirstlane(mtc * p11.nchunks * G11::WSLOTS * 4096), s_, wlane, (unsigned)(hi * G11::PQ + uu) * 4u, cy11);
    long long ce0 = 0;
    if constexpr (DBG) ce0 = (long long)__builtin_readcyclecounter();
    // ---- output transform + epilogue: y = (A^T M + res_3 + res_7 + res_11) / div, sixteen-byte stores
    if (lane_ok) {
      char* const ybase = reinterpret_cast<char*>(p3.y + (long long)bz * p3.y_bs + (long long)(mt * 32) * p3.y_ld + n0);
      const float dv = p3.div, rc = 1.0f / dv;
      const bool dodiv = (p3.flags & F_DIV) != 0;
      auto dv1 = [&](float x) { const float q = x * rc; return __builtin_fmaf(__builtin_fmaf(-q, dv, x), rc, q); };
      auto quarter = [&](auto q_c, float4 (&cur)[12], float4 (&nxt)[12]) {
        constexpr int Q = decltype(q_c)::value;
        if constexpr (!PIPE) rq(cur, Q);
        float4 vo[4];
#pragma unroll
        for (int r = 0; r < 4; r += 2) w4_output_transform2<G11, NACC>(M, 4 * Q + r, vo[r], vo[r + 1]);
        if constexpr (PIPE && Q < 3) {
          __builtin_amdgcn_sched_barrier(0);                 // the next quarter's loads go out here, not behind this quarter's stores
          rq(nxt, Q + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          w4_add4(vo[r], cur[r]); w4_add4(vo[r], cur[4 + r]); w4_add4(vo[r], cur[8 + r]);      // ((vo + res_3) + res_7) + res_11, packed
          if (dodiv) vo[r] = make_float4(dv1(vo[r].x), dv1(vo[r].y), dv1(vo[r].z), dv1(vo[r].w));
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<float4*>(ybase + (size_t)(8 * Q) * ylb + yo4[r]) = vo[r];
      };
      quarter(std::integral_constant<int, 0>{}, ra, rb);
      quarter(std::integral_constant<int, 1>{}, rb, ra);
      quarter(std::integral_constant<int, 2>{}, ra, rb);
      quarter(std::integral_constant<int, 3>{}, rb, ra);
    }
    if constexpr (DBG) cy_epi += (long long)__builtin_readcyclecounter() - ce0;
  }
  if constexpr (DBG) if (tid == 0) {
    long long* d = p3.dbg + (20000LL + blockIdx.x) * 16;
    d[0] = my_tiles; d[1] = (long long)__builtin_readcyclecounter() - cy_all0; d[2] = cy3[0]; d[3] = cy3[1]; d[4] = cy7[0]; d[5] = cy7[1]; d[6] = cy11[0]; d[7] = cy11[1];
    d[8] = cy_epi; d[10] = wall0; d[11] = (long long)wall_clock64();
  }
}

// ------------------------------------------------------------------------------------------------ launch
template <int NRT, int PERM, bool F44 = false>
static int acc3_launch_n(const WinoAcc3& g, hipStream_t st) {
  using AG = Acc3Geo<NRT, PERM, F44>;
  static_assert(AG::LDS_BYTES <= 160 * 1024, "tile does not fit");
  const unsigned grid = (unsigned)std::min<long long>(g.total, (long long)device_cu_count());
  if constexpr (PERM == 5 && NRT >= 2) {                      // stamped build: the launches the model takes at C >= 64
    if (g.a[0].dbg) {
      auto kern = conv_wino4_acc3_kernel<NRT, PERM, F44, F44, true>;
      SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
      hipLaunchKernelGGL(kern, dim3(grid), dim3(512), (size_t)AG::LDS_BYTES, st, g);
      return SVOC_OK;
    }
  }
  auto kern = conv_wino4_acc3_kernel<NRT, PERM, F44, F44>;
  SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), (size_t)AG::LDS_BYTES, st, g);
  return SVOC_OK;
}
// members in chain order (k = 3, 7, 11, dilation 1, epilogue flags of a plain residual), one tile space of `total` tiles; in_perm: 0, or
// the dilation (3 / 5) of the convolutions that wrote the members' inputs window-major; a[0] carries the output, F_DIV and div
int wino4_launch_acc3(const WinoArgs* a, int NRT, int in_perm, bool f44, long long total, hipStream_t st) {
  WinoAcc3 g;
  for (int i = 0; i < 3; ++i) g.a[i] = a[i];
  g.a[0].flags |= a[2].flags & F_DIV;                       // the division by the number of chains rides on the last member
  g.a[0].div = a[2].div;
  g.total = (int)total;
  if (!f44) return 1;                                      // (the F(4,3) form of the merged launch went with SVOC_W4_F44 in round 5)
#define SVOC_W4M(P) (NRT == 4 ? acc3_launch_n<4, P, true>(g, st) : (NRT == 2 ? acc3_launch_n<2, P, true>(g, st) : acc3_launch_n<1, P, true>(g, st)))
  if (in_perm == 5) return SVOC_W4M(5);
  if (in_perm == 3) return SVOC_W4M(3);
  return SVOC_W4M(0);
#undef SVOC_W4M
}
}  // namespace svoc
