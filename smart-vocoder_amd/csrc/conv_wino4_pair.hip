// One ResBlock1 iteration - c1 (dilation D1) -> lrelu -> c2 (dilation 1) -> + x (reference modules.py:212-219) - of the C = 32 MRF stage
// (and, NRT = 2, the undilated iteration of the C = 64 stage) in ONE launch, both convolutions in Winograd F(4,3) form (F44: k = 7 / 11 in
// F(4,4) form, conv_wino4.h), the intermediate tile kept in LDS (round 4).
//
// Conv by conv (conv_wino4.hip, one row tile per workgroup) the C = 32 stage moves five tensor passes per iteration (x in, c1 out, c1 in,
// x as residual, y out) where the fused direct-form kernel of round 1 moved two, and its k = 3 members are HBM-bound.  With one row tile
// a workgroup holds ALL 32 channels of its columns, so the c1 tile can feed the c2 of the same workgroup: the consumers' c1 epilogue
// writes lrelu(c1 + bias) - zero outside [0, L), c2's padding - into an LDS tile that IS the raw tile of c2 (32 rows x 4 NWC1 columns,
// 64 KB; c1's own raw staging rows alias its head, they are dead by then), the producers transform c2's stages from it, and c2's
// epilogue adds x from global (L2-hot: c1's producers have just read it) and stores.  Per tile: c1 computes NWC1 = (128 / D1) D1
// windows from the q-block-aligned column m0 at or below c2's raw origin; c2 keeps NW2 windows (124 ... 117 of 128 lanes), the largest
// count whose raw columns lie inside c1's; tiles advance by 4 NW2 columns (c1 recomputes the 2 x halo: 3 - 9 %).  One extra workgroup
// barrier per tile (the intermediate tile is complete) and one exposed producer pass (c2's first stage) against what is saved: c1's
// 16-byte stores, c2's global staging and 3 of 5 tensor passes.  Eight-channel stages for every member (k = 3 too: the 16-channel
// planes would not fit beside the intermediate tile); two plane sets.
#include "svoc_internal.h"
#include "wino_common.h"
#include "conv_wino4.h"
#include "conv_wino4_consume.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace svoc {

struct PairMember {
  const float* x; long long x_bs; int x_ld;                // input = residual [B][32][x_ld]
  float* y; long long y_bs; int y_ld;                       // output
  const float* wp1; const float* bias1;                     // c1: F(4,3) (F44: k = 7 / 11 F(4,4)) image, bias
  const float* wp2; const float* bias2;                     // c2
  unsigned eflags; float div;                               // accumulate form: F_ACC (y += ...) and F_DIV (... / div) of this member's epilogue
};
struct PairGroup { PairMember m[3]; int end[3]; int L; int B; int xcd; unsigned flags; float slope; long long* dbg; };      // dbg: stamped build only (tools/pair_timeline.py)   // members k = 11, 7, 3; end[i] = first tile id behind member i
// NRT = 1: C = 32 (1 x 4 consumers, 128 windows per tile); NRT = 2: C = 64 (2 x 2 consumers, 64 windows per tile, two 32-channel chunks)

// NW2CAP > 0: keep at most that many c2 windows per tile (the accumulate form gives the three members ONE tile space: the k = 11 member's count)
template <int K, int D1, int NRT, bool F44_ = false, int NW2CAP = 0>
struct PairGeo {
  static constexpr int KD = K == 3 ? 2 : 1;                 // 8 (C = 32) / 16 (C = 64) channels per stage for every member: four stages per phase
  static constexpr bool F44 = F44_ && K >= 7;               // k = 3 stays F(4,3)
  using G1 = W4Geo<K, D1, NRT, 0, KD, F44>;
  using G2 = W4Geo<K, 1, NRT, 0, KD, F44>;
  static constexpr int SPAN = G2::NV;                       // samples of a window: 6 (F44: 7)
  static constexpr int KS = G1::KS, ROWS = 32 * NRT, NWT = G1::NWT;
  static_assert(G1::KS == 8 * NRT && G2::KS == 8 * NRT && ROWS / KS == 4, "four stages per phase");
  static constexpr int NWC1 = (NWT / D1) * D1;              // c1 windows per tile: whole q blocks
  static constexpr int COLS1 = 4 * NWC1;                    // columns of the intermediate tile
  static constexpr int NW2OWN = (COLS1 - 4 * D1 + 4 - G2::LEAD - SPAN) / 4 - G2::G + 2;   // c2 windows this member could keep per tile
  static constexpr int NW2 = (NW2CAP > 0 && NW2CAP < NW2OWN) ? NW2CAP : NW2OWN;          // c2 windows kept per tile
  static constexpr int NE2 = NW2 + G2::G - 1;               // c2 plane entries needed
  static constexpr int W2 = 4 * NW2;                        // tile step in output columns
  static constexpr int MIDS = COLS1;                        // row stride of the intermediate tile
  static constexpr int MID_FLOATS = ROWS * MIDS;
  static constexpr int cmax(int a, int b) { return a > b ? a : b; }
  static constexpr int PLFMAX = cmax(G1::PLF, G2::PLF);
  static_assert(G1::RAW_FLOATS <= MID_FLOATS, "c1's raw staging rows alias the head of the intermediate tile");
  static_assert((4 * D1 - 4) + G2::LEAD + 4 * (NE2 - 1) + SPAN <= COLS1, "c2's raw columns lie inside c1's");
  static constexpr int LDS_FLOATS = MID_FLOATS + 2 * PLFMAX;
};

template <int K, int D1, int NRT, bool F44, bool DBG = false>
__device__ __forceinline__ void pair_member(const PairMember& pm, const PairGroup& g, const int first, const int vend, const int blk, const int G_) {
  using PG = PairGeo<K, D1, NRT, F44>;
  using G1 = typename PG::G1;
  using G2 = typename PG::G2;
  constexpr int NWC1 = PG::NWC1, NW2 = PG::NW2, NE2 = PG::NE2, W2 = PG::W2, MIDS = PG::MIDS, PLFMAX = PG::PLFMAX, NACC = G1::NACC;
  constexpr int PADT = G1::PADT, KS = PG::KS;
  if (vend <= first) return;
  int v0 = blk - first % G_;
  if (v0 < 0) v0 += G_;
  v0 += first;
  if (v0 >= vend) return;
  extern __shared__ __attribute__((aligned(16))) float wl[];
  float* const mid = wl;                                    // intermediate tile [C][MIDS]; c1's raw rows [KS][RAW1] alias its head
  float* const raw = wl;
  float* const pl = wl + PG::MID_FLOATS;                    // two plane sets of PLFMAX floats
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = g.L;
  const int ntn = (L + W2 - 1) / W2;
  const int ntiles_all = vend - first;
  const int my_tiles = (vend - v0 + G_ - 1) / G_;
  // tile v -> (batch element, first output column n2 of c2, origin m0 of c1 / of the intermediate tile, offset of c2's raw origin in it)
  auto locate = [&](int v, int& bz_, int& n2_, int& m0_, int& off2_) {
    const int tl = xcd_linear(v - first, ntiles_all, g.xcd);
    bz_ = tl / ntn;
    n2_ = (tl - bz_ * ntn) * W2;
    const int xs2 = n2_ + G2::XOFF;                         // c2's raw origin: a multiple of four, >= -8
    const int b0 = (xs2 + 4 * D1 * 4) / (4 * D1) - 4;       // floor(xs2 / 4 D1)
    m0_ = 4 * D1 * b0;
    off2_ = xs2 - m0_;
  };
  // c1's raw tile for the tile whose windows start at column m0: first raw column xs (multiple of four), raw index of window 0's d0
  auto origin1 = [&](int m0_, int& xs_, int& lead_) {
    const int f0 = m0_ - PADT * D1;
    xs_ = f0 & ~3;
    lead_ = f0 - xs_;
  };
  int s_ = 0;                                               // plane set of the next stage (both sides count stages alike)
  // stamped build: [dilation 1 / 3 / 5][member k = 11 / 7 / 3][workgroup][16] behind row 8192 of the stamp buffer
  long long* const drow = DBG ? g.dbg + (8192LL + ((D1 == 1 ? 0 : (D1 == 3 ? 1 : 2)) * 3 + (K == 11 ? 0 : (K == 7 ? 1 : 2))) * 1024 + blockIdx.x) * 16 : nullptr;
  long long pc_all0 = 0, pc_bar = 0;
  if constexpr (DBG) pc_all0 = (long long)__builtin_readcyclecounter();
  auto pbarrier = [&]() {
    long long b0 = 0;
    if constexpr (DBG) b0 = (long long)__builtin_readcyclecounter();
    __syncthreads();
    if constexpr (DBG) pc_bar += (long long)__builtin_readcyclecounter() - b0;
  };

  if (wave >= 4) {
    // ================================================================= producers
    const int pw_ = wave - 4;
    __builtin_amdgcn_s_setprio(3);
    constexpr int RPW = KS / 4, RAW1 = G1::RAW, R4 = RAW1 / 4, NGW = RPW * R4, SPW = (NGW + 63) / 64;
    constexpr int NE1 = G1::NE, NIW1 = RPW * NE1, TPW1 = (NIW1 + 63) / 64, PQ1 = G1::PQ;
    constexpr int NIW2 = RPW * NE2, TPW2 = (NIW2 + 63) / 64, PQ2 = G2::PQ;
    const long long ldb = (long long)pm.x_ld * 4;
    const float slope = g.slope;
    unsigned goff[SPW];
    float* rdst[SPW];
#pragma unroll
    for (int u = 0; u < SPW; ++u) {
      const int it = min(lane + 64 * u, NGW - 1);
      const int row = RPW * pw_ + it / R4, g4 = it % R4;
      goff[u] = (unsigned)(row * pm.x_ld + 4 * g4) * 4u;
      rdst[u] = raw + row * RAW1 + 4 * g4;
    }
    int t1off[TPW1], t1dst[TPW1];                            // c1 transform items: raw offset of d0 (without the tile's lead), plane entry
#pragma unroll
    for (int u = 0; u < TPW1; ++u) {
      const int it = min(lane + 64 * u, NIW1 - 1);
      const int row = RPW * pw_ + it / NE1, e = it % NE1;
      const int qe = e / D1, pe = e - qe * D1;
      t1off[u] = row * RAW1 + 4 * D1 * qe + pe;
      t1dst[u] = row * PQ1 + e;
      asm volatile("" : "+v"(t1off[u]), "+v"(t1dst[u]));     // kept in registers, not re-computed at every stage (conv_wino4_kernels.h)
    }
    int t2src[TPW2], t2dst[TPW2];                            // c2 transform items: row and window of the intermediate tile, plane entry
#pragma unroll
    for (int u = 0; u < TPW2; ++u) {
      const int it = min(lane + 64 * u, NIW2 - 1);
      const int row = RPW * pw_ + it / NE2, e = it % NE2;
      t2src[u] = row * MIDS + 4 * e;
      t2dst[u] = row * PQ2 + e;
      asm volatile("" : "+v"(t2src[u]), "+v"(t2dst[u]));
    }
    float4 v[SPW];
    auto issue = [&](int bz_, int xs_, int ch) {
      const char* cb = reinterpret_cast<const char*>(pm.x + (long long)bz_ * pm.x_bs) + (long long)ch * KS * ldb;
      if (xs_ >= 0 && xs_ + RAW1 <= L) {
        const float* const xb = pm.x + (long long)bz_ * pm.x_bs;
        const int so = ch * KS * (int)ldb + xs_ * 4;
#pragma unroll
        for (int u = 0; u < SPW; ++u) v[u] = w4_load16(xb, goff[u], so);
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
    int bz, n2, m0, off2, xs1, lead1;
    locate(v0, bz, n2, m0, off2);
    origin1(m0, xs1, lead1);
    issue(bz, xs1, 0);
    for (int ti = 0; ti < my_tiles; ++ti) {
      int bzn = bz, n2n = n2, m0n = m0, off2n = off2, xs1n = xs1, lead1n = lead1;
      const bool more = ti + 1 < my_tiles;
      if (more) { locate(v0 + (ti + 1) * G_, bzn, n2n, m0n, off2n); origin1(m0n, xs1n, lead1n); }
      const bool interior = xs1 >= 0 && xs1 + RAW1 <= L;
      // ---------------- phase A: c1's four stages (8 channels each) from global
      for (int ch = 0; ch < 4; ++ch) {
        if (interior) {
#pragma unroll
          for (int u = 0; u < SPW; ++u) {
            if (64 * (u + 1) <= NGW || lane < NGW - 64 * u) {
              float4 q = v[u];
              wino_lrelu4(q, slope);
              *reinterpret_cast<float4*>(rdst[u]) = q;
            }
          }
        } else {                                             // L is a multiple of four: a 16-byte group is inside the row or padding
          int l_ = lane;
          asm volatile("" : "+v"(l_));
#pragma unroll
          for (int u = 0; u < SPW; ++u) {
            if (64 * (u + 1) <= NGW || lane < NGW - 64 * u) {
              const int tg = xs1 + 4 * ((l_ + 64 * u) % R4);
              float4 q = v[u];
              if (tg < 0 || tg + 3 >= L) q = make_float4(0.f, 0.f, 0.f, 0.f);
              wino_lrelu4(q, slope);
              *reinterpret_cast<float4*>(rdst[u]) = q;
            }
          }
        }
        if (ch + 1 < 4) issue(bz, xs1, ch + 1);
        float* const pb = pl + s_ * PLFMAX;
#pragma unroll
        for (int u = 0; u < TPW1; ++u) {
          if (64 * (u + 1) <= NIW1 || lane < NIW1 - 64 * u) {
            // d0 of the window sits at raw + t1off + lead1; D1 = 1: lead1 is 3 (k = 3, 11) or 1 (k = 7) = G1::LEAD, and the shared
            // transform takes the sixteen-byte group LEAD columns ahead of d0 (the compile-time forms of conv_wino4.h)
            const float* r = raw + t1off[u] + lead1;
            w4_transform_window<G1>(pb + t1dst[u], D1 == 1 ? r - G1::LEAD : r);
          }
        }
        pbarrier();                                          // B_s: plane set complete
        s_ ^= 1;
      }
      pbarrier();                                            // X: the consumers have written the rows of c2's first stage into the intermediate tile
      // ---------------- phase B: c2's four stages from the intermediate tile; the next tile's first raw rows are requested meanwhile
      if (more) issue(bzn, xs1n, 0);
      for (int ch = 0; ch < 4; ++ch) {
        float* const pb = pl + s_ * PLFMAX;
        const float* const mrow = mid + ch * KS * MIDS + off2;
#pragma unroll
        for (int u = 0; u < TPW2; ++u) {
          if (64 * (u + 1) <= NIW2 || lane < NIW2 - 64 * u) {
            w4_transform_window<G2>(pb + t2dst[u], mrow + t2src[u]);
          }
        }
        pbarrier();
        s_ ^= 1;
      }
      bz = bzn; n2 = n2n; m0 = m0n; off2 = off2n; xs1 = xs1n; lead1 = lead1n;
    }
    if constexpr (DBG) if (tid == 256) { drow[8] = (long long)__builtin_readcyclecounter() - pc_all0; drow[9] = pc_bar; }
    return;
  }

  // =================================================================== consumers: row tile rt, column tile ct of the workgroup
  const int l31 = lane & 31, hi = lane >> 5;
  const int rt = NRT == 2 ? (wave & 1) : 0, ct = NRT == 2 ? (wave >> 1) : wave;
  const int uu = ct * 32 + l31;                             // this lane's window of the tile
  const unsigned plbase = (unsigned)(size_t)pl;
  const unsigned wlane = (unsigned)lane * 16u;
  f32x16 M[NACC];
  WinoArgs p1{}, p2{};
  p1.wp = pm.wp1; p1.nchunks = NRT;
  p2.wp = pm.wp2; p2.nchunks = NRT;
  const int wt1 = __builtin_amdgcn_readfirstlane(rt * NRT * G1::WSLOTS * 4096), wt2 = __builtin_amdgcn_readfirstlane(rt * NRT * G2::WSLOTS * 4096);
  const float slope = g.slope;
  auto init_bias = [&](const float* bias) {                 // the bias starts in M1 (part of all four outputs)
    const float* bq = bias + 32 * rt + 4 * hi;
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) M[q][i] = q == 1 ? bq[(i & 3) + 8 * (i >> 2)] : 0.f;
  };
  auto ytrans = [&](auto q_c, float4 (&vo)[4]) {
    constexpr int Q = decltype(q_c)::value;
#pragma unroll
    for (int r = 0; r < 4; r += 2) w4_output_transform2<G1, NACC>(M, 4 * Q + r, vo[r], vo[r + 1]);
  };
  long long cyA[2] = {0, 0}, cyB[2] = {0, 0}, cy_epiA = 0, cy_epiB = 0, wall0 = 0;
  if constexpr (DBG) wall0 = (long long)wall_clock64();
  for (int ti = 0; ti < my_tiles; ++ti) {
    int bz, n2, m0, off2;
    locate(v0 + ti * G_, bz, n2, m0, off2);
    // ---------------- phase A: c1 into the accumulators
    init_bias(pm.bias1);
    acc3_consume<G1, NACC, 2, DBG>(p1, M, plbase, PLFMAX, wt1, s_, wlane, (unsigned)(hi * G1::PQ + uu) * 4u, cyA);
    long long ce0 = 0;
    if constexpr (DBG) ce0 = (long long)__builtin_readcyclecounter();
    // ---- epilogue A: lrelu(c1) -> the intermediate tile, zero outside [0, L).  Window uu = q block uu / D1, phase uu % D1: its four
    // outputs are D1 columns apart
    {
      const bool act = uu < NWC1;
      const int bq = uu / D1, ph = uu - bq * D1;
      const int c0 = 4 * D1 * bq + ph;                       // column of output 0 inside the tile
      const int n0 = m0 + c0;
      auto quarter = [&](auto q_c) {
        constexpr int Q = decltype(q_c)::value;
        if (!act) return;
        float4 vo[4];
        ytrans(q_c, vo);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          wino_lrelu4(vo[r], slope);
          float* d = mid + (32 * rt + 8 * Q + 4 * hi + r) * MIDS + c0;
          if constexpr (D1 == 1) {
            if (n0 < 0 || n0 >= L) vo[r] = make_float4(0.f, 0.f, 0.f, 0.f);      // L and n0 are multiples of four
            *reinterpret_cast<float4*>(d) = vo[r];
          } else {
            d[0] = (n0 >= 0 && n0 < L) ? vo[r].x : 0.f;
            d[D1] = (n0 + D1 >= 0 && n0 + D1 < L) ? vo[r].y : 0.f;
            d[2 * D1] = (n0 + 2 * D1 >= 0 && n0 + 2 * D1 < L) ? vo[r].z : 0.f;
            d[3 * D1] = (n0 + 3 * D1 >= 0 && n0 + 3 * D1 < L) ? vo[r].w : 0.f;
          }
        }
      };
      // X: c2's first stage stages channels 0 .. 8 NRT - 1 = the rows of quarters 0 .. NRT - 1 of row tile 0, so the producers are let go
      // as soon as THOSE are written and transform c2's first stage while the consumers store the remaining quarters (rows 8 NRT ..,
      // complete before the consumers arrive at that stage's barrier, behind which the producers turn to the next stage's rows).
      // (A first version held the producers until the whole intermediate tile was written: the producer pass it exposed was 3-4 % of a tile.)
      quarter(std::integral_constant<int, 0>{});
      if constexpr (NRT == 1) __syncthreads();
      quarter(std::integral_constant<int, 1>{});
      if constexpr (NRT == 2) __syncthreads();
      quarter(std::integral_constant<int, 2>{});
      quarter(std::integral_constant<int, 3>{});
    }
    if constexpr (DBG) cy_epiA += (long long)__builtin_readcyclecounter() - ce0;
    // ---------------- phase B: c2, then + x and store
    // The residual rows of the tile are requested HERE, ahead of c2's MFMA streams (round 6, tools/pair_timeline.py: requested at the head of the epilogue their
    // L2 round trip was exposed - the epilogue took 5.1 - 7.0 k cycles of a 28 - 63 k cycle tile at C = 32): all sixteen where the registers allow (one row tile per
    // workgroup: 164 + 64), the first eight with two row tiles (201 + 32), the rest at the head of the epilogue as before.
    constexpr int NEARLY = NRT == 1 ? 16 : 8;
    const int ne = n2 + 4 * uu;
    const bool lane_ok = uu < NW2 && ne < L;
    const char* const rbase = reinterpret_cast<const char*>(pm.x + (long long)bz * pm.x_bs + (long long)(32 * rt + 4 * hi) * pm.x_ld + (lane_ok ? ne : 0));
    const size_t rlb = (size_t)pm.x_ld * 4;
    float4 rvall[16];
    init_bias(pm.bias2);                                     // (first: its wait for the sixteen bias words would otherwise wait for the rows requested below as well)
    if (lane_ok) {
#pragma unroll
      for (int r = 0; r < NEARLY; ++r) rvall[r] = *reinterpret_cast<const float4*>(rbase + (size_t)(8 * (r >> 2) + (r & 3)) * rlb);
    }
    acc3_consume<G2, NACC, 2, DBG>(p2, M, plbase, PLFMAX, wt2, s_, wlane, (unsigned)(hi * G2::PQ + uu) * 4u, cyB);
    if constexpr (DBG) ce0 = (long long)__builtin_readcyclecounter();
    if (lane_ok) {
      char* const ybase = reinterpret_cast<char*>(pm.y + (long long)bz * pm.y_bs + (long long)(32 * rt + 4 * hi) * pm.y_ld + ne);
      const size_t ylb = (size_t)pm.y_ld * 4;
#pragma unroll
      for (int r = NEARLY; r < 16; ++r) rvall[r] = *reinterpret_cast<const float4*>(rbase + (size_t)(8 * (r >> 2) + (r & 3)) * rlb);
      auto quarter = [&](auto q_c) {
        constexpr int Q = decltype(q_c)::value;
        float4 vo[4];
        ytrans(q_c, vo);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          w4_add4(vo[r], rvall[4 * Q + r]);
          *reinterpret_cast<float4*>(ybase + (size_t)(8 * Q + r) * ylb) = vo[r];
        }
      };
      quarter(std::integral_constant<int, 0>{});
      quarter(std::integral_constant<int, 1>{});
      quarter(std::integral_constant<int, 2>{});
      quarter(std::integral_constant<int, 3>{});
    }
    if constexpr (DBG) cy_epiB += (long long)__builtin_readcyclecounter() - ce0;
  }
  if constexpr (DBG) if (tid == 0) {      // [0] tiles, [1] total cycles, c1: [2] barrier waits [3] MFMA streams [4] epilogue -> LDS, c2: [5] [6] [7] epilogue; [8] [9] producer wave 0: total, barrier waits
    drow[0] = my_tiles; drow[1] = (long long)__builtin_readcyclecounter() - pc_all0;
    drow[2] = cyA[0]; drow[3] = cyA[1]; drow[4] = cy_epiA; drow[5] = cyB[0]; drow[6] = cyB[1]; drow[7] = cy_epiB;
    drow[10] = wall0; drow[11] = (long long)wall_clock64();
  }
}

template <int D1, int NRT, bool F44, bool DBG = false>
__global__ void __launch_bounds__(512, 2) conv_wino4_pair_kernel(const PairGroup g) {
  const int b = blockIdx.x, G_ = gridDim.x;
  pair_member<11, D1, NRT, F44, DBG>(g.m[0], g, 0, g.end[0], b, G_);
  __syncthreads();
  pair_member<7, D1, NRT, F44, DBG>(g.m[1], g, g.end[0], g.end[1], b, G_);
  __syncthreads();
  pair_member<3, D1, NRT, F44, DBG>(g.m[2], g, g.end[1], g.end[2], b, G_);
}

// ------------------------------------------------------------------------------------------------ the accumulate form (round 5)
// The LAST step of the C = 32 MRF stage (reference models.py:149-155: xs = sum of the three chains' ResBlocks, x = xs / 3): c1 (dilation D1) -> c2
// of the three chains AND the sum over the chains in ONE launch.  TILE-major: a workgroup runs the k = 11, k = 7 and k = 3 pairs of one tile back to
// back - the members share ONE tile space (the k = 11 member's c2 windows per tile, PairGeo's NW2CAP) and one output tensor: the k = 11 member writes
// y = c2 + x_11, the k = 7 member adds its c2 + x_7 to what that very lane stored a few microseconds earlier (program order; the tile is still in the
// L2), the k = 3 member adds, divides and stores.  Against the window-major c1 launch + the merged accumulate launch (conv_wino4_acc.hip): c1's three
// stores and c2's three global staging passes are gone - HBM sees three inputs and one output (member-major, the first version: the read-modify-write
// went through HBM, 1217 against 1239 us).  F(4,4) images for k = 7 / 11, F(4,3) for k = 3 (each member keeps its own form: no shared accumulators).
template <class T> struct PairTag { using type = T; };
template <int K, int D1, bool F44>
struct PairAccT {
  using PG = PairGeo<K, D1, 1, F44, PairGeo<11, D1, 1, F44>::NW2OWN>;
  using G1 = typename PG::G1;
  using G2 = typename PG::G2;
  static constexpr int KS = PG::KS, RPW = KS / 4, RAW1 = G1::RAW, R4 = RAW1 / 4, NGW = RPW * R4, SPW = (NGW + 63) / 64;
  static constexpr int NE1 = G1::NE, NIW1 = RPW * NE1, TPW1 = (NIW1 + 63) / 64, PQ1 = G1::PQ;
  static constexpr int NE2 = PG::NE2, NIW2 = RPW * NE2, TPW2 = (NIW2 + 63) / 64, PQ2 = G2::PQ;
};
// tile (batch element bz, first c2 output column n2) as member K sees it: origin m0 of its c1 / intermediate tile, offset of c2's raw origin in it,
// c1's first raw column xs1 (a multiple of four) and the raw index lead1 of window 0's first sample
template <class T>
__device__ __forceinline__ void pairacc_origin(const int n2, int& m0, int& off2, int& xs1, int& lead1) {
  constexpr int D1 = T::G1::DIL;
  const int xs2 = n2 + T::G2::XOFF;
  const int b0 = (xs2 + 4 * D1 * 4) / (4 * D1) - 4;
  m0 = 4 * D1 * b0;
  off2 = xs2 - m0;
  const int f0 = m0 - T::G1::PADT * D1;
  xs1 = f0 & ~3;
  lead1 = f0 - xs1;
}
template <class T>
__device__ __forceinline__ void pairacc_issue(float4 (&v)[T::SPW], const PairMember& pm, const int L, const int bz, const int xs, const int ch, const int lane, const int pw_) {
  const long long ldb = (long long)pm.x_ld * 4;
  const float* const xb = pm.x + (long long)bz * pm.x_bs;
  const int so = ch * T::KS * (int)ldb;
  const bool interior = xs >= 0 && xs + T::RAW1 <= L;
  int l_ = lane;
  asm volatile("" : "+v"(l_));                              // keeps the per-lane address arithmetic inside the call (conv_wino4_acc.hip)
#pragma unroll
  for (int u = 0; u < T::SPW; ++u) {
    const int it = min(l_ + 64 * u, T::NGW - 1);
    const int row = T::RPW * pw_ + it / T::R4, tg = xs + 4 * (it % T::R4);
    v[u] = w4_load16(xb, (unsigned)(row * pm.x_ld + ((interior || (tg >= 0 && tg + 3 < L)) ? tg : 0)) * 4u, so);
  }
}
// one member of one tile, producer side: c1's four stages from global (v holds stage 0's rows on entry), then - behind the consumers' hand-over
// barrier - c2's four stages from the intermediate tile; next(): requests whatever follows this member's c1 (the next member's, or the next tile's, first rows)
template <class T, class Next>
__device__ __forceinline__ void pairacc_produce(const PairMember& pm, const PairGroup& g, float4 (&v)[T::SPW], float* const mid, float* const pl, const int PLFMAX,
                                                const int bz, const int n2, int& s_, const int lane, const int pw_, Next&& next) {
  using G1 = typename T::G1;
  using G2 = typename T::G2;
  constexpr int D1 = G1::DIL, RPW = T::RPW, RAW1 = T::RAW1, R4 = T::R4, NGW = T::NGW, SPW = T::SPW, NE1 = T::NE1, NIW1 = T::NIW1, TPW1 = T::TPW1, PQ1 = T::PQ1;
  constexpr int NE2 = T::NE2, NIW2 = T::NIW2, TPW2 = T::TPW2, PQ2 = T::PQ2, MIDS = T::PG::MIDS, KS = T::KS;
  float* const raw = mid;
  const int L = g.L;
  const float slope = g.slope;
  int m0, off2, xs1, lead1;
  pairacc_origin<T>(n2, m0, off2, xs1, lead1);
  const bool interior = xs1 >= 0 && xs1 + RAW1 <= L;
  const int l_ = lane;                                      // (no barrier against hoisting the member's per-lane constants over the tile loop: conv_wino4_acc.hip)
  float* rdst[SPW];
#pragma unroll
  for (int u = 0; u < SPW; ++u) {
    const int it = min(l_ + 64 * u, NGW - 1);
    rdst[u] = raw + (RPW * pw_ + it / R4) * RAW1 + 4 * (it % R4);
  }
  // byte offsets of this lane's sixteen-byte groups inside a stage's rows: the same for the member's four stages (pairacc_issue re-derives them per call; as in
  // conv_wino4_acc.hip that was ~50 producer instructions per stage)
  unsigned voff[SPW];
#pragma unroll
  for (int u = 0; u < SPW; ++u) {
    const int it = min(l_ + 64 * u, NGW - 1);
    const int row = RPW * pw_ + it / R4, tg = xs1 + 4 * (it % R4);
    voff[u] = (unsigned)(row * pm.x_ld + ((interior || (tg >= 0 && tg + 3 < L)) ? tg : 0)) * 4u;
  }
  const float* const xb_ = pm.x + (long long)bz * pm.x_bs;
  const int sstep = KS * pm.x_ld * 4;
  int t1off[TPW1], t1dst[TPW1];
#pragma unroll
  for (int u = 0; u < TPW1; ++u) {
    const int it = min(l_ + 64 * u, NIW1 - 1);
    const int row = RPW * pw_ + it / NE1, e = it % NE1;
    const int qe = e / D1, pe = e - qe * D1;
    t1off[u] = row * RAW1 + 4 * D1 * qe + pe;
    t1dst[u] = row * PQ1 + e;
    asm volatile("" : "+v"(t1off[u]), "+v"(t1dst[u]));
  }
  // ---------------- phase A: c1's four stages (8 channels each) from global
  for (int ch = 0; ch < 4; ++ch) {
#pragma unroll
    for (int u = 0; u < SPW; ++u) {
      if (64 * (u + 1) <= NGW || lane < NGW - 64 * u) {
        float4 q = v[u];
        if (!interior) {                                     // L is a multiple of four: a 16-byte group is inside the row or padding
          const int tg = xs1 + 4 * ((l_ + 64 * u) % R4);
          if (tg < 0 || tg + 3 >= L) q = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        wino_lrelu4(q, slope);
        *reinterpret_cast<float4*>(rdst[u]) = q;
      }
    }
    if (ch + 1 < 4) {
      const int so = (ch + 1) * sstep;
#pragma unroll
      for (int u = 0; u < SPW; ++u) v[u] = w4_load16(xb_, voff[u], so);
    }
    float* const pb = pl + s_ * PLFMAX;
#pragma unroll
    for (int u = 0; u < TPW1; ++u) {
      if (64 * (u + 1) <= NIW1 || lane < NIW1 - 64 * u) {
        const float* r = raw + t1off[u] + lead1;
        w4_transform_window<G1>(pb + t1dst[u], D1 == 1 ? r - G1::LEAD : r);
      }
    }
    __syncthreads();                                         // B_s: plane set complete
    s_ ^= 1;
  }
  __syncthreads();                                           // X: the consumers have written the rows of c2's first stage into the intermediate tile
  next();
  // ---------------- phase B: c2's four stages from the intermediate tile
  int t2src[TPW2], t2dst[TPW2];
#pragma unroll
  for (int u = 0; u < TPW2; ++u) {
    const int it = min(l_ + 64 * u, NIW2 - 1);
    const int row = RPW * pw_ + it / NE2, e = it % NE2;
    t2src[u] = row * MIDS + 4 * e;
    t2dst[u] = row * PQ2 + e;
    asm volatile("" : "+v"(t2src[u]), "+v"(t2dst[u]));
  }
  for (int ch = 0; ch < 4; ++ch) {
    float* const pb = pl + s_ * PLFMAX;
    const float* const mrow = mid + ch * KS * MIDS + off2;
#pragma unroll
    for (int u = 0; u < TPW2; ++u) {
      if (64 * (u + 1) <= NIW2 || lane < NIW2 - 64 * u) w4_transform_window<G2>(pb + t2dst[u], mrow + t2src[u]);
    }
    __syncthreads();
    s_ ^= 1;
  }
}
// one member of one tile, consumer side
template <class T>
__device__ __forceinline__ void pairacc_consume(const PairMember& pm, const PairGroup& g, float* const mid, const unsigned plbase, const int PLFMAX, const int bz,
                                                const int n2, int& s_, const int lane, const int wave) {
  using PG = typename T::PG;
  using G1 = typename T::G1;
  using G2 = typename T::G2;
  constexpr int D1 = G1::DIL, NACC = G1::NACC, NWC1 = PG::NWC1, NW2 = PG::NW2, MIDS = PG::MIDS;
  const int L = g.L;
  const int l31 = lane & 31, hi = lane >> 5;
  const int uu = wave * 32 + l31;                           // this lane's window of the tile (1 x 4 consumers)
  const unsigned wlane = (unsigned)lane * 16u;
  int m0, off2, xs1, lead1;
  pairacc_origin<T>(n2, m0, off2, xs1, lead1);
  f32x16 M[NACC];
  WinoArgs p1{}, p2{};
  p1.wp = pm.wp1; p1.nchunks = 1;
  p2.wp = pm.wp2; p2.nchunks = 1;
  const float slope = g.slope;
  auto init_bias = [&](const float* bias) {                 // the bias starts in M1 (part of all four outputs)
    const float* bq = bias + 4 * hi;
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) M[q][i] = q == 1 ? bq[(i & 3) + 8 * (i >> 2)] : 0.f;
  };
  auto ytrans = [&](auto q_c, float4 (&vo)[4]) {
    constexpr int Q = decltype(q_c)::value;
#pragma unroll
    for (int r = 0; r < 4; r += 2) w4_output_transform2<G1, NACC>(M, 4 * Q + r, vo[r], vo[r + 1]);
  };
  // ---------------- phase A: c1 into the accumulators, lrelu(c1) -> the intermediate tile (zero outside [0, L))
  init_bias(pm.bias1);
  acc3_consume<G1, NACC, 2>(p1, M, plbase, PLFMAX, 0, s_, wlane, (unsigned)(hi * G1::PQ + uu) * 4u);
  {
    const bool act = uu < NWC1;
    const int bq = uu / D1, ph = uu - bq * D1;
    const int c0 = 4 * D1 * bq + ph;
    const int n0 = m0 + c0;
    auto quarter = [&](auto q_c) {
      constexpr int Q = decltype(q_c)::value;
      if (!act) return;
      float4 vo[4];
      ytrans(q_c, vo);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        wino_lrelu4(vo[r], slope);
        float* d = mid + (8 * Q + 4 * hi + r) * MIDS + c0;
        if constexpr (D1 == 1) {
          if (n0 < 0 || n0 >= L) vo[r] = make_float4(0.f, 0.f, 0.f, 0.f);
          *reinterpret_cast<float4*>(d) = vo[r];
        } else {
          d[0] = (n0 >= 0 && n0 < L) ? vo[r].x : 0.f;
          d[D1] = (n0 + D1 >= 0 && n0 + D1 < L) ? vo[r].y : 0.f;
          d[2 * D1] = (n0 + 2 * D1 >= 0 && n0 + 2 * D1 < L) ? vo[r].z : 0.f;
          d[3 * D1] = (n0 + 3 * D1 >= 0 && n0 + 3 * D1 < L) ? vo[r].w : 0.f;
        }
      }
    };
    quarter(std::integral_constant<int, 0>{});
    __syncthreads();                                         // X: rows 0 .. 7 (c2's first stage) are in the intermediate tile (conv_wino4_pair.hip's hand-over)
    quarter(std::integral_constant<int, 1>{});
    quarter(std::integral_constant<int, 2>{});
    quarter(std::integral_constant<int, 3>{});
  }
  // ---------------- phase B: c2, then + x (+ what the previous member stored) (/ div) and store
  init_bias(pm.bias2);
  acc3_consume<G2, NACC, 2>(p2, M, plbase, PLFMAX, 0, s_, wlane, (unsigned)(hi * G2::PQ + uu) * 4u);
  const int ne = n2 + 4 * uu;
  if (uu < NW2 && ne < L) {
    char* const ybase = reinterpret_cast<char*>(pm.y + (long long)bz * pm.y_bs + (long long)(4 * hi) * pm.y_ld + ne);
    const char* const rbase = reinterpret_cast<const char*>(pm.x + (long long)bz * pm.x_bs + (long long)(4 * hi) * pm.x_ld + ne);
    const size_t ylb = (size_t)pm.y_ld * 4, rlb = (size_t)pm.x_ld * 4;
    const bool eacc = (pm.eflags & F_ACC) != 0, ediv = (pm.eflags & F_DIV) != 0;
    const float dv = pm.div, rc = 1.0f / dv;
    auto dv1 = [&](float x) { const float q = x * rc; return __builtin_fmaf(__builtin_fmaf(-q, dv, x), rc, q); };      // x / div (conv_wino4_kernels.h)
    // the residual rows (and, from the second member on, what this lane stored one member earlier) are requested for TWO quarters at a time, ahead of the
    // first of their output transforms (round 6: a quarter at a time the epilogue paid four round trips in a row; all four at once would not fit the registers;
    // the first two requested ahead of c2's MFMA streams, as the pair kernel does: 234 -> 253 registers, launch 1149 -> 1146 us - not kept)
    auto half = [&](auto h_c) {
      constexpr int HQ = decltype(h_c)::value;               // quarters 2 HQ, 2 HQ + 1
      float4 rv[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) rv[r] = *reinterpret_cast<const float4*>(rbase + (size_t)(8 * (2 * HQ + (r >> 2)) + (r & 3)) * rlb);
      if (eacc) {                                            // what this lane stored one member earlier
        float4 yv[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) yv[r] = *reinterpret_cast<const float4*>(ybase + (size_t)(8 * (2 * HQ + (r >> 2)) + (r & 3)) * ylb);
#pragma unroll
        for (int r = 0; r < 8; ++r) w4_add4(rv[r], yv[r]);
      }
      auto quarter = [&](auto q_c) {
        constexpr int Q = decltype(q_c)::value;
        float4 vo[4];
        ytrans(q_c, vo);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          w4_add4(vo[r], rv[4 * (Q - 2 * HQ) + r]);
          if (ediv) vo[r] = make_float4(dv1(vo[r].x), dv1(vo[r].y), dv1(vo[r].z), dv1(vo[r].w));
          *reinterpret_cast<float4*>(ybase + (size_t)(8 * Q + r) * ylb) = vo[r];
        }
      };
      quarter(std::integral_constant<int, 2 * HQ>{});
      quarter(std::integral_constant<int, 2 * HQ + 1>{});
    };
    half(std::integral_constant<int, 0>{});
    half(std::integral_constant<int, 1>{});
  }
}

template <int D1, bool F44>
__global__ void __launch_bounds__(512, 2) conv_wino4_pairacc_kernel(const PairGroup g) {
  using T11 = PairAccT<11, D1, F44>;
  using T7 = PairAccT<7, D1, F44>;
  using T3 = PairAccT<3, D1, F44>;
  constexpr int W2 = T11::PG::W2;
  static_assert(T7::PG::W2 == W2 && T3::PG::W2 == W2, "one tile space");
  constexpr int MIDF = T11::PG::MID_FLOATS;
  static_assert(T7::PG::MID_FLOATS == MIDF && T3::PG::MID_FLOATS == MIDF, "one intermediate tile");
  constexpr int PLFMAX = T11::PG::cmax(T11::PG::PLFMAX, T11::PG::cmax(T7::PG::PLFMAX, T3::PG::PLFMAX));
  extern __shared__ __attribute__((aligned(16))) float wl[];
  float* const mid = wl;
  float* const pl = wl + MIDF;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int total = g.end[0], stride = gridDim.x, v0 = blockIdx.x;
  if (v0 >= total) return;
  const int my_tiles = (total - v0 + stride - 1) / stride;
  const int L = g.L;
  const int ntn = (L + W2 - 1) / W2;
  auto locate = [&](int v, int& bz_, int& n2_) {
    const int tl = xcd_linear(v, total, g.xcd);
    bz_ = tl / ntn;
    n2_ = (tl - bz_ * ntn) * W2;
  };
  int s_ = 0;
  if (wave >= 4) {
    const int pw_ = wave - 4;
    __builtin_amdgcn_s_setprio(3);
    float4 v11[T11::SPW], v7[T7::SPW], v3[T3::SPW];
    int bz, n2;
    locate(v0, bz, n2);
    auto first_rows = [&](auto tc, auto& v, const PairMember& pm, int bz_, int n2_) {
      using T = typename decltype(tc)::type;
      int m0, off2, xs1, lead1;
      pairacc_origin<T>(n2_, m0, off2, xs1, lead1);
      pairacc_issue<T>(v, pm, L, bz_, xs1, 0, lane, pw_);
    };
    first_rows(PairTag<T11>{}, v11, g.m[0], bz, n2);
    for (int ti = 0; ti < my_tiles; ++ti) {
      int bzn = bz, n2n = n2;
      const bool more = ti + 1 < my_tiles;
      if (more) locate(v0 + (ti + 1) * stride, bzn, n2n);
      pairacc_produce<T11>(g.m[0], g, v11, mid, pl, PLFMAX, bz, n2, s_, lane, pw_, [&]() { first_rows(PairTag<T7>{}, v7, g.m[1], bz, n2); });
      pairacc_produce<T7>(g.m[1], g, v7, mid, pl, PLFMAX, bz, n2, s_, lane, pw_, [&]() { first_rows(PairTag<T3>{}, v3, g.m[2], bz, n2); });
      pairacc_produce<T3>(g.m[2], g, v3, mid, pl, PLFMAX, bz, n2, s_, lane, pw_, [&]() { if (more) first_rows(PairTag<T11>{}, v11, g.m[0], bzn, n2n); });
      bz = bzn; n2 = n2n;
    }
    return;
  }
  const unsigned plbase = (unsigned)(size_t)pl;
  for (int ti = 0; ti < my_tiles; ++ti) {
    int bz, n2;
    locate(v0 + ti * stride, bz, n2);
    pairacc_consume<T11>(g.m[0], g, mid, plbase, PLFMAX, bz, n2, s_, lane, wave);
    pairacc_consume<T7>(g.m[1], g, mid, plbase, PLFMAX, bz, n2, s_, lane, wave);
    pairacc_consume<T3>(g.m[2], g, mid, plbase, PLFMAX, bz, n2, s_, lane, wave);
  }
}
template <int D1, bool F44>
static int pairacc_launch_d(PairGroup& g, hipStream_t st) {
  using T11 = PairAccT<11, D1, F44>;
  constexpr int PLFMAX = T11::PG::cmax(T11::PG::PLFMAX, T11::PG::cmax(PairAccT<7, D1, F44>::PG::PLFMAX, PairAccT<3, D1, F44>::PG::PLFMAX));
  constexpr int lds = (T11::PG::MID_FLOATS + 2 * PLFMAX) * 4;
  static_assert(lds <= 160 * 1024, "tile does not fit");
  const long long total = (long long)((g.L + T11::PG::W2 - 1) / T11::PG::W2) * g.B;
  if (total > 0x7fffffffLL) return 1;
  g.end[0] = g.end[1] = g.end[2] = (int)total;
  auto kern = conv_wino4_pairacc_kernel<D1, F44>;
  SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
  const unsigned grid = (unsigned)std::min<long long>(total, (long long)device_cu_count());
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), (size_t)lds, st, g);
  return SVOC_OK;
}

template <int D1, int NRT, bool F44>
static int pair_launch_d(PairGroup& g, hipStream_t st) {
  using P11 = PairGeo<11, D1, NRT, F44>;
  constexpr int lds = P11::cmax(P11::LDS_FLOATS, P11::cmax(PairGeo<7, D1, NRT, F44>::LDS_FLOATS, PairGeo<3, D1, NRT, F44>::LDS_FLOATS)) * 4;
  static_assert(lds <= 160 * 1024, "tile does not fit");
  const int w2[3] = {P11::W2, PairGeo<7, D1, NRT, F44>::W2, PairGeo<3, D1, NRT, F44>::W2};
  long long total = 0;
  for (int i = 0; i < 3; ++i) {
    total += (long long)((g.L + w2[i] - 1) / w2[i]) * g.B;
    if (total > 0x7fffffffLL) return 1;
    g.end[i] = (int)total;
  }
  const unsigned grid = (unsigned)std::min<long long>(total, (long long)device_cu_count());
  if (g.dbg && (D1 == 1 || (D1 == 3 && NRT == 1))) {        // stamped build (tools/pair_timeline.py): the launches the model takes
    if constexpr (D1 == 1 || (D1 == 3 && NRT == 1)) {
      auto kern = conv_wino4_pair_kernel<D1, NRT, F44, true>;
      SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
      hipLaunchKernelGGL(kern, dim3(grid), dim3(512), (size_t)lds, st, g);
      return SVOC_OK;
    }
  }
  auto kern = conv_wino4_pair_kernel<D1, NRT, F44>;
  SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), (size_t)lds, st, g);
  return SVOC_OK;
}

bool wino4_pair_enabled() {
  return wino4_c32_enabled();
}
// tiles a launch would have (the engine's size gate)
bool wino44_enabled();
long long wino4_pair_tiles(int C, int L, int B, int D1) {
  const int w = C == 32 ? (D1 == 1 ? PairGeo<11, 1, 1, true>::W2 : (D1 == 3 ? PairGeo<11, 3, 1, true>::W2 : PairGeo<11, 5, 1, true>::W2)) : PairGeo<11, 1, 2, true>::W2;
  return 3LL * B * ((L + w - 1) / w);
}
// The three chains' c1 (dilation D1) -> c2 pairs of one MRF step at C = 32 / 64; members k = 11, 7, 3.  1 = not eligible.
// accum: the accumulate form (C = 32, F(4,4) images): every y[i] is the SAME tensor, written by the k = 11 member, read-modified-written by the k = 7
// and k = 3 members, the last one dividing by `div` (the number of chains)
int launch_wino4_pair(const PackedWino* const* pw1, const PackedWino* const* pw2, const float* const* x, float* const* y, long long bs, int ld,
                      int B, int L, int D1, float slope, hipStream_t st, int accum, float div) {
  if (!wino4_pair_enabled() || !wino44_enabled() || B <= 0 || (L & 3) || (ld & 3) || (bs & 3) || !(D1 == 1 || D1 == 3 || D1 == 5)) return 1;
  if (accum && (!wino44_enabled() || y[0] != y[1] || y[1] != y[2] || !(div > 0.f))) return 1;
  if (64LL * ld * 4 >= (1LL << 31)) return 1;               // the producers' staging loads: 32-bit offsets within one batch element
  static const int ks[3] = {11, 7, 3};
  const int C = pw1[0] ? pw1[0]->Cin : 0;
  // C = 64 (64-window tiles: the halo costs more lanes): measured -97 us for the d = 1 pair, -8 us for the d = 3 pair (c2 keeps 57 of 64
  // windows there) - only the first is taken
  if (!(C == 32 || (C == 64 && D1 == 1 && !accum))) return 1;
  PairGroup g{};
  double flops = 0, exec = 0;
  for (int i = 0; i < 3; ++i) {
    if (!pw1[i] || !pw2[i] || pw1[i]->K != ks[i] || pw2[i]->K != ks[i] || pw1[i]->Cin != C || pw1[i]->Cout != C || pw2[i]->Cin != C ||
        pw2[i]->Cout != C || !pw1[i]->wp4.p || !pw2[i]->wp4.p || pw1[i]->f44 != (wino44_enabled() && ks[i] >= 7) || pw2[i]->f44 != pw1[i]->f44) return 1;
    if ((reinterpret_cast<uintptr_t>(x[i]) & 15) || (reinterpret_cast<uintptr_t>(y[i]) & 15)) return 1;
    g.m[i].x = x[i]; g.m[i].x_bs = bs; g.m[i].x_ld = ld;
    g.m[i].y = y[i]; g.m[i].y_bs = bs; g.m[i].y_ld = ld;
    g.m[i].wp1 = pw1[i]->wp4.f(); g.m[i].bias1 = pw1[i]->bias.f();
    g.m[i].wp2 = pw2[i]->wp4.f(); g.m[i].bias2 = pw2[i]->bias.f();
    g.m[i].eflags = accum ? (i == 0 ? 0u : (i == 1 ? (unsigned)F_ACC : (unsigned)(F_ACC | F_DIV))) : 0u;
    g.m[i].div = div;
    if (accum && x[i] == y[i]) return 1;                    // never in place: neighbouring tiles read the input's halo
    const double f = (pw1[i]->flops_per_col + pw2[i]->flops_per_col) * (double)B * (double)L;
    const int G = (ks[i] + 1) / 4;
    flops += f;
    exec += f * (pw1[i]->f44 ? 1.75 * G : 1.5 * G + (G - 1)) / (double)ks[i];
  }
  // tile totals are validated before the launch is counted or profiled: "1 = not eligible" must leave no trace
  if (wino4_pair_tiles(C, L, B, D1) > 0x7fffffffLL) return 1;           // an upper bound of pair_launch_d's total (the k = 11 member has the narrowest tiles)
  g.L = L; g.B = B; g.xcd = xcd_mapping_enabled(); g.slope = slope;
  g.flags = 0u;
  g.dbg = debug_stamp_buffer();
  stats_add_conv(flops, 6, exec);
  int prof_idx = -1;
  if (prof_enabled()) {
    char d[160];
    snprintf(d, sizeof(d), "wino4P Ci%-4d Co%-4d k11/7/3 c1(d%d)+c2%s N%-7d B%-3d%s", C, C, D1, accum ? " accumulate" : "", L, B, wino44_enabled() ? " F(4,4)" : "");
    prof_idx = prof_begin(st, d, flops);
  }
#define SVOC_W4P(F) (C == 32 ? (D1 == 1 ? pair_launch_d<1, 1, F>(g, st) : (D1 == 3 ? pair_launch_d<3, 1, F>(g, st) : pair_launch_d<5, 1, F>(g, st))) \
                             : pair_launch_d<1, 2, F>(g, st))
  const int rc = accum ? (D1 == 1 ? pairacc_launch_d<1, true>(g, st) : (D1 == 3 ? pairacc_launch_d<3, true>(g, st) : pairacc_launch_d<5, true>(g, st)))
                       : SVOC_W4P(true);
#undef SVOC_W4P
  prof_end(st, prof_idx);
  if (rc != SVOC_OK) return rc < 0 ? rc : SVOC_ERR_UNSUPPORTED;
  SVOC_HIP(hipGetLastError());
  return SVOC_OK;
}

}  // namespace svoc
