// Geometry of the Winograd F(4,3) kernels (conv_wino4.hip: one convolution per persistent launch or three grouped;
// conv_wino4_acc.hip: the three MRF chains' last convolutions merged into one set of accumulators).
#pragma once
#include "svoc_internal.h"
#include "wino_common.h"

namespace svoc {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// NRT = row tiles per workgroup.  The four consumers are 4 x 1 (NRT = 4: 128 rows x 32 windows, C % 128 == 0) or 2 x 2
// (NRT = 2: 64 rows x 64 windows, C = 64) - always one consumer per SIMD (a first 64-row variant with two consumers and two
// producers per workgroup, two workgroups per CU, put both workgroups' consumers on the same two SIMDs and ran at 45 % of the
// pipe: profiles/r03_f_winograd_f43_c64_null.txt).  Channels per stage KS: 32 (one weight chunk); 64 for k = 3 with NRT = 4
// (only 96 MFMAs per chunk and consumer: two chunks per stage; the plane sets of k = 7 / 11 would not fit twice); with NRT = 2
// the planes are twice as wide, so k = 7 / 11 stage 16 channels = half a weight chunk (k-groups {0,1} or {2,3} of every slot).
// NRT = 1 (round 4: C = 32, the last MRF stage): 1 x 4 consumers, 32 rows x 128 windows = 512 outputs per workgroup tile; one row
// tile feeds on planes four column tiles wide, so a stage is 16 channels (k = 3: k-groups {0,1} / {2,3}) or 8 channels (k = 7 / 11:
// one k-group of every slot, four stages per weight chunk) - 48 / 64 / 104 MFMAs per stage and consumer.
#ifndef W4_NPS3_MAX_NRT
#define W4_NPS3_MAX_NRT 4                                  // three plane sets for every layout whose LDS budget allows them
#endif
// KSDIV: channels per stage divided by KSDIV (conv_wino4_acc.hip stages its k = 3 member half as wide so that three plane sets fit)
//
// F44 (round 4): the groups are FOUR-tap groups at tap offsets 0, 4, 8 in minimal F(4,4) form - four outputs from the seven inputs
// d_j = x[4q - pad + 4g + j] with SEVEN products (Cook-Toom on the points 0, +-1, +-2, +-1/2; the last group of k = 7 / 11 carries a
// zero fourth tap), no left-over taps: 7 G products per window instead of 6 G + 4 (G - 1) - k = 7: 14 instead of 16, k = 11: 21 instead
// of 26 (k = 3: 7 instead of 6, used only where a k = 3 member has to share accumulators with the others).  The window step still
// equals the group spacing, so every group reads the same seven planes V_p[c][q + g]:
//     V0 = (d0 - d6) + 5.25 (d4 - d2)                                   U0 = w0
//     V1,2 = (d2 + d6 - 4.25 d4) +- (d1 + d5 - 4.25 d3)                 U1,2 = -2/9 ((w0 + w2) +- (w1 + w3))
//     V3,4 = (d6 + 0.25 d2 - 1.25 d4) +- (0.5 d1 - 2.5 d3 + 2 d5)       U3,4 = 1/90 ((w0 + 4 w2) +- (2 w1 + 8 w3))
//     V5,6 = (d6 + 4 d2 - 5 d4) +- (2 d1 - 2.5 d3 + 0.5 d5)             U5,6 = 32/45 ((w0 + w2 / 4) +- (w1 / 2 + w3 / 8))
//     y_i = sum_p a_p^i M_p,  a = (0, 1, -1, 2, -2, 1/2, -1/2),  M_p = sum_c U_p[c] V_p[c]
// fp32 error of one C = 192 convolution against float64: 1.03e-6 relative RMS (F(4,3): 0.86e-6, direct form 0.26e-6; tools/wino_numerics.py).
template <int K, int D, int NRT = 4, int PERM = 0, int KSDIV = 1, bool F44_ = false>
struct W4Geo {
  static constexpr bool F44 = F44_;
  static constexpr int NV = F44 ? 7 : 6;                  // transformed planes = products per group and window
  static_assert(PERM == 0 || D == 1, "a window-major input belongs to an undilated convolution (the c2 behind a dilated c1)");
  static constexpr int DIL = D;
  static constexpr int NCT = 4 / NRT;                     // column tiles (of 32 windows) per workgroup
  static constexpr int KS = (NRT == 1 ? (K == 3 ? 16 : 8) : (NRT == 2 ? (K == 3 ? 32 : 16) : ((K == 3 && D == 1 && !F44_) ? 64 : 32))) / KSDIV;   // channels per stage (F44: seven slots per chunk, one chunk per stage)
  static_assert(KS >= 8, "a stage is at least one k-group");
  static constexpr int CPS = KS >= KC ? KS / KC : 1;      // weight chunks per stage
  static constexpr int HALVES = KS < KC ? KC / KS : 1;    // stages per weight chunk
  static constexpr int KGS = KS < KC ? KS / 8 : 4;        // k-groups (of 8 channels = 4 k-steps) per stage and chunk
  // Weight registers: a slot of a stage is KGS float4 per lane = 4 KGS MFMAs.  With KGS = 4 the next slot is requested one slot
  // (1024 cycles of MFMAs) ahead in the other of two register sets; with KGS = 2 / 1 one slot is only 512 / 256 cycles - less than a
  // loaded L2 round trip (round 4: 77-84 cycles per MFMA in the C = 32 streams) - so those run a ring of four sets, three slots ahead.
  // (k = 3 with KGS = 4 on the ring of four as well - its streams run at 75 cycles per MFMA against 68 for k = 7 / 11 - measured: alone
  // 75.6 -> 74.1 at C = 128, 92.1 -> 85.7 at C = 64, inside the grouped launches nothing; not taken.)
  static constexpr int NSET = KGS == 4 ? 2 : 4;
  static constexpr int PD = NSET - 1;                     // slots ahead
  static constexpr int G = (K + 1) / 4;                   // three-tap (F44: four-tap) groups at tap offsets 0, 4, 8
  static constexpr int ND = F44 ? 0 : G - 1;              // left-over single taps (3, 7)
  static constexpr int PADT = (K - 1) / 2;                // padding in taps (columns: PADT * D)
  static constexpr int WSLOTS = NV * G + ND;              // weight slots per 32-channel chunk
  static constexpr int NGS = NV * G * KGS;                // steps (4 MFMAs each) of the groups; a tap adds 4 * KGS steps
  static constexpr int NSTEP = NGS + 4 * KGS * ND;
  // Windows are numbered along a row: window w = D * b + ph (q block b, phase ph) owns the outputs 4 D b + ph + r D, r = 0..3.  A
  // workgroup tile is NWT = 32 NCT CONSECUTIVE windows [w0, w0 + NWT): with D = 1 that is the output range [4 w0, 4 w0 + 4 NWT); with
  // D > 1 a tile may start inside a q block (ph0 = w0 % D), so every lane has a window (a first version gave a tile (32 / D) D
  // windows: 30 of 32 lanes, 7 % more tiles).  Entry e of a plane row = window w0 + e; group g reads entry e + g D.
  static constexpr int NWT = 32 * NCT;                    // windows per workgroup tile
  static constexpr int W = 4 * NWT;                       // D = 1: output columns per workgroup tile
  static constexpr int NE = NWT + (G - 1) * D;            // windows staged per row and stage
  // plane row stride: NE rounded up so that a plane (KS rows) is a multiple of 64 floats - the producers' six to ten stores of a window
  // (one per plane, PLANE floats apart) then pair up as ds_write2st64_b32 (offsets in units of 256 bytes): half the store instructions
  // of the producer-bound 64- / 32-row layouts (round 4; the 128-row layout's planes happened to be multiples of 64 already)
  static constexpr int PQA = KS >= 64 ? 1 : 64 / KS;
  static constexpr int PQ = (NE + PQA - 1) / PQA * PQA;
  static constexpr int XOFF = -((PADT * D + 3) & ~3);     // D = 1: raw tile starts at 4 w0 + XOFF (multiple of 4)
  static constexpr int LEAD = -XOFF - PADT * D;           // D = 1: raw index of d0 of window w0
  // raw tile columns.  D = 1: d5 (F44: d6) of the last window + 1.  D > 1: first sample f(w0) = 4 D b0 + ph0 - PADT D rounded down to a
  // multiple of 4 (lead <= 3); f grows by at most 4 NE + 5 D over NE windows and a window spans 5 D (F44: 6 D) more
  static constexpr int RAW = D == 1 ? ((LEAD + 4 * (NE - 1) + (NV - 1) + 1 + 3) & ~3) : ((4 * NE + (NV + 4) * D + 4 + 3) & ~3);
  static constexpr int NPL = F44 ? 7 : (ND > 0 ? 10 : 6); // V0..V5 (+ X0..X3); F44: V0..V6
  static constexpr int NACC = F44 ? 7 : (ND > 0 ? 8 : 6);
  static constexpr int PLANE = KS * PQ;
  static constexpr int PLF = NPL * PLANE;                 // floats per plane set
  // PERM = P > 0 (round 4): the INPUT rows are in the window-major order a dilation-P convolution's epilogue writes with 16-byte
  // stores (P[4 w + r] = y[4 P b + ph + r P], w = P b + ph): the producers load whole q blocks and scatter each group's four samples
  // P columns apart into the raw tile, whose rows get slack on both sides for the columns of those blocks outside [xs, xs + RAW)
  static constexpr int PORG = PERM > 0 ? 4 * PERM : 0;                           // slack ahead of column xs
  static constexpr int RAWS = PERM > 0 ? ((RAW + 12 * PERM + 3) & ~3) : RAW;     // raw row stride
  static constexpr int PNBLK = PERM > 0 ? (RAW + 4 * PERM - 2) / (4 * PERM) + 1 : 0;   // q blocks that a RAW-wide range can touch
  static constexpr int PNG = PERM * PNBLK;                                       // 16-byte groups loaded per row and stage
  static constexpr int RAW_FLOATS = KS * RAWS;
  // plane sets: two (producers one stage ahead); three where they fit (k = 7 / 11 at dilation 1 in every layout, the dilated ones where the
  // wider planes allow: the producers run two stages ahead, so that they work through the consumers' epilogue and a late stage does not
  // stall the streams)
  static constexpr int NPS = ((NRT <= W4_NPS3_MAX_NRT) && (RAW_FLOATS + 3 * PLF) * 4 <= 160 * 1024) ? 3 : 2;
  static constexpr int LDS_BYTES = (RAW_FLOATS + NPS * PLF) * 4;
  // step t of a chunk: which weight slot, plane, column (in windows) and accumulator
  static constexpr bool tap(int t) { return t >= NGS; }
  static constexpr int tr(int t) { return ((t - NGS) % (4 * KGS)) / KGS; }               // tap steps: output index r
  static constexpr int kgi(int t) { return t % KGS; }                                    // k-group of the step inside the stage
  static constexpr bool slot_first(int t) { return t < NGS ? t % KGS == 0 : (t - NGS) % (4 * KGS) == 0; }
  static constexpr int wslot(int t) { return t < NGS ? t / KGS : 6 * G + (t - NGS) / (4 * KGS); }
  static constexpr int plane(int t) { return t < NGS ? (t / KGS) % NV : 6 + (tr(t) + 2) % 4; }
  static constexpr int colq(int t) { return t < NGS ? (t / KGS) / NV : (t - NGS) / (4 * KGS) + (tr(t) + 2) / 4; }
  static constexpr int acc(int t) { return t < NGS ? (t / KGS) % NV : (tr(t) == 0 ? 0 : (tr(t) == 3 ? 5 : 5 + tr(t))); }
};

// Sixteen bytes at base + soff + voff: the per-lane part of the address is a 32-bit VGPR offset, everything else sits in SGPRs (buffer
// descriptor + scalar offset), so the producers' staging loads cost no VALU address arithmetic (as pointers they were a 64-bit
// v_lshl_add_u64 per load and stage).  base / soff wave-uniform, base + soff + voff within 2 GB of base (host: wino4_args).
__device__ __forceinline__ float4 w4_load16(const void* base, const unsigned voff, const int soff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
  const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0);
  return *reinterpret_cast<const float4*>(&t);
}

// (o, e) -> (e + o, e - o) in one packed instruction: low result = high half + low half, high result = high half - low half
__device__ __forceinline__ f32x2 w4_pk_sum_diff(const f32x2 oe) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(oe));
  return r;
}

// ---- the transforms, shared by every kernel of the family (conv_wino4_kernels.h, conv_wino4_acc.hip, conv_wino4_pair.hip)
// input transform of one window: d0 .. d5 (F44: d6) -> the planes of plane set `o` (this item's entry), PLANE floats apart
template <class Geo>
__device__ __forceinline__ void w4_input_transform(float* const o, const float d0, const float d1, const float d2, const float d3, const float d4,
                                                   const float d5, const float d6) {
  constexpr int PLANE = Geo::PLANE;
  if constexpr (Geo::F44) {
    // Packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two lanes' worth per issue slot): the samples pair up as (odd, even)
    // = (d1, d2), (d3, d4), (d5, d6) - adjacent registers of the window's 16- and 8-byte LDS reads - so each (o_i, e_i) is two or three
    // packed instructions and each (e_i + o_i, e_i - o_i) one: 14 VALU instructions per window instead of 23.  Every producer VALU
    // instruction costs the matrix pipe of its SIMD ~6.5 cycles, which is what this buys back.
    const f32x2 p12 = {d1, d2}, p34 = {d3, d4}, p56 = {d5, d6};
    const f32x2 s1 = __builtin_elementwise_fma(p34, (f32x2){-4.25f, -4.25f}, p12 + p56);
    const f32x2 s2 = __builtin_elementwise_fma(p56, (f32x2){2.f, 1.f}, __builtin_elementwise_fma(p34, (f32x2){-2.5f, -1.25f}, p12 * (f32x2){0.5f, 0.25f}));
    const f32x2 s3 = __builtin_elementwise_fma(p56, (f32x2){0.5f, 1.f}, __builtin_elementwise_fma(p34, (f32x2){-2.5f, -5.f}, p12 * (f32x2){2.f, 4.f}));
    const f32x2 v12 = w4_pk_sum_diff(s1), v34 = w4_pk_sum_diff(s2), v56 = w4_pk_sum_diff(s3);
    o[0] = __builtin_fmaf(5.25f, d4 - d2, d0 - d6);
    o[PLANE] = v12.x;
    o[2 * PLANE] = v12.y;
    o[3 * PLANE] = v34.x;
    o[4 * PLANE] = v34.y;
    o[5 * PLANE] = v56.x;
    o[6 * PLANE] = v56.y;
  } else {
    // packed likewise: (b, a) = (d3 - 4 d1, d4 - 4 d2), (e, c) = (2 (d3 - d1), d4 - d2): 9 VALU instructions instead of 12
    const f32x2 p12 = {d1, d2}, p34 = {d3, d4};
    const f32x2 ba = __builtin_elementwise_fma(p12, (f32x2){-4.f, -4.f}, p34);
    const f32x2 ec = __builtin_elementwise_fma(p34, (f32x2){2.f, 1.f}, p12 * (f32x2){-2.f, -1.f});
    const f32x2 v12 = w4_pk_sum_diff(ba), v34 = w4_pk_sum_diff(ec);
    o[0] = __builtin_fmaf(4.f, d0, __builtin_fmaf(-5.f, d2, d4));
    o[PLANE] = v12.x;
    o[2 * PLANE] = v12.y;
    o[3 * PLANE] = v34.x;
    o[4 * PLANE] = v34.y;
    o[5 * PLANE] = __builtin_fmaf(4.f, d1, __builtin_fmaf(-5.f, d3, d5));
    if constexpr (Geo::ND > 0) { o[6 * PLANE] = d1; o[7 * PLANE] = d2; o[8 * PLANE] = d3; o[9 * PLANE] = d4; }
  }
}
// the window of item `r` (D = 1: r points at the sixteen-byte group LEAD columns ahead of d0, conv_wino4_kernels.h) -> planes
template <class Geo>
__device__ __forceinline__ void w4_transform_window(float* const o, const float* const r) {
  constexpr int D = Geo::DIL, LEAD = Geo::LEAD;
  if constexpr (D == 1 && LEAD == 3) {
    const float4 fm = *reinterpret_cast<const float4*>(r + 4);
    if constexpr (Geo::F44) {
      const float2 ft = *reinterpret_cast<const float2*>(r + 8);
      w4_input_transform<Geo>(o, r[3], fm.x, fm.y, fm.z, fm.w, ft.x, ft.y);
    } else w4_input_transform<Geo>(o, r[3], fm.x, fm.y, fm.z, fm.w, r[8], 0.f);
  } else if constexpr (D == 1) {
    static_assert(D != 1 || LEAD == 3 || LEAD == 1, "window alignment");
    const float4 fa = *reinterpret_cast<const float4*>(r), fb = *reinterpret_cast<const float4*>(r + 4);
    w4_input_transform<Geo>(o, fa.y, fa.z, fa.w, fb.x, fb.y, fb.z, fb.w);
  } else {
    w4_input_transform<Geo>(o, r[0], r[D], r[2 * D], r[3 * D], r[4 * D], r[5 * D], Geo::F44 ? r[6 * D] : 0.f);
  }
}
// output transform of accumulator element i: the four outputs of the lane's window in row i
template <class Geo, int NACC>
__device__ __forceinline__ float4 w4_output_transform(const f32x16 (&M)[NACC], const int i) {
  const float t1 = M[1][i] + M[2][i], t2 = M[1][i] - M[2][i], t3 = M[3][i] + M[4][i], t4 = M[3][i] - M[4][i];
  if constexpr (Geo::F44) {
    static_assert(NACC >= 7, "seven products");
    const float t5 = M[5][i] + M[6][i], t6 = M[5][i] - M[6][i];
    const float y0 = (M[0][i] + t1) + (t3 + t5);
    const float y1 = __builtin_fmaf(0.5f, t6, __builtin_fmaf(2.f, t4, t2));
    const float y2 = __builtin_fmaf(0.25f, t5, __builtin_fmaf(4.f, t3, t1));
    const float y3 = __builtin_fmaf(0.125f, t6, __builtin_fmaf(8.f, t4, t2));
    return make_float4(y0, y1, y2, y3);
  } else {
    const float y0 = M[0][i] + (t1 + t3);
    float y1 = __builtin_fmaf(2.f, t4, t2);
    float y2 = __builtin_fmaf(4.f, t3, t1);
    const float y3 = __builtin_fmaf(8.f, t4, t2) + M[5][i];
    if constexpr (Geo::ND > 0 || NACC == 8) { y1 += M[6][i]; y2 += M[7][i]; }
    return make_float4(y0, y1, y2, y3);
  }
}

// The same for accumulator elements i and i + 1 (i even: an aligned register pair of every accumulator) in packed fp32 math (round 5): the output
// transform is VALU work on the consumers' SIMDs, where it cannot overlap anybody's MFMAs - 15 packed instructions per two rows instead of 30 (F(4,4));
// the same operations in the same order on each row, so results are bit-identical to w4_output_transform.
template <class Geo, int NACC>
__device__ __forceinline__ void w4_output_transform2(const f32x16 (&M)[NACC], const int i, float4& ya, float4& yb) {
  auto P = [&](int p) -> f32x2 { return (f32x2){M[p][i], M[p][i + 1]}; };
  auto fma2 = [](float c, f32x2 a, f32x2 b) -> f32x2 { return __builtin_elementwise_fma((f32x2){c, c}, a, b); };
  const f32x2 t1 = P(1) + P(2), t2 = P(1) - P(2), t3 = P(3) + P(4), t4 = P(3) - P(4);
  f32x2 y0, y1, y2, y3;
  if constexpr (Geo::F44) {
    static_assert(NACC >= 7, "seven products");
    const f32x2 t5 = P(5) + P(6), t6 = P(5) - P(6);
    y0 = (P(0) + t1) + (t3 + t5);
    y1 = fma2(0.5f, t6, fma2(2.f, t4, t2));
    y2 = fma2(0.25f, t5, fma2(4.f, t3, t1));
    y3 = fma2(0.125f, t6, fma2(8.f, t4, t2));
  } else {
    y0 = P(0) + (t1 + t3);
    y1 = fma2(2.f, t4, t2);
    y2 = fma2(4.f, t3, t1);
    y3 = fma2(8.f, t4, t2) + P(5);
    if constexpr (Geo::ND > 0 || NACC == 8) { y1 += P(6); y2 += P(7); }
  }
  ya = make_float4(y0.x, y1.x, y2.x, y3.x);
  yb = make_float4(y0.y, y1.y, y2.y, y3.y);
}
// a += b on the two halves of a float4 (two packed adds)
__device__ __forceinline__ void w4_add4(float4& a, const float4& b) {
  const f32x2 lo = (f32x2){a.x, a.y} + (f32x2){b.x, b.y}, hi = (f32x2){a.z, a.w} + (f32x2){b.z, b.w};
  a = make_float4(lo.x, lo.y, hi.x, hi.y);
}

}  // namespace svoc
