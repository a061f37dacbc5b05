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
template <int K, int D, int NRT = 4, int PERM = 0, int KSDIV = 1>
struct W4Geo {
  static_assert(PERM == 0 || D == 1, "a window-major input belongs to an undilated convolution (the c2 behind a dilated c1)");
  static constexpr int DIL = D;
  static constexpr int NCT = 4 / NRT;                     // column tiles (of 32 windows) per workgroup
  static constexpr int KS = (NRT == 1 ? (K == 3 ? 16 : 8) : (NRT == 2 ? (K == 3 ? 32 : 16) : ((K == 3 && D == 1) ? 64 : 32))) / KSDIV;   // channels per stage
  static_assert(KS >= 8, "a stage is at least one k-group");
  static constexpr int CPS = KS >= KC ? KS / KC : 1;      // weight chunks per stage
  static constexpr int HALVES = KS < KC ? KC / KS : 1;    // stages per weight chunk
  static constexpr int KGS = KS < KC ? KS / 8 : 4;        // k-groups (of 8 channels = 4 k-steps) per stage and chunk
  // Weight registers: a slot of a stage is KGS float4 per lane = 4 KGS MFMAs.  With KGS = 4 the next slot is requested one slot
  // (1024 cycles of MFMAs) ahead in the other of two register sets; with KGS = 2 / 1 one slot is only 512 / 256 cycles - less than a
  // loaded L2 round trip (round 4: 77-84 cycles per MFMA in the C = 32 streams) - so those run a ring of four sets, three slots ahead.
  static constexpr int NSET = KGS == 4 ? 2 : 4;
  static constexpr int PD = NSET - 1;                     // slots ahead
  static constexpr int G = (K + 1) / 4;                   // three-tap groups at tap offsets 0, 4, 8
  static constexpr int ND = G - 1;                        // left-over single taps (3, 7)
  static constexpr int PADT = (K - 1) / 2;                // padding in taps (columns: PADT * D)
  static constexpr int WSLOTS = 6 * G + ND;               // weight slots per 32-channel chunk
  static constexpr int NGS = 6 * G * KGS;                 // steps (4 MFMAs each) of the groups; a tap adds 4 * KGS steps
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
  // raw tile columns.  D = 1: d5 of the last window + 1.  D > 1: first sample f(w0) = 4 D b0 + ph0 - PADT D rounded down to a
  // multiple of 4 (lead <= 3); f grows by at most 4 NE + 5 D over NE windows and a window spans 5 D more
  static constexpr int RAW = D == 1 ? ((LEAD + 4 * (NE - 1) + 5 + 1 + 3) & ~3) : ((4 * NE + 10 * D + 4 + 3) & ~3);
  static constexpr int NPL = ND > 0 ? 10 : 6;             // V0..V5 (+ X0..X3)
  static constexpr int NACC = ND > 0 ? 8 : 6;
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
  static constexpr int plane(int t) { return t < NGS ? (t / KGS) % 6 : 6 + (tr(t) + 2) % 4; }
  static constexpr int colq(int t) { return t < NGS ? (t / KGS) / 6 : (t - NGS) / (4 * KGS) + (tr(t) + 2) / 4; }
  static constexpr int acc(int t) { return t < NGS ? (t / KGS) % 6 : (tr(t) == 0 ? 0 : (tr(t) == 3 ? 5 : 5 + tr(t))); }
};

}  // namespace svoc
