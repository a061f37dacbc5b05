// Shared by the Winograd convolution kernels (conv_wino.hip: F(2,3); conv_wino4.hip: F(4,3)): launch arguments and the
// small device helpers (immediate-offset LDS fragment reads, static loops, packed leaky relu).
#pragma once
#include "svoc_internal.h"

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
  long long* dbg; int dbg_base;                                     // optional [workgroups][16] stamps (svoc_debug_set_stamp_buffer)
  int out_perm;                                                     // F(4,3), dilation > 1: write rows window-major (conv_wino4.hip)
  unsigned abl;                                                     // stamped build only (SVOC_DBG_ABL): work removed for power / clock ablations (conv_wino4_kernels.h)
};
struct WinoGroup { WinoArgs a[3]; int end[3]; int k[3]; };

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
typedef float f32x2 __attribute__((ext_vector_type(2)));
// leaky relu of four values in 6 instructions (2 packed multiplies + 4 max; fmaxf() costs a canonicalising max more each; round 4:
// the select on the sign bit with integer compare / select pairs instead of the four v_max_f32 is 10 instructions and measured
// SLOWER, 16x512 step 25.8 -> 26.7 ms: profiles/r04_producer_valu_diet.txt)
__device__ __forceinline__ void wino_lrelu4(float4& q, const float slope) {
  const f32x2 s2 = {slope, slope};
  const f32x2 a = (f32x2){q.x, q.y} * s2, b = (f32x2){q.z, q.w} * s2;
  asm("v_max_f32 %0, %1, %2" : "=v"(q.x) : "v"(q.x), "v"(a.x));
  asm("v_max_f32 %0, %1, %2" : "=v"(q.y) : "v"(q.y), "v"(a.y));
  asm("v_max_f32 %0, %1, %2" : "=v"(q.z) : "v"(q.z), "v"(b.x));
  asm("v_max_f32 %0, %1, %2" : "=v"(q.w) : "v"(q.w), "v"(b.y));
}


}  // namespace svoc
