// Shared pieces of the WN layer kernels whose in_layer runs in Winograd F(2,5) form (wn_fused.hip: one launch per layer; wn_stack.hip: one
// persistent launch per WN stack): geometry constants and the VALU-free res_skip GEMM stream.
#pragma once
#include "svoc_internal.h"
#include "wino_common.h"

namespace svoc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float wn_f32x4 __attribute__((ext_vector_type(4)));
constexpr int WNF_H = 192, WNF_XROW = 40, WNF_AROW = 33, WNF_NP = 6, WNF_NQ = 16, WNF_PLANE = WNF_H * WNF_NQ;
constexpr int WNF_KS = 24;                                  // k-steps (of 4 channels) per wave: half of the 192 channels
constexpr int WNF_LDS_FLOATS = WNF_H * WNF_XROW + 6 * WNF_PLANE + 2 * WNF_NP * 16 * 64;

// Compile-time form of wn_gemm for the K-split kernel's common geometry (H = 192: three 32-channel chunks per wave; row stride,
// tap count and dilation fixed): fp32 MFMAs and VALU instructions exclude each other on a SIMD (tools/mfma_valu_probe.hip) and
// the generic loop above issues one VALU instruction per MFMA (fragment addresses from a runtime row stride, 64-bit weight
// pointers), which stretched phase A from 92k to 105k cycles (tools/wn_timeline.py).  Here the stream holds only MFMAs,
// ds_read_b32 at immediate offsets, buffer loads with the group offset in an SGPR, waits and SALU.
//   acc[2] += W[two 32-row tiles][3 chunks x KT taps x 32 channels] * B
// `baddr`: LDS byte address of (row hi of the wave's first chunk, this lane's column for tap 0); `wofs0/1`: byte offsets of the
// two row tiles' first k-step group of that chunk inside the packed image.
typedef unsigned int wn_u32x4 __attribute__((ext_vector_type(4)));
template <int ROWLEN, int KT, int DIL, bool TWO>
__device__ __forceinline__ void wn_gemm_ct(f32x16 (&acc)[2][1], const float* wp, const int wofs0, const int wofs1, const unsigned baddr,
                                           const unsigned wlane) {
  constexpr int NCH = 3, NG = NCH * KT * 4;                // groups of four k-steps (8 channels of one tap)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, 0x7fffffff, 0x00020000);
  auto wload = [&](float4& d, int soff) {
    const wn_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)wlane, soff, 0);
    d = *reinterpret_cast<const float4*>(&t);
  };
  float4 a0[2], a1[2];
  float fb[2][4];
  auto request = [&](auto gc) {
    constexpr int GI = decltype(gc)::value;
    if constexpr (GI < NG) {
      constexpr int CL = GI / (KT * 4), J = (GI / 4) % KT, KG = GI % 4;
      constexpr int O = ((CL * 32 + 8 * KG) * ROWLEN + J * DIL) * 4;
      fb[GI & 1][0] = wino_lds_rd<O>(baddr);
      fb[GI & 1][1] = wino_lds_rd<O + 2 * ROWLEN * 4>(baddr);
      fb[GI & 1][2] = wino_lds_rd<O + 4 * ROWLEN * 4>(baddr);
      fb[GI & 1][3] = wino_lds_rd<O + 6 * ROWLEN * 4>(baddr);
      wload(a0[GI & 1], wofs0 + GI * 1024);
      if constexpr (TWO) wload(a1[GI & 1], wofs1 + GI * 1024);
    }
  };
  auto group = [&](auto gc) {
    constexpr int GI = decltype(gc)::value;
    {
      float(&b)[4] = fb[GI & 1];
      if constexpr (GI + 1 < NG) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
    }
    const float4 av0 = a0[GI & 1], av1 = a1[GI & 1];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wino_pick(av0, s_), fb[GI & 1][s_], acc[0][0], 0, 0, 0);
      if constexpr (TWO) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wino_pick(av1, s_), fb[GI & 1][s_], acc[1][0], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    request(std::integral_constant<int, GI + 2>{});
    __builtin_amdgcn_sched_barrier(0);
  };
  request(std::integral_constant<int, 0>{});
  request(std::integral_constant<int, 1>{});
  wino_static_for<0, NG>(group);
}

}  // namespace svoc
