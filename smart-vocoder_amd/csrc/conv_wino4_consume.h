// The consumer side of one member (one convolution's stages of one tile) of the F(4,3) kernels that run several convolutions per
// tile: conv_wino4_acc.hip (the three chains' last convolutions in one set of accumulators) and conv_wino4_pair.hip (c1 -> c2 of a
// ResBlock iteration with the intermediate tile in LDS).  The stream is conv_wino4.hip's.
#pragma once
#include "conv_wino4.h"

namespace svoc {

// ------------------------------------------------------------------------------------------------ consumer side
// One member of one tile: its stages' MFMA streams into the shared accumulators (conv_wino4.hip's stream: fragment reads two steps
// ahead at immediate LDS offsets, weights through buffer loads with SGPR slot offsets, one or three slots ahead).
// DBG (stamped instantiations, tools/pair_timeline.py): cyc[0] += cycles waited at the stage barriers, cyc[1] += cycles of the MFMA streams
template <class Geo, int NACC, int NPS, bool DBG = false>
__device__ __forceinline__ void acc3_consume(const WinoArgs& p, f32x16 (&M)[NACC], const unsigned plbase, const int PLFMAX_, const int wt, int& s_,
                                            const unsigned wlane, const unsigned lanefrag, long long* const cyc = nullptr) {
  constexpr int PQ = Geo::PQ, WSLOTS = Geo::WSLOTS, PLANE = Geo::PLANE, NSTEP = Geo::NSTEP, CPS = Geo::CPS, HALVES = Geo::HALVES, KGS = Geo::KGS;
  constexpr int NSET = Geo::NSET, PD = Geo::PD;
  const int nst = p.nchunks * HALVES / CPS;
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wp), 0, 0x7fffffff, 0x00020000);
  float4 a[NSET][KGS];
  auto wload = [&](float4& dst, int soff, auto kg_c) {
    constexpr int KGO = decltype(kg_c)::value * 1024;
    const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)wlane + KGO, soff, 0);
    dst = *reinterpret_cast<const float4*>(&t);
  };
  auto wload4 = [&](float4 (&dst)[KGS], int soff, auto hf) {
    constexpr int K0 = decltype(hf)::value * KGS;
    wload(dst[0], soff, std::integral_constant<int, K0>{});
    if constexpr (KGS >= 2) wload(dst[1], soff, std::integral_constant<int, K0 + 1>{});
    if constexpr (KGS == 4) { wload(dst[2], soff, std::integral_constant<int, K0 + 2>{}); wload(dst[3], soff, std::integral_constant<int, K0 + 3>{}); }
  };
  // the member's first slots (requested here, right behind the previous member's last MFMA)
  wload4(a[0], wt, std::integral_constant<int, 0>{});
  if constexpr (PD > 1) { wload4(a[1], wt + 4096, std::integral_constant<int, 0>{}); wload4(a[2], wt + 2 * 4096, std::integral_constant<int, 0>{}); }
  auto mfma_chunk = [&](const unsigned baddr, const int wa, const int wnext, auto par, auto cc, auto hf) {
    constexpr int PAR = decltype(par)::value, CC = decltype(cc)::value, HF = decltype(hf)::value;
    constexpr int NHF = HF + 1 < HALVES ? HF + 1 : 0;
    float fb[2][4];
    auto request = [&](auto tc) {
      constexpr int T = decltype(tc)::value;
      if constexpr (T < NSTEP) wino_frag<PQ, Geo::plane(T) * PLANE + CC * KC * PQ, Geo::kgi(T), Geo::colq(T) * Geo::DIL>(fb[T & 1], baddr);
    };
    auto step = [&](auto tc) {
      constexpr int T = decltype(tc)::value;
      constexpr int WS = Geo::wslot(T), KG = Geo::kgi(T);
      if constexpr (Geo::slot_first(T)) {
        if constexpr (WS + PD < WSLOTS) wload4(a[(PAR + WS + PD) % NSET], wa + (WS + PD) * 4096, hf);
        else if (wnext >= 0) wload4(a[(PAR + WS + PD) % NSET], wnext + (WS + PD - WSLOTS) * 4096, std::integral_constant<int, NHF>{});
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
  auto stage = [&](int st_, auto par, auto hf) {
    constexpr int HF = decltype(hf)::value;
    long long c0 = 0, c1 = 0;
    if constexpr (DBG) c0 = (long long)__builtin_readcyclecounter();
    __syncthreads();                                       // B_s: plane set s & 1 is complete, the other one may be overwritten
    if constexpr (DBG) c1 = (long long)__builtin_readcyclecounter();
    const unsigned baddr = plbase + (unsigned)(s_ * PLFMAX_) * 4u + lanefrag;
    s_ = s_ + 1 == NPS ? 0 : s_ + 1;
    constexpr auto c0_ = std::integral_constant<int, 0>{};
    if constexpr (HALVES > 1) {
      const int wa = wt + (st_ / HALVES) * WSLOTS * 4096;
      const int wnext = HF + 1 < HALVES ? wa : (st_ + 1 < nst ? wa + WSLOTS * 4096 : -1);
      mfma_chunk(baddr, wa, wnext, par, c0_, hf);
    } else if constexpr (CPS == 1) {
      const int wa = wt + st_ * WSLOTS * 4096;
      const int wnext = st_ + 1 < nst ? wa + WSLOTS * 4096 : -1;
      mfma_chunk(baddr, wa, wnext, par, c0_, c0_);
    } else {
      static_assert(CPS == 1 || (WSLOTS & 1) == 0, "two chunks per stage need an even slot count");
      const int wa = wt + (st_ * CPS) * WSLOTS * 4096;
      mfma_chunk(baddr, wa, wa + WSLOTS * 4096, par, c0_, c0_);
      const int wnext = st_ + 1 < nst ? wa + 2 * WSLOTS * 4096 : -1;
      mfma_chunk(baddr, wa + WSLOTS * 4096, wnext, par, std::integral_constant<int, CPS - 1>{}, c0_);
    }
    if constexpr (DBG) { cyc[0] += c1 - c0; cyc[1] += (long long)__builtin_readcyclecounter() - c1; }
  };
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
}

}  // namespace svoc
