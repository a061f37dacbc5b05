// One ResBlock1 iteration in one kernel (reference modules.py:211-218):
//     y = c2( lrelu( c1( lrelu(x) ) ) ) + x          c1: k taps, dilation d;  c2: k taps, dilation 1
// for the narrow decoder stages (C = 32, 64), where conv-by-conv execution is bound by HBM traffic (each conv
// writes its full 268 MB output; the write path tops out at ~2.7 TB/s, profiles/r01_membw_probe.txt).  Here the
// intermediate activation never leaves the CU: the x tile (all C channels, time tile + both halos) is staged into LDS
// once, already leaky-relu'd (x is recovered from it for the residual: negative samples times 1/slope), phase A
// computes c1 on N_A = WN*NR*32 columns and writes lrelu(.) (zeroed outside [0,L): c2's own zero padding) into a
// second LDS tile, phase B runs c2 on it for the N2 = N_A-(k-1) interior columns, and the
// residual is read back from the staged tile.  HBM traffic per iteration: read x once + write y once
// (was: 2 writes + 3 reads).  Same MFMA instruction, packed-weight streams and numerics as conv_mfma.hip.
#include "svoc_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace svoc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float fz_pick4(const float4& v, int s) { return s == 0 ? v.x : (s == 1 ? v.y : (s == 2 ? v.z : v.w)); }

struct FusedArgs {
  const float* x; long long x_bs; int x_ld; int L;
  const float* wp1; const float* bias1; int ksg1; int dil1; int pad1;
  const float* wp2; const float* bias2; int ksg2; int pad2;
  int ktaps; int nchunks; int C;
  float* y; long long y_bs; int y_ld; unsigned flags; float div;
  int n2; int xoff0; int xrow; int yrow;
  float slope;
  int xcd;                   // XCD-aware tile order (svoc_internal.h xcd_linear)
  long long* dbg;            // optional [nblocks][8] cycle stamps (svoc_debug_set_stamp_buffer)
};

// ---------------------------------------------------------------------------------------------------------------
// Every piece of geometry is known at compile time (C, k, c1 dilation); a generic run-time-geometry kernel (round 1) was removed in
// round 5 - other shapes run as two convolutions.  Round-1 PMC counters (profiles/r01_e_pmc_instruction_mix.txt) showed that generic kernel issuing 2.5 (C=64) / 4.0 (C=32)
// vector-ALU instructions per MFMA, and on gfx950 every vector instruction takes matrix-pipe time (fp32 MFMA runs at
// the vector rate on the same lanes).  With compile-time row strides the LDS fragment reads, the residual pull and the
// activation exchange use immediate offsets; the staging walks an exact slot count, interior tiles (all but two per
// row) skip the zero-padding clamps and selects, the weight stream is addressed as a uniform base + 32-bit lane offset,
// and the output rows as uniform row bases + one per-lane offset.
template <int C, int K, int D, int NRT = 2>
struct RbGeo {
  static constexpr int WM = C / 32;                       // waves along the rows (one 32-row tile each)
  static constexpr int WN = 4 / WM;
  static constexpr int NR = NRT;                          // 32-column tiles per wave (1: half-width tiles for short inputs)
  static constexpr int NA = WN * NR * 32;                 // c1 columns per workgroup
  static constexpr int PAD1 = (K - 1) * D / 2;
  static constexpr int PAD2 = (K - 1) / 2;
  static constexpr int N2 = (NA - (K - 1)) & ~3;          // output columns per workgroup
  static constexpr int XOFF0 = -((PAD1 + PAD2 + 3) & ~3); // x tile starts at t0 + XOFF0 (multiple of 4)
  static constexpr int LASTC = NA - 1 - PAD2 + (K - 1) * D - PAD1 - XOFF0;
  static constexpr int XROW = ((LASTC + 1 > N2 - XOFF0 ? LASTC + 1 : N2 - XOFF0) + 3) & ~3;
  static constexpr int YROW = ((NA + K - 1 + 3) & ~3) + 1;
  static constexpr int NCH = C / KC;
  static constexpr int KSG = NCH * K * (KC / 8);          // groups of 4 k-steps per 32-row tile
  static constexpr int R4 = XROW / 4;
  static constexpr int RPP = 256 / R4;                    // tile rows staged per pass (one float4 per thread)
  static constexpr int NPASS = (C + RPP - 1) / RPP;
  static constexpr int LDS_BYTES = C * (XROW > YROW ? XROW : YROW) * 4;
};

__device__ __forceinline__ float4 ld_w(const char* __restrict__ base, unsigned off) {
  return *reinterpret_cast<const float4*>(base + off);
}

// acc[nr] += W[32 rows][K] * B.  `wbase` is wave-uniform (packed weights of this wave's row tile), `voff` = lane * 16;
// `bp` = this lane's LDS pointer: tile + hi * ROW + (first column).  Group order = packing order: chunk, tap, 8-channel
// group.  Ping-pong fragment registers; the next group's requests go out after the first k-step's MFMAs.
// LDS fragment read with an immediate byte offset.  hipcc merges neighbouring fragment reads into ds_read2_b32, whose
// 8-bit dword offsets cannot span the row strides of the tiles, and then spends one v_add_u32 per pair on a new base
// register; every such vector instruction takes matrix-pipe time.  ds_read_b32 has a 16-bit byte offset: one base
// register per lane serves the whole tile.  The compiler does not track inline-asm memory operations, so the consumer
// waits explicitly (lds_wait) before the first use.
template <int OFF>
__device__ __forceinline__ float lds_rd(unsigned addr) {
  float v;
  static_assert(OFF >= 0 && OFF < 65536, "ds_read_b32 offset field");
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int NR>
__device__ __forceinline__ void lds_wait(float (&b)[4][NR]) {
  if constexpr (NR == 2) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[2][0]), "+v"(b[2][1]), "+v"(b[3][0]), "+v"(b[3][1]));
  } else {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0][0]), "+v"(b[1][0]), "+v"(b[2][0]), "+v"(b[3][0]));
  }
}
template <int NR, int ROW, int BASE>
__device__ __forceinline__ void ld_frag(float (&b)[4][NR], unsigned addr) {
  b[0][0] = lds_rd<BASE * 4>(addr);
  b[1][0] = lds_rd<(BASE + 2 * ROW) * 4>(addr);
  b[2][0] = lds_rd<(BASE + 4 * ROW) * 4>(addr);
  b[3][0] = lds_rd<(BASE + 6 * ROW) * 4>(addr);
  if constexpr (NR == 2) {
    b[0][1] = lds_rd<(BASE + 32) * 4>(addr);
    b[1][1] = lds_rd<(BASE + 2 * ROW + 32) * 4>(addr);
    b[2][1] = lds_rd<(BASE + 4 * ROW + 32) * 4>(addr);
    b[3][1] = lds_rd<(BASE + 6 * ROW + 32) * 4>(addr);
  }
}

// acc[nr] += W[32 rows][K] * B.  `wbase` is wave-uniform (packed weights of this wave's row tile), `voff` = lane * 16;
// `baddr` = this lane's LDS byte address: tile + hi * ROW + (first column).  Group order = packing order: chunk, tap,
// 8-channel group.  Ping-pong fragment registers; the next group's requests go out after the first k-step's MFMAs.
template <int NR, int ROW, int NCH, int K, int DIL>
__device__ __forceinline__ void gemm_ct(f32x16 (&acc)[NR], const char* __restrict__ wbase, unsigned voff, unsigned baddr) {
  static_assert((NCH * KC + 6) * ROW * 4 + 128 + 4 < 65536, "fragment offsets must fit the ds_read_b32 offset field");
  static_assert(NCH <= 2, "C <= 64");
  float4 a0 = ld_w(wbase, voff), a1;
  float b0[4][NR], b1[4][NR];
  ld_frag<NR, ROW, 0>(b0, baddr);
  // MFMAs of one group (4 k-steps); the next group's requests (weights: global/L2, fragments: LDS) are issued after the
  // first k-step.  (Requesting the weights two groups ahead changes nothing: profiles/r02_fused_resblock_ab.txt.)
  auto mfma_head = [&](const float4& ac, float(&bc)[4][NR]) {
    lds_wait<NR>(bc);
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) acc[nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac.x, bc[0][nr], acc[nr], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mfma_tail = [&](const float4& ac, float(&bc)[4][NR]) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 1; s < 4; ++s) {
      const float av = fz_pick4(ac, s);
#pragma unroll
      for (int nr = 0; nr < NR; ++nr) acc[nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bc[s][nr], acc[nr], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // one tap of chunk CH = 4 groups (8 channels each); NEXT: 0 = the following tap of the same chunk (column + DIL),
  // 1 = first tap of the next chunk, 2 = nothing follows
  auto tap = [&](auto chc, auto nextc, unsigned ba, unsigned va) {
    constexpr int CH = decltype(chc)::value, NEXT = decltype(nextc)::value;
    constexpr int R0 = CH * KC * ROW;
    mfma_head(a0, b0); a1 = ld_w(wbase, va + 1024); ld_frag<NR, ROW, R0 + 8 * ROW>(b1, ba); mfma_tail(a0, b0);
    mfma_head(a1, b1); a0 = ld_w(wbase, va + 2048); ld_frag<NR, ROW, R0 + 16 * ROW>(b0, ba); mfma_tail(a1, b1);
    mfma_head(a0, b0); a1 = ld_w(wbase, va + 3072); ld_frag<NR, ROW, R0 + 24 * ROW>(b1, ba); mfma_tail(a0, b0);
    mfma_head(a1, b1);
    if constexpr (NEXT == 0) { a0 = ld_w(wbase, va + 4096); ld_frag<NR, ROW, R0 + DIL>(b0, ba); }
    if constexpr (NEXT == 1) { a0 = ld_w(wbase, va + 4096); ld_frag<NR, ROW, R0 + KC * ROW - (K - 1) * DIL>(b0, ba); }
    mfma_tail(a1, b1);
  };
  unsigned va = voff;
  auto chunk = [&](auto chc) {
    constexpr int CH = decltype(chc)::value;
    unsigned ba = baddr;
#pragma unroll 1
    for (int j = 0; j < K - 1; ++j) {
      tap(chc, std::integral_constant<int, 0>{}, ba, va);
      ba += DIL * 4;
      va += 4096u;
    }
    tap(chc, std::integral_constant<int, (CH + 1 < NCH) ? 1 : 2>{}, ba, va);
    va += 4096u;
  };
  chunk(std::integral_constant<int, 0>{});
  if constexpr (NCH > 1) chunk(std::integral_constant<int, 1>{});
}

typedef float rb_f32x2 __attribute__((ext_vector_type(2)));
// leaky relu without fmaxf's canonicalising extra max: 1 packed multiply per two values + 1 max per value
__device__ __forceinline__ void rb_lrelu2(float& a, float& b, const float slope) {
  const rb_f32x2 m = (rb_f32x2){a, b} * (rb_f32x2){slope, slope};
  asm("v_max_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(m.x));
  asm("v_max_f32 %0, %1, %2" : "=v"(b) : "v"(b), "v"(m.y));
}

template <int C, int K, int D, int NRT>
__global__ void __launch_bounds__(256, 2) resblock_fused_ct_kernel(const FusedArgs p) {
  using G = RbGeo<C, K, D, NRT>;
  constexpr int WN = G::WN, NR = G::NR, XROW = G::XROW, YROW = G::YROW, R4 = G::R4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const XT = lds;                                   // lrelu(x) tile  [C][XROW]
  float* const YT = lds;                                   // lrelu(c1(.)) tile [C][YROW], aliases the x tile

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, hi = lane >> 5;
  const int tl = xcd_linear(blockIdx.x + gridDim.x * blockIdx.z, gridDim.x * gridDim.z, p.xcd);
  const int b = tl / (int)gridDim.x;
  const int t0 = (tl - b * (int)gridDim.x) * G::N2;        // first output sample of this tile
  const int L = p.L;
  const float slope = p.slope;
  long long ts[6] = {0, 0, 0, 0, 0, 0};                    // diagnostics: 100 MHz wall-clock stamps of the phases
  if (p.dbg) ts[0] = (long long)wall_clock64();
  // ---- stage the x tile: all C channels, columns [t0 + XOFF0, +XROW); zero outside [0, L); holds lrelu(x).
  // Thread (r0, g) of the first RPP * R4 threads owns float4 group g of rows r0, r0 + RPP, r0 + 2 RPP, ...: the global
  // address of pass u is a wave-uniform row base plus one per-thread offset, the LDS address one per-thread base plus
  // an immediate - no per-slot address arithmetic on the vector ALU.
  {
    constexpr int RPP = G::RPP, NPASS = G::NPASS;
    const int xs_start = t0 + G::XOFF0;
    const int r0 = tid / R4, g = tid - r0 * R4;
    const bool mine = tid < RPP * R4;
    const char* const xb = reinterpret_cast<const char*>(p.x + (long long)b * p.x_bs);
    const long long ldb = (long long)p.x_ld * 4;
    const bool interior = xs_start >= 0 && xs_start + XROW <= L;
    float4 v[NPASS];
    if (interior) {
      const int r0c = mine ? r0 : 0;                       // idle threads (tid >= RPP * R4) re-read row 0
      const unsigned toff = (unsigned)(r0c * p.x_ld + xs_start + 4 * g) * 4u;
#pragma unroll
      for (int u = 0; u < NPASS; ++u) {
        if (u * RPP + RPP - 1 < C) {                       // compile time: every row of this pass exists
          v[u] = *reinterpret_cast<const float4*>(xb + (long long)(u * RPP) * ldb + toff);
        } else {                                           // last, partial pass: rows beyond C re-read row C-1 (not written)
          const int c = min(r0c + u * RPP, C - 1);
          v[u] = *reinterpret_cast<const float4*>(xb + (long long)c * ldb + (long long)(xs_start + 4 * g) * 4);
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < NPASS; ++u) {
        const int c = min(r0 + u * RPP, C - 1);
        int t = xs_start + 4 * g;
        t = (t >= 0 && t + 3 < L) ? t : 0;
        v[u] = *reinterpret_cast<const float4*>(xb + (long long)c * ldb + (long long)t * 4);
      }
      const int t = xs_start + 4 * g;
      if (!(t >= 0 && t + 3 < L)) {      // a group that straddles an end: element-wise (rows are only 16-byte aligned as a whole)
#pragma unroll
        for (int u = 0; u < NPASS; ++u) {
          const int c = min(r0 + u * RPP, C - 1);
          const float* row = reinterpret_cast<const float*>(xb + (long long)c * ldb);
          v[u].x = (t >= 0 && t < L) ? row[t] : 0.f;
          v[u].y = (t + 1 >= 0 && t + 1 < L) ? row[t + 1] : 0.f;
          v[u].z = (t + 2 >= 0 && t + 2 < L) ? row[t + 2] : 0.f;
          v[u].w = (t + 3 >= 0 && t + 3 < L) ? row[t + 3] : 0.f;
        }
      }
    }
    float* const dst = XT + r0 * XROW + 4 * g;
#pragma unroll
    for (int u = 0; u < NPASS; ++u) {
      const bool full = u * RPP + RPP - 1 < C;            // compile time: every row of this pass exists
      if (mine && (full || r0 + u * RPP < C)) {
        float4 q = v[u];
        rb_lrelu2(q.x, q.y, slope);
        rb_lrelu2(q.z, q.w, slope);
        *reinterpret_cast<float4*>(dst + u * RPP * XROW) = q;
      }
    }
  }
  const int mt = wm;                                       // this wave's 32-row tile
  const int ncol0 = wn * NR * 32;
  const char* const w1 = reinterpret_cast<const char*>(p.wp1) + (size_t)mt * G::KSG * 1024;
  const char* const w2 = reinterpret_cast<const char*>(p.wp2) + (size_t)mt * G::KSG * 1024;
  const unsigned voff = (unsigned)lane * 16u;
  f32x16 acc[NR];
  // ---- phase A: c1 on columns m in [0, NA) <-> global time t0 - PAD2 + m
  {
    const float* bias = p.bias1 + mt * 32 + 4 * hi;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float bv = bias[(i & 3) + 8 * (i >> 2)];
#pragma unroll
      for (int nr = 0; nr < NR; ++nr) acc[nr][i] = bv;
    }
  }
  __syncthreads();
  if (p.dbg) ts[1] = (long long)wall_clock64();
  const unsigned lds0 = (unsigned)(size_t)lds;           // LDS byte address of the dynamic segment
  gemm_ct<NR, XROW, G::NCH, K, D>(acc, w1, voff, lds0 + (unsigned)(hi * XROW + (ncol0 + l31 - G::PAD2 - G::PAD1 - G::XOFF0)) * 4u);
  if (p.dbg) ts[2] = (long long)wall_clock64();
  // The x tile is now only needed for the residual: pull this wave's values into registers, then (after a barrier) the
  // same LDS region is overwritten with lrelu(c1(.)) as c2's B operand.  (Re-reading x from global memory instead - 32 loads
  // for 32 LDS reads + 64 vector instructions - is 4..8 % SLOWER: tools/rb_bench.py.)
  const float inv_slope = 1.0f / slope;
  float resv[NR][16];
#pragma unroll
  for (int nr = 0; nr < NR; ++nr) {
    const int cidx = min(ncol0 + nr * 32 + l31 - G::XOFF0, XROW - 1);
    const float* rbase = XT + (mt * 32 + 4 * hi) * XROW + cidx;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      float v0 = rbase[((r & 3) + 8 * (r >> 2)) * XROW], v1 = rbase[(((r + 1) & 3) + 8 * ((r + 1) >> 2)) * XROW];
      const rb_f32x2 m = (rb_f32x2){v0, v1} * (rb_f32x2){inv_slope, inv_slope};      // x from lrelu(x): min(v, v / slope)
      asm("v_min_f32 %0, %1, %2" : "=v"(v0) : "v"(v0), "v"(m.x));
      asm("v_min_f32 %0, %1, %2" : "=v"(v1) : "v"(v1), "v"(m.y));
      resv[nr][r] = v0; resv[nr][r + 1] = v1;
    }
  }
  __syncthreads();
  {
    float* ybase = YT + (mt * 32 + 4 * hi) * YROW + ncol0 + l31;
    const int tA0 = t0 - G::PAD2 + ncol0 + l31;            // global time of this lane's first c1 column
    if (t0 - G::PAD2 >= 0 && t0 - G::PAD2 + G::NA <= L) {  // whole c1 range inside [0, L): no zero padding to apply
#pragma unroll
      for (int nr = 0; nr < NR; ++nr)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          float v0 = acc[nr][r], v1 = acc[nr][r + 1];
          rb_lrelu2(v0, v1, slope);
          ybase[((r & 3) + 8 * (r >> 2)) * YROW + nr * 32] = v0;
          ybase[(((r + 1) & 3) + 8 * ((r + 1) >> 2)) * YROW + nr * 32] = v1;
        }
    } else {
#pragma unroll
      for (int nr = 0; nr < NR; ++nr) {
        const int tA = tA0 + nr * 32;
        const bool ok = tA >= 0 && tA < L;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[nr][r];
          ybase[((r & 3) + 8 * (r >> 2)) * YROW + nr * 32] = ok ? fmaxf(v, v * slope) : 0.f;
        }
      }
    }
  }
  // ---- phase B: c2 on the interior columns n in [0, N2) <-> global time t0 + n
  {
    const float* bias = p.bias2 + mt * 32 + 4 * hi;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float bv = bias[(i & 3) + 8 * (i >> 2)];
#pragma unroll
      for (int nr = 0; nr < NR; ++nr) acc[nr][i] = bv;
    }
  }
  __syncthreads();
  if (p.dbg) ts[3] = (long long)wall_clock64();
  if (ncol0 >= G::N2 || t0 + ncol0 >= L) return;
  gemm_ct<NR, YROW, G::NCH, K, 1>(acc, w2, voff, lds0 + (unsigned)(hi * YROW + ncol0 + l31) * 4u);
  if (p.dbg) ts[4] = (long long)wall_clock64();
  // ---- epilogue: + residual, sink flags, store.  Row bases are wave-uniform, the lane adds one offset.
  const long long row0 = (long long)b * p.y_bs + (long long)(mt * 32) * p.y_ld;
#pragma unroll
  for (int nr = 0; nr < NR; ++nr) {
    const int n = ncol0 + nr * 32 + l31;
    const int t = t0 + n;
    if (n >= G::N2 || t >= L) continue;
    const unsigned lo = (unsigned)(4 * hi * p.y_ld + t) * 4u;
    char* const yb = reinterpret_cast<char*>(p.y + row0);
    float vo[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) vo[r] = acc[nr][r] + resv[nr][r];
    if (p.flags & F_ACC) {
      float yo[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) yo[r] = *reinterpret_cast<const float*>(yb + (size_t)((r & 3) + 8 * (r >> 2)) * p.y_ld * 4 + lo);
#pragma unroll
      for (int r = 0; r < 16; ++r) vo[r] = yo[r] + vo[r];
    }
    if (p.flags & F_DIV) {          // one uniform branch (an in-loop `if` becomes 16 unconditional divisions)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int r = 0; r < 16; ++r) vo[r] = vo[r] / p.div;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) *reinterpret_cast<float*>(yb + (size_t)((r & 3) + 8 * (r >> 2)) * p.y_ld * 4 + lo) = vo[r];
  }
  if (p.dbg && threadIdx.x == 0) {
    ts[5] = (long long)wall_clock64();
    long long* d = p.dbg + 8 * (blockIdx.x + (long long)gridDim.x * blockIdx.z);
#pragma unroll
    for (int i = 0; i < 6; ++i) d[i] = ts[i];
    d[6] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_ID: wave, simd, cu, sh, se
    d[7] = (long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);      // XCC_ID
  }
}

// (Half-width tiles, NRT = 1, for launches with fewer workgroups than CUs were measured neutral at 1 x 200 frames.)
template <int C, int K, int D>
static int launch_ct(const FusedArgs& a, int B, int L, hipStream_t st) {
  using G = RbGeo<C, K, D, 2>;
  static_assert(G::N2 > 0 && G::LDS_BYTES <= 160 * 1024, "tile does not fit");
  auto kern = resblock_fused_ct_kernel<C, K, D, 2>;
  SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
  const int ntn = (L + G::N2 - 1) / G::N2;
  size_t lds = (size_t)G::LDS_BYTES;
  if (const char* e = getenv("SVOC_RB_LDS_MIN")) lds = std::max(lds, (size_t)atoi(e));      // diagnostics (tools/rb_timeline.py): cap the occupancy
  hipLaunchKernelGGL(kern, dim3(ntn, 1, B), dim3(256), lds, st, a);
  return SVOC_OK;
}

// Dispatch on the compile-time geometries of the model (C = 32 / 64, k = 3 / 7 / 11, c1 dilation 1 / 3 / 5).  Returns
// false (nothing launched) for any other shape: the generic kernel above handles it.
static bool launch_v2(const FusedArgs& a, const PackedConv& c1, const PackedConv& c2, int B, int L, hipStream_t st, int* rc) {
  const int C = c1.Cin, k = c1.ktaps, d = c1.dil;
  if (!resblock_fused_ct_supported(C, k, d)) return false;
  if ((long long)4 * 32 * a.y_ld + L >= (1LL << 30)) return false;     // 32-bit lane offsets
#define SVOC_RB(CC, KK, DD) if (C == CC && k == KK && d == DD) { *rc = launch_ct<CC, KK, DD>(a, B, L, st); return true; }
  SVOC_RB(32, 3, 1) SVOC_RB(32, 3, 3) SVOC_RB(32, 3, 5) SVOC_RB(32, 7, 1) SVOC_RB(32, 7, 3) SVOC_RB(32, 7, 5)
  SVOC_RB(32, 11, 1) SVOC_RB(32, 11, 3) SVOC_RB(32, 11, 5)
  SVOC_RB(64, 3, 1) SVOC_RB(64, 3, 3) SVOC_RB(64, 3, 5) SVOC_RB(64, 7, 1) SVOC_RB(64, 7, 3) SVOC_RB(64, 7, 5)
  SVOC_RB(64, 11, 1) SVOC_RB(64, 11, 3) SVOC_RB(64, 11, 5)
#undef SVOC_RB
  return false;
}

// Eligibility + launch.  Returns 1 if the fused kernel does not apply (caller runs the two convolutions).
bool resblock_fused_ct_supported(int C, int k, int dil) {
  return (C == 32 || C == 64) && (k == 3 || k == 7 || k == 11) && (dil == 1 || dil == 3 || dil == 5);
}

int launch_resblock_fused(const PackedConv& c1, const PackedConv& c2, const float* x, long long x_bs, int x_ld, float* y,
                          long long y_bs, int y_ld, unsigned flags, float div, int B, int L, hipStream_t st) {
  static const bool enabled = !(getenv("SVOC_FUSE") && atoi(getenv("SVOC_FUSE")) == 0);
  if (!enabled) return 1;
  const int C = c1.Cin;
  if (!(C == 32 || C == 64) || c1.Cout != C || c2.Cin != C || c2.Cout != C) return 1;
  if (c1.ktaps != c2.ktaps || c2.dil != 1 || c1.transposed || c2.transposed || c1.paired || c2.paired) return 1;
  if ((x_ld & 3) || (x_bs & 3) || (reinterpret_cast<uintptr_t>(x) & 15)) return 1;
  if (flags & ~(unsigned)(F_ACC | F_DIV)) return 1;
  const int k = c1.ktaps;
  // other kernel sizes / dilations run as two convolutions (a generic run-time-geometry copy of this kernel existed until round 5)
  if (!resblock_fused_ct_supported(C, k, c1.dil)) return 1;
  if ((long long)4 * 32 * y_ld + L >= (1LL << 30)) return 1;           // the kernel's 32-bit lane offsets
  const int NA = (C == 32) ? 256 : 128;
  FusedArgs a;
  a.x = x; a.x_bs = x_bs; a.x_ld = x_ld; a.L = L;
  a.wp1 = c1.wp.f(); a.bias1 = c1.bias.f(); a.ksg1 = c1.ksg_total; a.dil1 = c1.dil; a.pad1 = c1.pad;
  a.wp2 = c2.wp.f(); a.bias2 = c2.bias.f(); a.ksg2 = c2.ksg_total; a.pad2 = c2.pad;
  a.ktaps = k; a.nchunks = C / KC; a.C = C;
  a.y = y; a.y_bs = y_bs; a.y_ld = y_ld; a.flags = flags; a.div = div;
  a.n2 = (NA - (k - 1)) & ~3;
  if (a.n2 <= 0) return 1;
  a.xoff0 = (-(c2.pad + c1.pad)) & ~3;                                  // floor to a multiple of 4
  const int last_col = NA - 1 - c2.pad + (k - 1) * c1.dil - c1.pad - a.xoff0;   // largest x-tile column read by phase A
  a.xrow = round_up(std::max(last_col + 1, a.n2 - a.xoff0), 4);
  a.yrow = round_up(NA + k - 1, 4) + 1;                                // odd stride: both half-waves hit distinct banks
  a.slope = 0.1f;
  a.xcd = xcd_mapping_enabled();
  a.dbg = debug_stamp_buffer();
  const size_t lds = (size_t)C * std::max(a.xrow, a.yrow) * sizeof(float);
  if (lds > 160 * 1024) return 1;
  const int ntn = (L + a.n2 - 1) / a.n2;
  dim3 grid(ntn, 1, B);
  stats_add_conv((c1.flops_per_col + c2.flops_per_col) * (double)B * (double)L, 2);
  int prof_idx = -1;
  if (prof_enabled()) {
    char d[160];
    snprintf(d, sizeof(d), "fusedRB C%-4d k%-2d d%-2d N%-7d B%-3d NA%d", C, k, c1.dil, L, B, NA);
    prof_idx = prof_begin(st, d, (c1.flops_per_col + c2.flops_per_col) * (double)B * (double)L);
  }
  int rc2 = SVOC_OK;
  const bool launched = launch_v2(a, c1, c2, B, L, st, &rc2);         // resblock_fused_ct_supported() held above
  if (!launched) rc2 = SVOC_ERR_UNSUPPORTED;
  if (rc2 != SVOC_OK) { prof_end(st, prof_idx); return rc2; }
  prof_end(st, prof_idx);
  SVOC_HIP(hipGetLastError());
  return SVOC_OK;
}

}  // namespace svoc
