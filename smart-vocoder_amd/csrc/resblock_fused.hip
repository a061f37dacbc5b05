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

namespace svoc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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

__device__ __forceinline__ float fz_pick4(const float4& v, int s) { return s == 0 ? v.x : (s == 1 ? v.y : (s == 2 ? v.z : v.w)); }

// acc[nr] += W[rows of m-tile mt][K] * B, B fragments read from an LDS tile `tile` ([channels][row_len]) at column
// offset col0 + tap*dil; ACT applies leaky-relu to the fragments as they are read.
template <int NR, bool ACT>
__device__ __forceinline__ void fused_gemm(f32x16 (&acc)[NR], const float4* __restrict__ wp4, long long abase, int ksg_total,
                                           const float* tile, int row_len, int col0, int ktaps, int dil, int nchunks,
                                           int hi, float slope) {
  // ping-pong fragment registers; the next group's requests are issued after the first k-step's MFMAs
  float4 a0 = wp4[abase], a1;
  int ksg = 0;
  for (int ch = 0; ch < nchunks; ++ch) {
    const float* bp = tile + (ch * KC + hi) * row_len + col0;
    float b0[4][NR], b1[4][NR];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int nr = 0; nr < NR; ++nr) {
        const float v = bp[(2 * s) * row_len + nr * 32];
        b0[s][nr] = ACT ? fmaxf(v, v * slope) : v;            // lrelu for slope in (0,1)
      }
    const int ngroups = ktaps * (KC / 8);
    int g = 0;
    auto run_group = [&](float4& ac, float(&bc)[4][NR], float4& an, float(&bn)[4][NR], bool last_group) {
#pragma unroll
      for (int nr = 0; nr < NR; ++nr) acc[nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac.x, bc[0][nr], acc[nr], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      ++ksg;
      an = wp4[abase + (long long)(ksg < ksg_total ? ksg : 0) * 64];
      const float* bpn = (g == KC / 8 - 1) ? bp + dil - (KC - 8) * row_len : bp + 8 * row_len;
      if (!last_group) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int nr = 0; nr < NR; ++nr) bn[s][nr] = bpn[(2 * s) * row_len + nr * 32];
      }
      bp = bpn;
      g = (g + 1) & (KC / 8 - 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 1; s < 4; ++s) {
        const float av = fz_pick4(ac, s);
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) acc[nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bc[s][nr], acc[nr], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (ACT && !last_group) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int nr = 0; nr < NR; ++nr) bn[s][nr] = fmaxf(bn[s][nr], bn[s][nr] * slope);
      }
    };
    for (int gi = 0; gi < ngroups; gi += 2) {
      run_group(a0, b0, a1, b1, false);
      run_group(a1, b1, a0, b0, gi + 2 >= ngroups);
    }
  }
}

template <int WM, int WN, int NR>
__global__ void __launch_bounds__(256, 2) resblock_fused_kernel(const FusedArgs p) {
  constexpr int SU = 13;                                   // one batch of staging loads per thread
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const XT = lds;                                   // lrelu(x) tile  [C][xrow]
  float* const YT = lds;                                   // lrelu(c1(.)) tile [C][yrow], ALIASES the x tile (see below)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, hi = lane >> 5;
  const int tl = xcd_linear(blockIdx.x + gridDim.x * blockIdx.z, gridDim.x * gridDim.z, p.xcd);
  const int b = tl / (int)gridDim.x;
  const int t0 = (tl - b * (int)gridDim.x) * p.n2;         // first output sample of this tile
  const int h2 = p.pad2;
  long long ts[6] = {0, 0, 0, 0, 0, 0};
  if (p.dbg) ts[0] = __builtin_readcyclecounter();

  // ---- stage the x tile: all C channels, columns [t0 + xoff0, +xrow); zero outside [0, L)
  {
    const int R4 = p.xrow >> 2;
    const int total = p.C * R4;
    const int xs_start = t0 + p.xoff0;
    const float* xb = p.x + (long long)b * p.x_bs;
    int wc = tid / R4, wg = tid - wc * R4;
    const int dc = 256 / R4, dg = 256 - dc * R4;
    for (int base = tid; base < total; base += 256 * SU) {
      float4 v[SU];
      const int wc_s = wc, wg_s = wg;
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int c = min(wc, p.C - 1);
        int t = xs_start + 4 * wg;
        t = (t >= 0 && t < p.L) ? t : 0;
        v[u] = *reinterpret_cast<const float4*>(xb + (long long)c * p.x_ld + t);
        wc += dc; wg += dg;
        if (wg >= R4) { wg -= R4; ++wc; }
      }
      int wc2 = wc_s, wg2 = wg_s;
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        if (base + u * 256 < total) {
          const int t = xs_start + 4 * wg2;
          float4 q = v[u];
          q.x = (t >= 0 && t < p.L) ? q.x : 0.f;
          q.y = (t + 1 >= 0 && t + 1 < p.L) ? q.y : 0.f;
          q.z = (t + 2 >= 0 && t + 2 < p.L) ? q.z : 0.f;
          q.w = (t + 3 >= 0 && t + 3 < p.L) ? q.w : 0.f;
          // the tile holds lrelu(x): c1 re-reads every element once per tap, activating at staging costs k times less
          // vector ALU work beside the MFMAs; the residual recovers x from it (see below)
          q.x = fmaxf(q.x, q.x * p.slope);
          q.y = fmaxf(q.y, q.y * p.slope);
          q.z = fmaxf(q.z, q.z * p.slope);
          q.w = fmaxf(q.w, q.w * p.slope);
          *reinterpret_cast<float4*>(XT + wc2 * p.xrow + 4 * wg2) = q;
        }
        wc2 += dc; wg2 += dg;
        if (wg2 >= R4) { wg2 -= R4; ++wc2; }
      }
    }
  }
  const int mt = wm;                                       // this wave's 32-row tile (WM tiles cover C)
  f32x16 acc[NR];
  // ---- phase A: c1 on columns m in [0, NA) <-> global time t0 - h2 + m
#pragma unroll
  for (int nr = 0; nr < NR; ++nr)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nr][i] = p.bias1[mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi];
  __syncthreads();
  if (p.dbg) ts[1] = __builtin_readcyclecounter();
  {
    const int col0 = wn * NR * 32 + l31 - h2 - p.pad1 - p.xoff0;
    fused_gemm<NR, false>(acc, reinterpret_cast<const float4*>(p.wp1), (long long)mt * p.ksg1 * 64 + lane, p.ksg1, XT, p.xrow, col0,
                         p.ktaps, p.dil1, p.nchunks, hi, p.slope);
  }
  // The x tile is now only needed for the residual: every wave pulls its own residual values into registers, then
  // (after a barrier) the same LDS region is overwritten with lrelu(c1(.)) as c2's B operand.  Halving the LDS
  // footprint doubles the number of resident workgroups.
  if (p.dbg) ts[2] = __builtin_readcyclecounter();
  const int ncol0 = wn * NR * 32;
  const float inv_slope = 1.0f / p.slope;
  float resv[NR][16];
#pragma unroll
  for (int nr = 0; nr < NR; ++nr) {
    const int n = ncol0 + nr * 32 + l31;
    const int cidx = min(n - p.xoff0, p.xrow - 1);
    const float* rbase = XT + (mt * 32 + 4 * hi) * p.xrow + cidx;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // x from lrelu(x): negative values were scaled by the slope (0.1: 1/slope = 10 exactly in fp32; the round trip
      // x*0.1*10 differs from x by at most one ulp, 6e-8 relative, on negative samples only)
      const float v = rbase[((r & 3) + 8 * (r >> 2)) * p.xrow];
      resv[nr][r] = fminf(v, v * inv_slope);      // inv_slope > 1: min picks v*inv_slope for v < 0, v otherwise
    }
  }
  __syncthreads();
  // lrelu, zero outside [0, L) (c2 zero-pads ITS input)
#pragma unroll
  for (int nr = 0; nr < NR; ++nr) {
    const int m = wn * NR * 32 + nr * 32 + l31;
    const int tA = t0 - h2 + m;
    const bool ok = tA >= 0 && tA < p.L;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      float v = acc[nr][r];
      v = fmaxf(v, v * p.slope);
      YT[row * p.yrow + m] = ok ? v : 0.f;
    }
  }
  // ---- phase B: c2 on the interior columns n in [0, n2) <-> global time t0 + n
#pragma unroll
  for (int nr = 0; nr < NR; ++nr)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nr][i] = p.bias2[mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi];
  __syncthreads();
  if (p.dbg) ts[3] = __builtin_readcyclecounter();
  if (ncol0 >= p.n2 || t0 + ncol0 >= p.L) return;
  fused_gemm<NR, false>(acc, reinterpret_cast<const float4*>(p.wp2), (long long)mt * p.ksg2 * 64 + lane, p.ksg2, YT, p.yrow,
                        ncol0 + l31, p.ktaps, 1, p.nchunks, hi, 1.0f);
  if (p.dbg) ts[4] = __builtin_readcyclecounter();
  // ---- epilogue: + residual (x recovered from the LDS tile above), sink flags, store
#pragma unroll
  for (int nr = 0; nr < NR; ++nr) {
    const int n = ncol0 + nr * 32 + l31;
    const int t = t0 + n;
    if (n >= p.n2 || t >= p.L) continue;
    float* ybase = p.y + (long long)b * p.y_bs + (long long)(mt * 32 + 4 * hi) * p.y_ld + t;
    float yo[16];
    if (p.flags & F_ACC) {
#pragma unroll
      for (int r = 0; r < 16; ++r) yo[r] = ybase[(long long)((r & 3) + 8 * (r >> 2)) * p.y_ld];
    }
    float vo[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[nr][r] + resv[nr][r];
      if (p.flags & F_ACC) v = yo[r] + v;
      vo[r] = v;
    }
    if (p.flags & F_DIV) {          // one uniform branch: see conv_mfma.hip (an in-loop `if` becomes 16 unconditional divisions)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int r = 0; r < 16; ++r) vo[r] = vo[r] / p.div;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) ybase[(long long)((r & 3) + 8 * (r >> 2)) * p.y_ld] = vo[r];
  }
  if (p.dbg && threadIdx.x == 0) {
    ts[5] = __builtin_readcyclecounter();
    long long* d = p.dbg + 8 * (blockIdx.x + (long long)gridDim.x * blockIdx.z);
#pragma unroll
    for (int i = 0; i < 6; ++i) d[i] = ts[i];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent variant with loader waves.  Phase stamps of the kernel above (tools/fused_phases.py) show the staging
// phase taking 19-40 % of a workgroup's life although it moves only 35-48 KB: co-resident workgroups run in lock
// step, so every CU requests its tiles at the same moment and the chip alternates between an HBM burst and an MFMA
// phase with idle HBM.  Here a workgroup has 4 MFMA waves (same code as above) plus 4 loader waves and walks a list
// of tiles: the loaders request tile i+1 right after publishing tile i and hold it in registers (12-13 float4 per
// lane) while the MFMA waves run both GEMMs of tile i, then write it (zero padding + leaky-relu) into the LDS tile
// as soon as c2 has finished reading it.  Requests are spread over the whole compute time and never sit in the MFMA
// waves' in-order vmcnt queue.  Barriers are raw s_barrier (a fence would drain the loaders' requests and the MFMA
// waves' output stores).  MEASURED: correct on the parity suite, but per-CU throughput equals the kernel above
// (C=64: k=3 0.60 vs 0.59 ms, k=11 1.80 vs 1.75 ms): with two workgroups per CU only two MFMA waves share a SIMD and
// the exchange / epilogue phases are no longer covered by a third and fourth workgroup.  Opt-in (SVOC_FUSE_WS=1).
__device__ __forceinline__ void fz_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int WM, int WN, int NR, int XREG>
__global__ void __launch_bounds__(512, 4) resblock_fused_ws_kernel(const FusedArgs p, const int ntn, const int total_tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const XT = lds;
  float* const YT = lds;                                   // aliases the x tile
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h2 = p.pad2;
  const int ntl = (total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // tiles of this workgroup (>= 1)

  if (wave >= 4) {
    // ===================================================================== loader waves
    const int pl = tid - 256;                              // 0..255
    const int R4 = p.xrow >> 2;
    const int total = p.C * R4;
    const int wc0 = pl / R4, wg0 = pl - wc0 * R4;
    const int dc = 256 / R4, dg = 256 - dc * R4;
    float4 v[XREG];
    auto issue = [&](int tile, float4(&vv)[XREG]) {
      const int b = tile / ntn;
      const int xs_start = (tile - b * ntn) * p.n2 + p.xoff0;
      const float* xb = p.x + (long long)b * p.x_bs;
      int wc = wc0, wg = wg0;
#pragma unroll
      for (int u = 0; u < XREG; ++u) {
        const int c = min(wc, p.C - 1);
        int t = xs_start + 4 * wg;
        t = (t >= 0 && t < p.L) ? t : 0;
        vv[u] = *reinterpret_cast<const float4*>(xb + (long long)c * p.x_ld + t);
        wc += dc; wg += dg;
        if (wg >= R4) { wg -= R4; ++wc; }
      }
    };
    auto publish = [&](int tile, const float4(&vv)[XREG]) {
      const int b = tile / ntn;
      const int xs_start = (tile - b * ntn) * p.n2 + p.xoff0;
      int wc = wc0, wg = wg0;
#pragma unroll
      for (int u = 0; u < XREG; ++u) {
        if (pl + u * 256 < total) {
          const int t = xs_start + 4 * wg;
          float4 q = vv[u];
          q.x = (t >= 0 && t < p.L) ? q.x : 0.f;
          q.y = (t + 1 >= 0 && t + 1 < p.L) ? q.y : 0.f;
          q.z = (t + 2 >= 0 && t + 2 < p.L) ? q.z : 0.f;
          q.w = (t + 3 >= 0 && t + 3 < p.L) ? q.w : 0.f;
          q.x = fmaxf(q.x, q.x * p.slope);
          q.y = fmaxf(q.y, q.y * p.slope);
          q.z = fmaxf(q.z, q.z * p.slope);
          q.w = fmaxf(q.w, q.w * p.slope);
          *reinterpret_cast<float4*>(XT + wc * p.xrow + 4 * wg) = q;
        }
        wc += dc; wg += dg;
        if (wg >= R4) { wg -= R4; ++wc; }
      }
    };
    int tile = blockIdx.x;
    issue(tile, v);
    for (int j = 0; j < ntl; ++j) {
      publish(tile, v);                                    // waits for the requests of this tile
      fz_barrier();                                        // S1: x tile ready
      tile += gridDim.x;
      if (j + 1 < ntl) issue(tile, v);                     // in flight until the next publish
      fz_barrier();                                        // S2: c1 done, residual pulled
      fz_barrier();                                        // S3: activation tile written
      fz_barrier();                                        // S4: c2 done -> the LDS tile may be overwritten
    }
    return;
  }

  // ======================================================================= MFMA waves
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, hi = lane >> 5;
  const int mt_ = wm;
  const int ncol0_ = wn * NR * 32;
  const float inv_slope = 1.0f / p.slope;
  f32x16 acc[NR];
  int tile = blockIdx.x;
  for (int j = 0; j < ntl; ++j, tile += gridDim.x) {
    const int b = tile / ntn;
    const int t0 = (tile - b * ntn) * p.n2;
    int opq = 0;
    asm volatile("" : "+s"(opq));      // opaque zero: keeps tile-invariant loads (bias) and the ~80 tile-invariant LDS /
    const int mt = mt_ + opq;          // global address registers of the exchange and the epilogue inside the loop
    const int ncol0 = ncol0_ + opq;    // (hoisted, they push the kernel past the 128 registers two workgroups need)
    // ---- phase A: c1 on columns m in [0, NA) <-> global time t0 - h2 + m
#pragma unroll
    for (int nr = 0; nr < NR; ++nr)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nr][i] = p.bias1[mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi];
    fz_barrier();                                          // S1
    {
      const int col0 = ncol0 + l31 - h2 - p.pad1 - p.xoff0;
      fused_gemm<NR, false>(acc, reinterpret_cast<const float4*>(p.wp1), (long long)mt * p.ksg1 * 64 + lane, p.ksg1, XT, p.xrow, col0,
                            p.ktaps, p.dil1, p.nchunks, hi, p.slope);
    }
    float resv[NR][16];
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
      const int n = ncol0 + nr * 32 + l31;
      const int cidx = min(n - p.xoff0, p.xrow - 1);
      const float* rbase = XT + (mt * 32 + 4 * hi) * p.xrow + cidx;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = rbase[((r & 3) + 8 * (r >> 2)) * p.xrow];
        resv[nr][r] = fminf(v, v * inv_slope);      // inv_slope > 1: min picks v*inv_slope for v < 0, v otherwise
      }
    }
    fz_barrier();                                          // S2
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
      const int m = ncol0 + nr * 32 + l31;
      const int tA = t0 - h2 + m;
      const bool ok = tA >= 0 && tA < p.L;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        float v = acc[nr][r];
        v = fmaxf(v, v * p.slope);
        YT[row * p.yrow + m] = ok ? v : 0.f;
      }
    }
#pragma unroll
    for (int nr = 0; nr < NR; ++nr)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nr][i] = p.bias2[mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi];
    fz_barrier();                                          // S3
    const bool live = !(ncol0 >= p.n2 || t0 + ncol0 >= p.L);
    if (live)
      fused_gemm<NR, false>(acc, reinterpret_cast<const float4*>(p.wp2), (long long)mt * p.ksg2 * 64 + lane, p.ksg2, YT, p.yrow,
                            ncol0 + l31, p.ktaps, 1, p.nchunks, hi, 1.0f);
    fz_barrier();                                          // S4: the loaders may overwrite the tile while the stores go out
    if (!live) continue;
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
      const int n = ncol0 + nr * 32 + l31;
      const int t = t0 + n;
      if (n >= p.n2 || t >= p.L) continue;
      float* ybase = p.y + (long long)b * p.y_bs + (long long)(mt * 32 + 4 * hi) * p.y_ld + t;
      float yo[16];
      if (p.flags & F_ACC) {
#pragma unroll
        for (int r = 0; r < 16; ++r) yo[r] = ybase[(long long)((r & 3) + 8 * (r >> 2)) * p.y_ld];
      }
      float vo[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[nr][r] + resv[nr][r];
        if (p.flags & F_ACC) v = yo[r] + v;
        vo[r] = v;
      }
      if (p.flags & F_DIV) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r) vo[r] = vo[r] / p.div;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) ybase[(long long)((r & 3) + 8 * (r >> 2)) * p.y_ld] = vo[r];
    }
  }
}

// Eligibility + launch.  Returns 1 if the fused kernel does not apply (caller runs the two convolutions).
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
  const int ncu = device_cu_count();
  static const bool ws_on = getenv("SVOC_FUSE_WS") && atoi(getenv("SVOC_FUSE_WS")) != 0;   // opt-in: measured at parity
  const long long total_tiles = (long long)ntn * B;
  const int xslots = (C * (a.xrow / 4) + 255) / 256;        // float4 per loader lane
  // persistent loader-wave variant: two 8-wave workgroups per CU, needs enough tiles to pipeline and the tile in 10/12 float4 per loader lane
  const bool use_ws = ws_on && total_tiles >= 8LL * ncu && total_tiles < 0x7fffffffLL && xslots <= (C == 32 ? 10 : 12) && 2 * lds <= 160 * 1024;
  if (use_ws) {
    const int gx = (int)std::min<long long>(total_tiles, 2LL * ncu);
    a.dbg = nullptr;
    if (C == 32) {
      auto kern = resblock_fused_ws_kernel<1, 4, 2, 10>;
      SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
      hipLaunchKernelGGL(kern, dim3(gx), dim3(512), lds, st, a, ntn, (int)total_tiles);
    } else {
      auto kern = resblock_fused_ws_kernel<2, 2, 2, 12>;
      SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
      hipLaunchKernelGGL(kern, dim3(gx), dim3(512), lds, st, a, ntn, (int)total_tiles);
    }
  } else if (C == 32) {
    auto kern = resblock_fused_kernel<1, 4, 2>;
    SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
  } else {
    auto kern = resblock_fused_kernel<2, 2, 2>;
    SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
  }
  prof_end(st, prof_idx);
  SVOC_HIP(hipGetLastError());
  return SVOC_OK;
}

}  // namespace svoc
