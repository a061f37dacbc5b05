// A whole WN stack (reference modules.py:148-176: n_layers x [in_layer k = 5 -> gate -> res_skip 1 x 1 -> residual / skip]) in ONE persistent
// launch (round 5): where wn_fused.hip's wn_layer_f25_kernel is launched once per layer - 48 dependent launches per `infer`,
// each of which stages its x tile from global memory, writes x and read-modify-writes the skip sum - here a workgroup keeps its 32-column tile
// of x in LDS across the layers of the stack and only the TWO-COLUMN EDGES of the tile travel: after layer i every workgroup stores the first and
// last two columns of its new x to a halo buffer (device-coherent stores), raises a per-tile counter (release), and before layer i + 1 waits for
// the counters of its two neighbours (acquire) and fetches their edges.  Neighbour-to-neighbour only - no grid-wide barrier.  Every workgroup of
// the launch must be resident at once (they wait for each other): the launcher takes the stack only while there are no more tiles than CUs (one
// 768-thread, 154 KB workgroup per CU).  When the GPU is shared, part of the launch is resident at first: workgroups are dispatched in tile order, the
// lowest resident tile's left neighbour has finished, and a window of more than n_layers resident tiles always lets its low end run to completion and
// free its CUs - progress needs n_layers + 1 CUs, not all of them.  The waits are bounded all the same (SVOC_PERSIST_TIMEOUT_MS of wall time, default
// 2 s; the launcher also asks the occupancy calculator whether the grid CAN be resident): a workgroup that gives up (a) poisons the call it is part
// of - its mask factor becomes NaN, so its x tile, the edges its neighbours fetch from it and its rows of the stack's output are NaN from that layer
// on, and the reference's WN.forward cannot return a wrong finite tensor either (modules.py:148-176) -, (b) stops waiting for the rest of the launch,
// (c) raises a host-visible error word (async_error_word(): the next call fails with the reason and the process takes the per-layer launches from
// then on, include/svoc.h svoc_check_async_error) - never a hung GPU, never a finite wrong result.
// Memory ordering (gfx950, one L2 per XCD): edges and counters are device-scope relaxed atomics (sc1: they bypass the non-coherent caches in both
// directions); the waves that store edges wait for the stores' acknowledgements with an explicit s_waitcnt vmcnt(0) (a workgroup-scope release fence
// alone compiles to s_waitcnt lgkmcnt(0) on gfx950: ADVICE r5) ahead of the workgroup
// barrier behind which thread 0 raises the counter; the reader's edge loads are issued behind the barrier that follows the successful poll.  No
// L2 write-back or invalidate is involved, so the weight images stay cached.  Every (layer, tile, edge) has a halo slot of its OWN, written once per
// launch.  The hand-shake counters are cleared by the last workgroup to LEAVE a launch, not by a memset ahead of it (see launch_wn_stack_f25: inside a
// replayed plan on a shared GPU the launch found them uncleared - nobody waited, and with two slots reused by layer parity the edges were visibly wrong).
// Per layer it is wn_layer_f25_kernel's arithmetic, instruction for instruction (F(2,5) in_layer on v_mfma_f32_16x16x4_f32 with the K halves on
// two waves, gate from exp2 / rcp, res_skip as the VALU-free 32x32x2 stream): results are bit-identical to the per-layer launches.
// H = 192, k = 5, dilation 1, no conditioning input (the path's five WN stacks: models.py:35-47, modules.py:324-343).
#include "svoc_internal.h"
#include "wino_common.h"
#include "wn_f25.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace svoc {

constexpr int WNS_MAXL = 16;
struct WnStackLayer {
  const float* wpf; const float* bias1;                 // in_layer: F(2,5) image (pack_wn_f25), bias in paired tile order
  const float* wp2; const float* bias2;                 // res_skip: split packing (last layer: plain)
};
struct WnStackArgs {
  const float* x; long long x_bs; int x_ld;             // stack input [B][H][x_ld] (masked by the caller)
  float* out; long long out_bs; int out_ld;             // skip sum -> the stack's output
  const float* mask; long long mask_bs;
  const WnStackLayer* layers;                           // [NL], in device memory (a by-value array indexed by the layer would be copied to scratch)
  int ksg2; int NL; int T;
  float* halo;                                          // [WNS_MAXL (layer)][tiles][2 (left edge, right edge)][H][2]: a slot is written ONCE per launch
  int* exited;                                          // [17] exit counters (two levels) } zero when a launch starts: the LAST workgroup to leave a launch
  int* done;                                            // [tiles]: layers completed   } clears both for the next one (no memset node: see launch_wn_stack_f25)
  int* err;                                             // raised when a bounded wait gave up
  long long* dbg;
  unsigned long long timeout;                           // bound of a wait in ticks of the 100 MHz wall counter (persist_timeout_ticks())
  int fault_tile;                                       // diagnostics: this tile's workgroup never raises its layer counter (-1: none)
};

typedef unsigned int wnk_u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(768) wn_stack_f25_kernel(const WnStackArgs p) {
  constexpr int H = WNF_H, NPAIRS = WNF_NP;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const XT = lds;                                    // x tile [H][40]: columns t0 - 4 .. t0 + 35, resident across the layers
  float* const PLN = lds + H * WNF_XROW;                    // V_p [6][H][16]; the acts tile [H][33] aliases it from the gate on
  float* const AT = PLN;
  float* const RED = PLN + 6 * WNF_PLANE;                   // exchange area [12 waves][16][64]
  volatile int* const BAD = reinterpret_cast<volatile int*>(lds + WNF_LDS_FLOATS);      // set by a wait that gave up: the workgroup's results are NaN from then on
  // the layer's two bias vectors (in_layer: 2 H in paired tile order; res_skip: 2 H, last layer H): requested from global memory at the top of the layer,
  // written here behind the input transform, read by the gate and by phase B's accumulator set-up - as global loads in place they were two exposed L2 round
  // trips between the two MFMA phases (round 6: tools/wn_stack_timeline.py, "gate" 1.78 us of a 42 us layer)
  float* const BIASL = lds + WNF_LDS_FLOATS + 4;
  if (threadIdx.x == 0) *BAD = 0;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = wave >= NPAIRS ? 1 : 0;                    // K half
  const int pi = wave - kh * NPAIRS;                        // row pair
  const int l31 = lane & 31;
  // Block -> tile through xcd_linear: the hardware deals workgroups to the eight XCDs round robin (block % 8), xcd_linear gives XCD c a CONTIGUOUS run of
  // tiles, and an XCD starts its blocks in order - so the tiles an XCD holds at any time are consecutive, neighbours share an XCD (and its L2) except at
  // seven seams (none at 16 x 512, where a run is two whole utterances), and on a shared GPU a run of more than n_layers resident tiles always finishes its
  // low end and frees CUs.  (Plain block order put neighbours on different XCDs: two processes on one GPU then held each other's missing neighbours, 30 s
  // stalls.  Tickets drawn from one device-wide counter were order-proof but cost 42 us per launch: 256 serialised device-scope atomics.)
  const int ntx = gridDim.x, ntiles = ntx * gridDim.z;
  const int gt = xcd_linear((int)(blockIdx.x + gridDim.x * blockIdx.z), ntiles, 1);
  const int b = gt / ntx, tile = gt - b * ntx;
  const int t0 = tile * 32;
  const int NL = p.NL;
  const bool has_left = tile > 0, has_right = tile + 1 < ntx;

  // ---- stage the x tile: all H channels, columns [t0 - 4, t0 + 36), zero outside [0, T)
  {
    constexpr int R4 = WNF_XROW / 4, total = H * R4, SU = 3;
    const int xs_start = t0 - 4;
    const float* xb = p.x + (long long)b * p.x_bs;
    const bool vec = ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0) && (p.x_ld & 3) == 0 && (p.x_bs & 3) == 0;
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int it = tid + 768 * u;
      if (it < total) {
        const int c = it / R4, g4 = it - c * R4;
        const int t = xs_start + 4 * g4;
        const float* row = xb + (long long)c * p.x_ld;
        float4 q;
        if (vec && t >= 0 && t + 3 < p.T) q = *reinterpret_cast<const float4*>(row + t);
        else {
          q.x = (t >= 0 && t < p.T) ? row[t] : 0.f;
          q.y = (t + 1 >= 0 && t + 1 < p.T) ? row[t + 1] : 0.f;
          q.z = (t + 2 >= 0 && t + 2 < p.T) ? row[t + 2] : 0.f;
          q.w = (t + 3 >= 0 && t + 3 < p.T) ? row[t + 3] : 0.f;
        }
        *reinterpret_cast<float4*>(XT + c * WNF_XROW + 4 * g4) = q;
      }
    }
  }
  const bool col_ok = t0 + l31 < p.T;                       // this lane's column in phase B / the epilogue: t0 + l31
  float mk = col_ok ? p.mask[(long long)b * p.mask_bs + t0 + l31] : 0.f;
  __syncthreads();

  for (int li = 0; li < NL; ++li) {
    const bool last = li == NL - 1;
    // Per-thread coordinates are rebuilt from an opaque copy of the thread id in every layer: left to itself the compiler hoists every loop-invariant
    // address (sixteen 64-bit row addresses per epilogue branch, the transform's, gate's and exchange's LDS addresses ...) out of the layer loop and
    // keeps them all live through the streams (~190 spilled registers)
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const int tid = tid_, lane = tid & 63;
    const int l31 = lane & 31, hi = lane >> 5, col = lane & 15, k4 = lane >> 4;
    const int te = t0 + l31, row0 = pi * 32 + 4 * hi;
    // diagnostics (tools/wn_stack_timeline.py): wall-clock stamps (10 ns) of thread 0 in the middle layer, [workgroup][16]
    const bool stamped = p.dbg != nullptr && li == (NL >> 1) && tid == 0;
    auto stamp = [&](int i) { if (stamped) p.dbg[(long long)gt * 16 + i] = (long long)__builtin_amdgcn_s_memrealtime(); };
    stamp(0);
    float* const red_mine = RED + (wave * 16) * 64 + lane;
    float* const red_peer = RED + ((kh ? pi : pi + NPAIRS) * 16) * 64 + lane;
    // phase A's weight stream: descriptor, ring of four 16-byte sets, three requests ahead (requested before anything else of the layer)
    constexpr int NST = WNF_KS * 6;
    // (wave-uniform; said so explicitly: behind the loop's global stores the compiler reads the table with vector loads and would wrap every
    // buffer load of the streams in a waterfall loop)
    auto uni = [](const float* q) -> const float* {
      const unsigned long long u = (unsigned long long)q;
      return reinterpret_cast<const float*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                                            (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u));
    };
    WnStackLayer ly = p.layers[li];
    ly.wpf = uni(ly.wpf); ly.bias1 = uni(ly.bias1); ly.wp2 = uni(ly.wp2); ly.bias2 = uni(ly.bias2);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ly.wpf), 0, 0x7fffffff, 0x00020000);
    const int w0 = __builtin_amdgcn_readfirstlane((pi * 2 + kh) * NST * 1024);
    const unsigned wlane = (unsigned)lane * 16u;
    float4 a[4];
    auto wload = [&](float4& d, int soff) {
      const wnk_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)wlane, soff, 0);
      d = *reinterpret_cast<const float4*>(&t);
    };
    wload(a[0], w0); wload(a[1], w0 + 1024); wload(a[2], w0 + 2048);
    const float bias_v = tid < 2 * H ? ly.bias1[tid] : ((!last || tid < 3 * H) ? ly.bias2[tid - 2 * H] : 0.f);      // 768 threads = 2 H + 2 H values
    // ---- the neighbours' edges of x_li (what they computed in layer li - 1): tile columns 2, 3 (t0 - 2, t0 - 1) and 36, 37 (t0 + 32, t0 + 33)
    if (li > 0) {
      if (((tid == 0 && has_left) || (tid == 64 && has_right)) && *BAD == 0) {      // (a workgroup that gave up once does not wait again)
        const int* f = p.done + (tid == 0 ? gt - 1 : gt + 1);
        // Bounded by WALL time (the 100 MHz counter), generously: when another process shares the GPU only part of this launch is resident at
        // first, and a workgroup then waits for CUs that the other process' kernels still hold (the low end of the resident window always runs to
        // completion and frees its CUs, so the launch makes progress with as few as n_layers + 1 resident workgroups - but slowly).  A first version
        // counted 2^20 polls (~50 ms) and produced wrong results with two ranks on one GPU.
        const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < li) {
          __builtin_amdgcn_s_sleep(2);
          if (__builtin_amdgcn_s_memrealtime() - t_start > p.timeout) {      // give up, never hang: poison this call, tell the host
            __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            *BAD = 1;
            break;
          }
        }
      }
      stamp(1);
      __syncthreads();
      if (*BAD != 0) mk = __builtin_nanf("");
      // 2 sides x H rows x 2 columns = 768 values: one per thread.  Left neighbour's RIGHT edge -> my columns 2, 3; right neighbour's LEFT edge -> 36, 37
      {
        const int side = tid / (2 * H), r = (tid - side * 2 * H) >> 1, c = tid & 1;
        const bool have = side == 0 ? has_left : has_right;
        if (have) {
          const int src_tile = side == 0 ? gt - 1 : gt + 1;
          const float* hp = p.halo + ((((long long)(li - 1) * ntiles + src_tile) * 2 + (side == 0 ? 1 : 0)) * H + r) * 2 + c;
          XT[r * WNF_XROW + (side == 0 ? 2 : 36) + c] = __hip_atomic_load(hp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      __syncthreads();
    }
    stamp(2);
    // ---- input transform: window q of channel c reads x[t0 + 2q - 2 .. + 3] = tile columns 2q + 2 .. 2q + 7 (8-byte aligned)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = (tid >> 4) + 48 * u, q = tid & 15;
      const float* r = XT + c * WNF_XROW + 2 * q + 2;
      const float2 f0 = *reinterpret_cast<const float2*>(r), f1 = *reinterpret_cast<const float2*>(r + 2), f2 = *reinterpret_cast<const float2*>(r + 4);
      const float d0 = f0.x, d1 = f0.y, d2 = f1.x, d3 = f1.y, d4 = f2.x, d5 = f2.y;
      const float a_ = __builtin_fmaf(-4.f, d2, d4), b_ = __builtin_fmaf(-4.f, d1, d3);
      const float c_ = d4 - d2, e_ = 2.f * (d3 - d1);
      float* o = PLN + c * WNF_NQ + q;
      o[0] = __builtin_fmaf(4.f, d0, __builtin_fmaf(-5.f, d2, d4));
      o[WNF_PLANE] = a_ + b_;
      o[2 * WNF_PLANE] = a_ - b_;
      o[3 * WNF_PLANE] = c_ + e_;
      o[4 * WNF_PLANE] = c_ - e_;
      o[5 * WNF_PLANE] = __builtin_fmaf(4.f, d1, __builtin_fmaf(-5.f, d3, d5));
    }
    BIASL[tid] = bias_v;
    __syncthreads();
    stamp(3);
    // ---- phase A: 24 k-steps x 6 products x 4 row tiles of v_mfma_f32_16x16x4_f32 (wn_layer_f25_kernel's stream)
    wn_f32x4 M[4][6];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int q = 0; q < 6; ++q) M[rt][q] = (wn_f32x4){0.f, 0.f, 0.f, 0.f};
    {
      const unsigned baddr0 = (unsigned)(size_t)PLN + (unsigned)(((kh * 96 + k4) * WNF_NQ + col) * 4);
      const unsigned baddr1 = baddr0 + 3u * WNF_PLANE * 4u;
      float fb[2];
      auto rdb = [&](auto ic) {
        constexpr int I = decltype(ic)::value;
        if constexpr (I < NST) {
          constexpr int KS_ = I / 6, P_ = I % 6;
          constexpr int O = ((P_ % 3) * WNF_PLANE + KS_ * 4 * WNF_NQ) * 4;
          fb[I & 1] = wino_lds_rd<O>(P_ < 3 ? baddr0 : baddr1);
        }
      };
      auto rqw = [&](auto ic) {
        constexpr int I = decltype(ic)::value;
        if constexpr (I < NST) wload(a[I & 3], w0 + I * 1024);
      };
      auto step = [&](auto ic) {
        constexpr int I = decltype(ic)::value;
        constexpr int P_ = I % 6;
        rqw(std::integral_constant<int, I + 3>{});
        {
          float& bq = fb[I & 1];
          if constexpr (I + 1 < NST) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(bq));
          else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq));
        }
        const float4 av = a[I & 3];
        const float bv = fb[I & 1];
        M[0][P_] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv, M[0][P_], 0, 0, 0);
        M[1][P_] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv, M[1][P_], 0, 0, 0);
        M[2][P_] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv, M[2][P_], 0, 0, 0);
        M[3][P_] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv, M[3][P_], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        rdb(std::integral_constant<int, I + 2>{});
        __builtin_amdgcn_sched_barrier(0);
      };
      rdb(std::integral_constant<int, 0>{}); rdb(std::integral_constant<int, 1>{});
      wino_static_for<0, NST>(step);
    }
    stamp(4);
    // ---- output transform of the partial sums; this wave finishes 16-row tile kh of both halves and hands the other to its peer
    // (the 16-row tile this wave keeps and the one it hands to its peer are picked with selects: as a wave-uniform branch per row tile the compiler kept
    // both outcomes of every tile alive - eight spilled register pairs per layer)
    float own[2][4][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float y[2][2];
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          const int rt = 2 * h + o;
          const float s12 = M[rt][1][i] + M[rt][2][i], d12 = M[rt][1][i] - M[rt][2][i];
          const float s34 = M[rt][3][i] + M[rt][4][i], d34 = M[rt][3][i] - M[rt][4][i];
          y[o][0] = M[rt][0][i] + (s12 + s34);
          y[o][1] = __builtin_fmaf(2.f, d34, d12) + M[rt][5][i];
        }
        own[h][i][0] = kh ? y[1][0] : y[0][0];
        own[h][i][1] = kh ? y[1][1] : y[0][1];
        red_mine[(h * 8 + 2 * i) * 64] = kh ? y[0][0] : y[1][0];
        red_mine[(h * 8 + 2 * i + 1) * 64] = kh ? y[0][1] : y[1][1];
      }
    }
    __syncthreads();     // partials published; every wave is done with the planes (the acts tile takes their place)
    stamp(5);
    {
      const float* bias1 = BIASL;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rr = 16 * kh + 4 * k4 + i;
        const int chn = pi * 32 + rr;
        const float bA = bias1[(2 * pi) * 32 + rr], bB = bias1[(2 * pi + 1) * 32 + rr];
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          const int m = 2 * col + o;
          const float vA = (own[0][i][o] + red_peer[(2 * i + o) * 64]) + bA;
          const float vB = (own[1][i][o] + red_peer[(8 + 2 * i + o) * 64]) + bB;
          AT[chn * WNF_AROW + m] = gate_tanh_sigmoid(vA, vB);
        }
      }
    }
    // ---- phase B: res_skip 1x1 on the acts tile: tile pi = x part, tile npairs + pi = skip part (last layer: the only tile)
    f32x16 acc[2][1];
    {
      const float* bias2 = BIASL + 2 * H;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          acc[h][0][i] = (!kh && (h == 0 || !last)) ? bias2[(h * NPAIRS + pi) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi] : 0.f;
    }
    stamp(6);
    __syncthreads();     // acts tile complete, exchange area free again
    stamp(7);
    {
      const unsigned baddr = (unsigned)(size_t)AT + (unsigned)((kh * 96 + hi) * WNF_AROW + l31) * 4u;
      const int wb0 = __builtin_amdgcn_readfirstlane((pi * p.ksg2 + kh * 12) * 1024);
      const int wb1 = __builtin_amdgcn_readfirstlane(((NPAIRS + pi) * p.ksg2 + kh * 12) * 1024);
      if (last) wn_gemm_ct<WNF_AROW, 1, 1, false>(acc, ly.wp2, wb0, wb0, baddr, (unsigned)lane * 16u);
      else wn_gemm_ct<WNF_AROW, 1, 1, true>(acc, ly.wp2, wb0, wb1, baddr, (unsigned)lane * 16u);
    }
    stamp(8);
    // exchange: kh=0 finishes tile 0 (residual part; on the last layer the only tile), kh=1 finishes tile 1 (skip part)
    if (!last) {
#pragma unroll
      for (int q = 0; q < 16; ++q) red_mine[q * 64] = kh ? acc[0][0][q] : acc[1][0][q];
    } else if (kh) {
#pragma unroll
      for (int q = 0; q < 16; ++q) red_mine[q * 64] = acc[0][0][q];
    }
    __syncthreads();
    stamp(9);
    // (assigned on every path - on the last layer K half 1 reads stale words it never uses: left conditionally unassigned the compiler carried the sixteen
    // registers across the layer loop's back edge and spilled them)
    float fin[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) fin[q] = (kh ? acc[1][0][q] : acc[0][0][q]) + red_peer[q * 64];
    // ---- epilogue (modules.py:168-175).  The addresses are rebuilt from opaque copies of the lane's coordinates in every layer: left to itself the
    // compiler hoists the sixteen 64-bit row addresses of each branch out of the layer loop and keeps them live through the streams (~190 spilled registers)
    const int te_ = te, row0_ = row0, l31_ = l31;
    if (!last) {
      if (!kh) {         // x = (x + rs[:H]) * mask: into the resident tile; the tile's first and last two columns also to the halo buffer
        float* xc = XT + row0_ * WNF_XROW + 4 + l31_;
        const bool edge = l31_ < 2 || l31_ >= 30;
        float* hp = p.halo + ((((long long)li * ntiles + gt) * 2 + (l31_ < 2 ? 0 : 1)) * H + row0_) * 2 + (l31_ & 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = (r & 3) + 8 * (r >> 2);
          const float v = (xc[rr * WNF_XROW] + fin[r]) * mk;
          xc[rr * WNF_XROW] = v;
          if (edge) __hip_atomic_store(hp + rr * 2, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the edge stores are acknowledged (write-through) before the barrier behind which the counter rises
      } else if (col_ok) {   // out += rs[H:]
        float* const ob = p.out + (long long)b * p.out_bs + (long long)row0_ * p.out_ld + te_;
        if (li == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) ob[(long long)((r & 3) + 8 * (r >> 2)) * p.out_ld] = fin[r];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) fin[r] += ob[(long long)((r & 3) + 8 * (r >> 2)) * p.out_ld];
#pragma unroll
          for (int r = 0; r < 16; ++r) ob[(long long)((r & 3) + 8 * (r >> 2)) * p.out_ld] = fin[r];
        }
      }
      stamp(10);
      // publish: the edge stores acknowledged (above: the waves that made them waited), then the layer count of this tile
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
      stamp(11);                                       // every wave's edge stores are ordered before the counter; the tile is whole for the next transform
      if (tid == 0 && gt != p.fault_tile) __hip_atomic_store(p.done + gt, li + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (!kh && col_ok) {    // last layer: out = (out + rs) * mask
      float* const ob = p.out + (long long)b * p.out_bs + (long long)row0_ * p.out_ld + te_;
      if (li > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) fin[r] += ob[(long long)((r & 3) + 8 * (r >> 2)) * p.out_ld];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) ob[(long long)((r & 3) + 8 * (r >> 2)) * p.out_ld] = fin[r] * mk;
    }
  }
  // ---- leave: the last workgroup out clears the hand-shake state for the next launch.  Thread 0 alone (it is the only one that writes counters):
  // its counter stores acknowledged, then a TWO-LEVEL exit count - sixteen sub-counters by tile number, whose last arrivals count on a top counter -
  // because 256 device-scope atomics on ONE word serialise at ~170 ns each (a single exit counter cost every launch 45 us).
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this thread's counter stores are acknowledged before it counts itself out
    const int sub = gt & 15;
    const int in_sub = (ntiles - sub + 15) >> 4, nsub = ntiles < 16 ? ntiles : 16;
    if (__hip_atomic_fetch_add(p.exited + 1 + sub, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_sub - 1) {
      if (__hip_atomic_fetch_add(p.exited, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsub - 1) {
        for (int i = 0; i < ntiles; ++i) __hip_atomic_store(p.done + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = 0; i < 17; ++i) __hip_atomic_store(p.exited + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ launch
bool wn_stack_enabled() {
  static const bool on = wn_f25_enabled() && !(getenv("SVOC_WN_STACK") && atoi(getenv("SVOC_WN_STACK")) == 0);      // SVOC_WN_STACK=0: one launch per layer
  return on;
}
// scratch: halo buffer | per-tile layer counters | error word | (64-byte aligned) the layers' pointer table
static size_t wn_stack_table_offset() { return (((size_t)WNS_MAXL * device_cu_count() * 2 * WNF_H * 2 + device_cu_count() + 20) * sizeof(float) + 63) / 64 * 64; }
size_t wn_stack_scratch_bytes() { return wn_stack_table_offset() + WNS_MAXL * sizeof(WnStackLayer); }
constexpr size_t WNS_LDS_BYTES = (size_t)(WNF_LDS_FLOATS + 4 + 4 * WNF_H) * sizeof(float);      // + the give-up flag + the layer's bias vectors
// Workgroups of the launch that can be resident at one time: what the runtime's occupancy calculator says for this kernel's registers, threads and
// LDS (one per CU with these figures), never more than the CU count the scratch area is sized for.  hipLaunchCooperativeKernel would make the same
// check at launch time - and is not taken because the launch has to be capturable: DESIGN.md section 4.4d.
static int wn_stack_capacity() { return std::min(persist_capacity((const void*)wn_stack_f25_kernel, 768, WNS_LDS_BYTES), device_cu_count()); }
// The stack applies while every tile's workgroup can be resident at once (they wait for their neighbours) and the per-layer kernel would be the F(2,5) one.
bool wn_stack_applies(int H, int K, int dil_rate, int NL, int B, int T) {
  if (!wn_stack_enabled() || persist_disabled() || H != WNF_H || K != 5 || dil_rate != 1 || NL < 2 || NL > WNS_MAXL || B <= 0 || T <= 0) return false;
  const long long tiles = (long long)B * ((T + 31) / 32);
  return !wn_layer_prefers_unfused(B, T) && tiles <= wn_stack_capacity() && B <= 65535;
}
// Batches of MORE tiles than the launch can hold resident (32 x 512, 8 x 4096): utterances are independent, so the batch is cut into n equal groups of whole
// utterances and the stack launched once per group, one after the other on the stream (the hand-shake state is left clean by every launch) - taken when every
// group fills at least 7/8 of the launch's capacity, i.e. when n launches of ~41 us per layer beat the per-layer kernel's tiles / CUs rounds of ~46 us.  0: no.
int wn_stack_groups(int H, int K, int dil_rate, int NL, int B, int T) {
  if (!wn_stack_enabled() || persist_disabled() || H != WNF_H || K != 5 || dil_rate != 1 || NL < 2 || NL > WNS_MAXL || B <= 1 || T <= 0) return 0;
  const long long ntx = (T + 31) / 32, tiles = (long long)B * ntx, cap = wn_stack_capacity();
  if (cap <= 0 || tiles <= cap || ntx > cap) return 0;
  const long long n = (tiles + cap - 1) / cap;
  if (B % n != 0) return 0;
  const long long per = (B / n) * ntx;
  return (per <= cap && per * 8 >= cap * 7 && wn_stack_applies(H, K, dil_rate, NL, (int)(B / n), T)) ? (int)n : 0;
}
// in_l / rs_l: the NL layers' packed convolutions; wpf[i]: their F(2,5) images; scratch: wn_stack_scratch_bytes() of device memory.  1 = not eligible.
int launch_wn_stack_f25(const PackedConv* const* in_l, const PackedConv* const* rs_l, const float* const* wpf, int NL, int H, const float* x, long long x_bs,
                        int x_ld, float* out, long long out_bs, int out_ld, const float* mask, long long mask_bs, float* scratch, int B, int T, hipStream_t st, bool first_group) {
  if (!wn_stack_applies(H, in_l[0]->ktaps, 1, NL, B, T) || !scratch) return 1;
  WnStackArgs a{};
  for (int i = 0; i < NL; ++i) {
    const bool last = i == NL - 1;
    if (!wpf[i] || !in_l[i]->paired || in_l[i]->Cin != H || in_l[i]->Cout != 2 * H || in_l[i]->ktaps != 5 || in_l[i]->dil != 1 || rs_l[i]->Cin != H ||
        rs_l[i]->ktaps != 1 || (last ? rs_l[i]->Cout != H : rs_l[i]->Cout != 2 * H) || rs_l[i]->ksg_total != 24 || rs_l[i]->ksg_total != rs_l[0]->ksg_total) return 1;
  }
  a.layers = reinterpret_cast<const WnStackLayer*>(reinterpret_cast<const char*>(scratch) + wn_stack_table_offset());      // written by wn_stack_prepare()
  a.x = x; a.x_bs = x_bs; a.x_ld = x_ld;
  a.out = out; a.out_bs = out_bs; a.out_ld = out_ld;
  a.mask = mask; a.mask_bs = mask_bs;
  a.ksg2 = rs_l[0]->ksg_total; a.NL = NL; a.T = T;
  const int ntx = (T + 31) / 32, ncu = device_cu_count();
  a.halo = scratch;
  a.exited = reinterpret_cast<int*>(scratch + (size_t)WNS_MAXL * ncu * 2 * WNF_H * 2);      // [17]: top counter, sixteen sub-counters
  a.done = a.exited + 17;
  a.err = async_error_word();                                // pinned host memory (misc_kernels.hip): looked at by the next call
  if (!a.err) return 1;
  a.dbg = debug_stamp_buffer();
  a.timeout = persist_timeout_ticks();
  a.fault_tile = persist_fault_tile();
  // The counters are NOT cleared by a memset ahead of the launch: a first version did that, and inside a captured plan replayed while another
  // process shared the GPU the launch found them uncleared (layer counters already raised, i.e. no waiting at all and edges of the launch before - which
  // tests that repeat one input cannot see; a ticket counter tried at the time ran past the grid: HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION).
  // The last workgroup to leave a launch clears them; kernel -> kernel order on the stream does the rest; wn_stack_prepare() zeroes them once.
  double flops = 0, exec = 0;
  for (int i = 0; i < NL; ++i) {
    flops += (in_l[i]->flops_per_col + rs_l[i]->flops_per_col) * (double)B * (double)T;
    exec += (0.6 * in_l[i]->flops_per_col + rs_l[i]->flops_per_col) * (double)B * (double)T;
  }
  stats_add_conv(flops, first_group ? 2 * NL : 0, exec);      // (a batch launched in groups of utterances counts its convolutions once)
  int prof_idx = -1;
  if (prof_enabled()) {
    char d[160];
    snprintf(d, sizeof(d), "stackWN H%-4d k5  d1  N%-7d B%-3d %d layers, one launch, F(2,5)", H, T, B, NL);
    prof_idx = prof_begin(st, d, flops);
  }
  auto kern = wn_stack_f25_kernel;
  SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
  // (hipLaunchCooperativeKernel was measured and dropped: capturable on ROCm 7, but every cooperative launch costs ~0.5 ms - DESIGN.md section 4.4d)
  hipLaunchKernelGGL(kern, dim3(ntx, 1, B), dim3(768), WNS_LDS_BYTES, st, a);
  prof_end(st, prof_idx);
  SVOC_HIP(hipGetLastError());
  return SVOC_OK;
}
// Zeroes the scratch area and writes the layers' pointer table behind it (once, when the WN module is created; the pointers do not move afterwards)
int wn_stack_prepare(float* scratch, const PackedConv* const* in_l, const PackedConv* const* rs_l, const float* const* wpf, int NL, hipStream_t st) {
  if (!scratch || NL > WNS_MAXL) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "wn_stack_prepare: bad arguments");
  if (!async_error_word()) SVOC_FAIL(SVOC_ERR_NOMEM, "wn_stack_prepare: no pinned host memory for the error word");      // allocated here, never inside a caller's stream capture
  WnStackLayer t[WNS_MAXL] = {};
  for (int i = 0; i < NL; ++i) { t[i].wpf = wpf[i]; t[i].bias1 = in_l[i]->bias.f(); t[i].wp2 = rs_l[i]->wp.f(); t[i].bias2 = rs_l[i]->bias.f(); }
  SVOC_HIP(hipMemsetAsync(scratch, 0, wn_stack_table_offset(), st));
  SVOC_HIP(hipMemcpyAsync(reinterpret_cast<char*>(scratch) + wn_stack_table_offset(), t, sizeof(t), hipMemcpyHostToDevice, st));
  SVOC_HIP(hipStreamSynchronize(st));                      // `t` lives on this stack frame
  return SVOC_OK;
}
}  // namespace svoc
