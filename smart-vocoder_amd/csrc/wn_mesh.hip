// A whole WN stack for SHORT inputs in ONE persistent launch (round 5, VERDICT r4 item 3: the 1 x 200 call; reference modules.py:148-176).
//
// Short inputs (wn_small.hip) ran a stack as one launch per layer, six workgroups per 32-column tile: 22 us per layer at 1 x 200, of which ~14 us are
// matrix-pipe time of the ONE CU a workgroup owns (every one of the six recomputes the previous layer's 192 x 192 residual GEMM on 48 columns; the
// F(2,5) in_layer of its 64 rows is 1152 MFMAs on three waves per SIMD) - with 42 of 256 CUs busy.  Here the layer is cut the other way and the cuts
// talk to each other inside one launch:
//   * TWELVE workgroups per 32-column tile; workgroup r owns channels 16 r .. 16 r + 15 of everything: the in_layer's tanh rows 16 r .. and sigmoid
//     rows H + 16 r .., hence those rows of the gated acts; the res_skip's residual rows 16 r .. (x) and skip rows H + 16 r .. (out).  Nothing is
//     recomputed: per layer a workgroup issues 576 + 192 MFMAs of v_mfma_f32_16x16x4_f32, its eight waves splitting the in_layer's K eight ways.
//   * per layer two hand-overs through device memory instead of two launches: the acts rows (all twelve workgroups of a tile need all 192 rows for
//     the 1 x 1) and the x rows (the twelve of the tile and of its two neighbours need them: the k = 5 halo).  Producer: device-coherent stores (sc1:
//     write-through), s_waitcnt vmcnt(0), workgroup barrier, thread 0 raises the workgroup's flag.  Consumer: wave 0 polls the 12 (acts) or 36 (x)
//     flags (a 128-byte line each) with ONE load per poll, barrier, sc1 loads of the tile.  No grid-wide barrier, no atomics on a shared word.
//     x rows are 32 floats on a line of their own; the first and last four columns of a tile travel a second time, packed, for the neighbours.
//   * the layer's weights (72 + 48 registers per lane, 195 KB per workgroup) are requested where nothing waits behind them: the 1 x 1 operands ahead of
//     the in_layer's MFMA stream, the NEXT layer's in_layer operands inside it, each register set right behind the step that consumed it.
//   * measured per layer at 1 x 200 (tools/wn_mesh_timeline.py, profiles/r05_wn_mesh_phase_stamps.txt): 10.4 us = acts hand-over 2.1 + 1 x 1 1.3 + x hand-over
//     2.5 + transform 0.8 + stream and output transform 2.9 + gate 0.3 + 0.5.
//   * the skip sum lives in registers for the whole stack; x rows in a double buffer by layer parity (the neighbour tile may still be reading
//     x_{i-1} when this tile's x_i is ready; x_{i+1} is only written after the neighbour's acts of layer i-1, i.e. after its last read of x_{i-1});
//     acts in a single buffer (a workgroup writes acts_i behind its tile's twelve x_i flags, each raised after that workgroup's last read of
//     acts_{i-1}).
//   * inputs of 11 .. 20 tiles (1 x 512): TWO neighbouring tiles per group of twelve workgroups (template parameter NT), one after the other with the
//     layer's weights in registers once; everything above stays per tile.  17.8 us per layer for the two tiles at 1 x 512.
// Every workgroup of the launch must be resident at once: the launcher takes the stack only while the grid is at most HALF of what the occupancy
// calculator says the device holds of this kernel (two processes sharing a GPU then both fit).  Otherwise wn_stack.hip's rules: waits bounded by
// SVOC_PERSIST_TIMEOUT_MS; a workgroup that gives up makes its mask factor NaN (its x rows, skip sums and so the call's result are NaN - never a finite
// wrong tensor), stops waiting, and raises the host-visible error word.  Flags are cleared by the last workgroup out.
// H = 192, k = 5, dilation 1, no conditioning input; n_layers >= 2.
#include "svoc_internal.h"
#include "wino_common.h"
#include "wn_f25.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace svoc {

constexpr int WNM_R = 12;                                   // workgroups per column tile
constexpr int WNM_AROW = 48;                                // acts tile row stride (columns 0 .. 31 used; 48: the B reads of the 1 x 1 are conflict-free)
constexpr int WNM_MAXL = 16;
constexpr int WNM_FS = 32;                                  // ints between two flags
constexpr int WNM_LDS_FLOATS = WNF_H * WNF_XROW + 6 * WNF_PLANE + WNM_MAXL * 8 + 4;      // x tile | planes | the layer table | the give-up flag
static_assert(WNF_H * WNM_AROW <= 6 * WNF_PLANE && 8 * 16 * 64 <= 6 * WNF_PLANE, "acts tile and reduction area alias the planes");

struct WnMeshLayer {
  const float* wm;                                       // in_layer: mesh image (pack_wn_mesh_kernel)
  const float* bias1;                                    // its bias, paired tile order (PackedConv)
  const float* wrs;                                      // res_skip: 16x16x4 image (pack_wn_rs16_kernel), bias in natural order behind it
  int rs_tiles; int pad_;                                // 24 (last layer: 12) row tiles in that image
};
struct WnMeshArgs {
  const float* x; long long x_bs; int x_ld;             // stack input [B][H][x_ld] (masked by the caller)
  float* out; long long out_bs; int out_ld;             // the stack's output (skip sum * mask)
  const float* mask; long long mask_bs;
  const WnMeshLayer* layers;                            // [NL], device memory
  int NL; int T; int ntx; int ntiles;                   // layers, frames, tiles per utterance, tiles of the batch
  float* xg; int xg_ld; long long xg_bs; long long xg_par;   // x rows: [2 (layer parity)][B][H][32 ntx] (rows start on a 128-byte line)
  float* xh; long long xh_par;                          // their edges again, packed: [2][tiles][2 (first | last four columns)][H][4]
  float* ag; int ag_ld; long long ag_bs;                // acts rows: [B][H][32 ntx]
  int* fa; int* fx;                                     // [tiles][12] x WNM_FS ints (one 128-byte line per flag): acts layers written / x layers written
  int* exited; int* err;                                // [17] two-level exit count; error word
  unsigned long long timeout;                           // bound of a wait in ticks of the 100 MHz wall counter
  int fault_tile;                                       // diagnostics: workgroup 0 of this tile never raises its x flags (-1: none)
  long long* dbg;                                       // diagnostics: [workgroup][16] wall-clock stamps (10 ns) of thread 0 in layer NL / 2 (tools/wn_mesh_timeline.py)
};

typedef unsigned int wnm_u32x4 __attribute__((ext_vector_type(4)));

// device-coherent (sc1) 16-byte load / 4-byte store through a buffer descriptor: what another XCD's workgroup wrote is read from memory, not from
// this XCD's L2 (aux bit 4 = sc1 on gfx94x / gfx950)
__device__ __forceinline__ float4 wnm_ld16_sc1(const __amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  const wnm_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 16);
  return *reinterpret_cast<const float4*>(&t);
}
__device__ __forceinline__ void wnm_st4_sc1(const __amdgpu_buffer_rsrc_t rs, float v, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff, soff, 16);
}

// wave 0: wait until the `n` flags f[0 .. n) (lanes with valid == false excepted) have reached `want`; bounded by `timeout` ticks of wall time.  A wait
// that gives up sets *bad (LDS: the workgroup's results are NaN from then on, and it does not wait again) and raises the host's error word.
__device__ __forceinline__ void wnm_wait(const int* f, int n, bool valid, int want, int* err, volatile int* bad, unsigned long long timeout) {
  if (*bad != 0) return;
  const int lane = threadIdx.x & 63;
  const bool mine = lane < n && valid;
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
  while (true) {
    const int v = mine ? __hip_atomic_load(f + lane * WNM_FS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
    if (__builtin_amdgcn_ballot_w64(v < want) == 0) break;
    __builtin_amdgcn_s_sleep(1);
    if (__builtin_amdgcn_s_memrealtime() - t_start > timeout) {
      if (lane == 0) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); *bad = 1; }
      break;
    }
  }
}

// NT: 32-column tiles per workgroup group (1, or 2 NEIGHBOURING tiles of one utterance handled one after the other with the same weights in registers:
// inputs of 11 .. 20 tiles - 1 x 512 - still fit half the CUs).  Flags, hand-over rows and every wait are per TILE; within a layer a workgroup runs the
// 1 x 1 of its tiles, then their in_layers (a tile's x wait includes the workgroup's own other tile, handed over in the 1 x 1 phase before).
template <int NT>
__global__ void __launch_bounds__(512) wn_mesh_f25_kernel(const WnMeshArgs p) {
  constexpr int H = WNF_H, XROW = WNF_XROW, PLANE = WNF_PLANE, NQ = WNF_NQ;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const XT = lds;                                    // x_i tile [H][40]: columns t0 - 4 .. t0 + 35 (of the tile in hand)
  float* const PLN = lds + H * XROW;                        // V_p [6][H][16]
  float* const AT = PLN;                                    // acts tile [H][48] (columns t0 .. t0 + 31), in the 1 x 1 phase
  float* const RED = PLN;                                   // [8 waves][16][64] partial outputs, behind the stream
  // the layer table, copied once: read from device memory per layer its pointers were two dependent ~0.8 us round trips ahead of every weight request
  WnMeshLayer* const TBL = reinterpret_cast<WnMeshLayer*>(PLN + 6 * PLANE);
  static_assert(sizeof(WnMeshLayer) == 32, "eight words per layer");
  volatile int* const BAD = reinterpret_cast<volatile int*>(PLN + 6 * PLANE + WNM_MAXL * 8);
  if (threadIdx.x < (unsigned)p.NL * 8) reinterpret_cast<int*>(TBL)[threadIdx.x] = reinterpret_cast<const int*>(p.layers)[threadIdx.x];
  if (threadIdx.x == 0) *BAD = 0;
  __syncthreads();

  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwg = gridDim.x;
  const int g = xcd_linear((int)blockIdx.x, nwg, 1);       // consecutive workgroups (a group's twelve, neighbouring groups) on one XCD
  const int gp = g / WNM_R, r = g - gp * WNM_R;
  const int ntx = p.ntx, npx = (ntx + NT - 1) / NT;        // tiles / tile groups per utterance
  const int b = gp / npx, px = gp - b * npx;
  const int tile0 = px * NT;
  const int nact = __builtin_amdgcn_readfirstlane(ntx - tile0 < NT ? ntx - tile0 : NT);      // tiles of this group (the last group of an utterance may hold fewer)
  const int gt0 = b * ntx + tile0;                          // flags, packed edges: by tile over the whole batch
  const int ntiles = p.ntiles;
  const int NL = p.NL, T = p.T;
  const __amdgpu_buffer_rsrc_t xg_rs = __builtin_amdgcn_make_buffer_rsrc(p.xg, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t xh_rs = __builtin_amdgcn_make_buffer_rsrc(p.xh, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t ag_rs = __builtin_amdgcn_make_buffer_rsrc(p.ag, 0, 0x7fffffff, 0x00020000);
  const int rt2 = wave & 1, nt = (wave >> 1) & 1;           // the 1 x 1: waves 0 .. 3 = (residual | skip tile, column half)
  float skip[NT][4];                                        // waves 1, 3: out rows 16 r + 4 k4 + i, column t0 + 16 nt + col
  float xown[NT][4];                                        // waves 0, 2: this workgroup's own x rows (the residual input of the next 1 x 1)
  float mk_rs[NT];                                          // waves 0 .. 3: mask of column t0 + 16 nt + col
#pragma unroll
  for (int j = 0; j < NT; ++j) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { skip[j][i] = 0.f; xown[j][i] = 0.f; }
    const int t = (tile0 + j) * 32 + 16 * nt + (int)(threadIdx.x & 15);
    mk_rs[j] = (wave < 4 && j < nact && t < T) ? p.mask[(long long)b * p.mask_bs + t] : 0.f;
  }

  auto uni = [](const float* q) -> const float* {
    const unsigned long long u = (unsigned long long)q;
    return reinterpret_cast<const float*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                                          (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u));
  };

  // the in_layer weights of this wave: 18 sixteen-byte loads = the A operands of its 72 MFMAs per tile
  float4 wv[18];
  {
    const float* wm = uni(TBL[0].wm);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wm), 0, 0x7fffffff, 0x00020000);
    const int w0 = __builtin_amdgcn_readfirstlane((r * 8 + wave) * 18 * 1024);
    const int vo = (int)(threadIdx.x & 63) * 16;
#pragma unroll
    for (int l = 0; l < 18; ++l) {
      const wnm_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, w0 + l * 1024, 0);
      wv[l] = *reinterpret_cast<const float4*>(&t);
    }
  }
  // the 1 x 1 of a layer: waves 0 .. 3; A operands (12 sixteen-byte loads) and the four bias values of this lane's rows
  float4 aw[12];
  float4 rsb = make_float4(0.f, 0.f, 0.f, 0.f);
  auto request_rs = [&](int layer) {
    const bool last_ = layer == NL - 1;
    if (wave < 4 && (rt2 == 1 || !last_)) {
      const WnMeshLayer lp = TBL[layer];
      const float* wrs = uni(lp.wrs);
      const int rs_tiles = __builtin_amdgcn_readfirstlane(lp.rs_tiles);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wrs), 0, 0x7fffffff, 0x00020000);
      const int ti = (rt2 == 1 && !last_) ? WNM_R + r : r;
      const int wb = __builtin_amdgcn_readfirstlane(ti * 12 * 1024);
      const int ln = (int)(threadIdx.x & 63);
#pragma unroll
      for (int ks4 = 0; ks4 < 12; ++ks4) {
        const wnm_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, ln * 16, wb + ks4 * 1024, 0);
        aw[ks4] = *reinterpret_cast<const float4*>(&t);
      }
      // bias (natural order behind the image): rows 16 r + 4 k4 .. + 3 of the residual part, or of the skip part (last layer: the only part)
      const int bo = __builtin_amdgcn_readfirstlane((rs_tiles * 12 * 256 + ((rt2 == 1 && !last_) ? WNF_H : 0) + 16 * r) * 4);
      const wnm_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, (ln >> 4) * 16, bo, 0);
      rsb = *reinterpret_cast<const float4*>(&t);
    }
  };

  for (int li = 0; li <= NL; ++li) {
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));                          // (per-layer addresses are rebuilt, not hoisted and kept live: wn_stack.hip)
    const int tid = tid_, lane = tid & 63;
    const int col = lane & 15, k4 = lane >> 4;
    const bool stamped = p.dbg && li == NL / 2 && tid == 0;
    auto stamp = [&](int i) { if (stamped) p.dbg[(long long)g * 16 + i] = (long long)__builtin_amdgcn_s_memrealtime(); };      // (of the group's first tile)
    const bool last = li == NL;                             // the pass behind the last layer: its 1 x 1 (H rows, all of them skip) only

    // =============================================================== res_skip of layer li - 1, tile by tile
    if (li > 0) {
      const bool gemm_wave = wave < 4 && (rt2 == 1 || !last);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (j < nact) {
          const int gt = gt0 + j, t0 = (tile0 + j) * 32;
          if (j == 0) stamp(0);
          if (wave == 0) wnm_wait(p.fa + gt * WNM_R * WNM_FS, WNM_R, true, li, p.err, BAD, p.timeout);
          if (j == 0) stamp(1);
          __syncthreads();
          if (*BAD != 0) {
#pragma unroll
            for (int jj = 0; jj < NT; ++jj) mk_rs[jj] = __builtin_nanf("");
          }
          {   // acts_{li-1} tile: H rows x 32 columns = 1536 sixteen-byte groups, three per thread
            const int sb = __builtin_amdgcn_readfirstlane((int)(((long long)b * p.ag_bs + t0) * 4));
#pragma unroll
            for (int u = 0; u < 3; ++u) {
              const int it = tid + 512 * u;
              const int c = it >> 3, g4 = it & 7;
              const float4 q = wnm_ld16_sc1(ag_rs, (c * p.ag_ld + 4 * g4) * 4, sb);
              *reinterpret_cast<float4*>(AT + c * WNM_AROW + 4 * g4) = q;
            }
          }
          __syncthreads();
          if (j == 0) stamp(2);
          if (gemm_wave) {
            wn_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* bp = AT + k4 * WNM_AROW + 16 * nt + col;
#pragma unroll
            for (int ks4 = 0; ks4 < 12; ++ks4) {
              const float4 a = aw[ks4];
              const float* bq = bp + 16 * ks4 * WNM_AROW;
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bq[0], acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bq[4 * WNM_AROW], acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bq[8 * WNM_AROW], acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bq[12 * WNM_AROW], acc, 0, 0, 0);
            }
            if (rt2 == 1) {
#pragma unroll
              for (int i = 0; i < 4; ++i) skip[j][i] += acc[i] + wino_pick(rsb, i);
            } else {   // x_li = (x_{li-1} + rs) * mask: this workgroup's sixteen rows of the centre columns -> the buffer of parity li, edges also packed
              const int sb = __builtin_amdgcn_readfirstlane((int)(((long long)(li & 1) * p.xg_par + (long long)b * p.xg_bs + t0) * 4));
              const int tc = 16 * nt + col;                 // tile column
              const bool edge = tc < 4 || tc >= 28;
              const int hb = __builtin_amdgcn_readfirstlane((int)(((long long)(li & 1) * p.xh_par + (long long)gt * 2 * H * 4) * 4));
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const int ch = 16 * r + 4 * k4 + i;
                const float v = (xown[j][i] + (acc[i] + wino_pick(rsb, i))) * mk_rs[j];
                xown[j][i] = v;
                wnm_st4_sc1(xg_rs, v, (ch * p.xg_ld + tc) * 4, sb);
                if (edge) wnm_st4_sc1(xh_rs, v, (((tc >= 28 ? 1 : 0) * H + ch) * 4 + (tc & 3)) * 4, hb);
              }
            }
          }
          if (!last) {
            if (j == 0) stamp(3);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the x stores are acknowledged (write-through) ...
            __syncthreads();                                      // (... and every wave is done with the acts tile)
            if (tid == 0 && !(gt == p.fault_tile && r == 0)) __hip_atomic_store(p.fx + (gt * WNM_R + r) * WNM_FS, li, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ... before the flag
            if (j == 0) stamp(4);
          } else __syncthreads();
        }
      }
      if (last) break;
    }

    // =============================================================== in_layer + gate of layer li, tile by tile
    // bias of this thread's (tanh, sigmoid) pair in the gate: ahead of the weight requests
    float2 gbias;
    {
      const float* bias1 = uni(TBL[li].bias1);
      const int ch = 16 * r + 4 * k4 + (wave >> 1);
      gbias.x = bias1[64 * (ch >> 5) + (ch & 31)];
      gbias.y = bias1[64 * (ch >> 5) + 32 + (ch & 31)];
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (j < nact) {
        const int tile = tile0 + j, gt = gt0 + j, t0 = tile * 32;
        if (li == 0) {   // the stack's input: all H channels, columns [t0 - 4, t0 + 36), zero outside [0, T)
          constexpr int R4 = XROW / 4, total = H * R4;
          const float* xb = p.x + (long long)b * p.x_bs;
          const bool vec = ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0) && (p.x_ld & 3) == 0 && (p.x_bs & 3) == 0;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int it = tid + 512 * u;
            if (it < total) {
              const int c = it / R4, g4 = it - c * R4;
              const int t = t0 - 4 + 4 * g4;
              const float* row = xb + (long long)c * p.x_ld;
              float4 q;
              if (vec && t >= 0 && t + 3 < T) q = *reinterpret_cast<const float4*>(row + t);
              else {
                q.x = (t >= 0 && t < T) ? row[t] : 0.f;
                q.y = (t + 1 >= 0 && t + 1 < T) ? row[t + 1] : 0.f;
                q.z = (t + 2 >= 0 && t + 2 < T) ? row[t + 2] : 0.f;
                q.w = (t + 3 >= 0 && t + 3 < T) ? row[t + 3] : 0.f;
              }
              *reinterpret_cast<float4*>(XT + c * XROW + 4 * g4) = q;
            }
          }
        } else {
          if (wave == 0) {   // x_li of this tile's and the two neighbours' workgroups: 36 flags, one load per poll
            const int d = lane / WNM_R - 1;
            const bool valid = lane < 3 * WNM_R && tile + d >= 0 && tile + d < ntx;
            wnm_wait(p.fx + (gt - 1) * WNM_R * WNM_FS, 3 * WNM_R, valid, li, p.err, BAD, p.timeout);
          }
          if (j == 0) stamp(5);
          __syncthreads();
          if (*BAD != 0) {
#pragma unroll
            for (int jj = 0; jj < NT; ++jj) mk_rs[jj] = __builtin_nanf("");
          }
          // centre: H rows x 32 columns (one 128-byte line per row), three sixteen-byte groups per thread
          const int sb = __builtin_amdgcn_readfirstlane((int)(((long long)(li & 1) * p.xg_par + (long long)b * p.xg_bs + t0) * 4));
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            const int it = tid + 512 * u;
            const int c = it >> 3, g4 = it & 7;
            const float4 q = wnm_ld16_sc1(xg_rs, (c * p.xg_ld + 4 * g4) * 4, sb);
            *reinterpret_cast<float4*>(XT + c * XROW + 4 + 4 * g4) = q;
          }
          // edges: the left neighbour's LAST four columns -> tile columns 0 .. 3, the right neighbour's FIRST four -> 36 .. 39 (packed: 3 KB each);
          // outside the utterance: the convolution's zero padding
          if (tid < 2 * H) {
            const int side = tid >= H ? 1 : 0, c = tid - side * H;
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            if (side == 0 ? tile > 0 : tile + 1 < ntx) {
              const int hb = __builtin_amdgcn_readfirstlane((int)((long long)(li & 1) * p.xh_par * 4));
              q = wnm_ld16_sc1(xh_rs, (((gt + (side ? 1 : -1)) * 2 + (side ? 0 : 1)) * H + c) * 16, hb);
            }
            *reinterpret_cast<float4*>(XT + c * XROW + (side ? 36 : 0)) = q;
          }
        }
        __syncthreads();
        if (j == 0) stamp(6);
        if (li == 0 && wave < 4 && rt2 == 0) {   // this workgroup's own rows of the input: the residual input of the first 1 x 1
#pragma unroll
          for (int i = 0; i < 4; ++i) xown[j][i] = XT[(16 * r + 4 * k4 + i) * XROW + 4 + 16 * nt + col];
        }
        // ---- input transform (wn_fused.hip): window q of channel c reads tile columns 2q + 2 .. 2q + 7
#pragma unroll
        for (int u = 0; u < 6; ++u) {
          const int c = (tid >> 4) + 32 * u, q = tid & 15;
          const float* rp = XT + c * XROW + 2 * q + 2;
          const float2 f0 = *reinterpret_cast<const float2*>(rp), f1 = *reinterpret_cast<const float2*>(rp + 2), f2 = *reinterpret_cast<const float2*>(rp + 4);
          const float d0 = f0.x, d1 = f0.y, d2 = f1.x, d3 = f1.y, d4 = f2.x, d5 = f2.y;
          const float a_ = __builtin_fmaf(-4.f, d2, d4), b_ = __builtin_fmaf(-4.f, d1, d3);
          const float c_ = d4 - d2, e_ = 2.f * (d3 - d1);
          float* o = PLN + c * NQ + q;
          o[0] = __builtin_fmaf(4.f, d0, __builtin_fmaf(-5.f, d2, d4));
          o[PLANE] = a_ + b_;
          o[2 * PLANE] = a_ - b_;
          o[3 * PLANE] = c_ + e_;
          o[4 * PLANE] = c_ - e_;
          o[5 * PLANE] = __builtin_fmaf(4.f, d1, __builtin_fmaf(-5.f, d3, d5));
        }
        __syncthreads();
        if (j == 0) stamp(7);
        // ---- the stream: this wave's 24 channels (six k-steps) x six products x (tanh tile, sigmoid tile) = 72 MFMAs; 36 fragment reads two ahead
        // WEIGHT TRAFFIC (195 KB per workgroup and layer = 1.3 us of the CU's 64 bytes per clock) is requested where nothing waits behind it.  Loads return in
        // order - requested ahead of a flag's s_waitcnt, of a poll or of a tile they hold those back by as much - and a wave that issues eighteen 1 KB loads
        // in a row stands still until the address path has taken them (requested in one go ahead of or behind the stream they made stream + output transform
        // 4.2 us instead of 2.3: tools/wn_mesh_timeline.py).  So: the 1 x 1 operands of this layer ahead of the first tile's stream (their registers are free
        // since the last 1 x 1), and the NEXT layer's in_layer operands INSIDE the last tile's stream - each sixteen-byte register set is requested again right
        // behind the step that consumed it, one load per four MFMAs (128 matrix-pipe cycles: with two waves per SIMD exactly the address path's rate).
        // (the stream's operands are marked as used first: they were requested in the iteration before, and across the loop's back edge the compiler no
        // longer knows how old they are - its wait for them would be a wait for everything in flight, the new requests included)
        if (j == 0) {
#pragma unroll
          for (int l = 0; l < 18; ++l) asm volatile("" : "+v"(wv[l].x), "+v"(wv[l].y), "+v"(wv[l].z), "+v"(wv[l].w));
          request_rs(li);
        }
        wn_f32x4 M[2][6];
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
          for (int q = 0; q < 6; ++q) M[tl][q] = (wn_f32x4){0.f, 0.f, 0.f, 0.f};
        {
          constexpr int NST = 36;
          const unsigned baddr = (unsigned)(size_t)PLN + (unsigned)(((24 * wave + k4) * NQ + col) * 4);
          const float* wmn = uni(TBL[li + 1 < NL ? li + 1 : li].wm);      // (the last layer requests its own again: no data-dependent branch in the stream)
          const __amdgpu_buffer_rsrc_t rsn = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wmn), 0, 0x7fffffff, 0x00020000);
          const int w0n = __builtin_amdgcn_readfirstlane((r * 8 + wave) * 18 * 1024);
          const int von = lane * 16;
          const bool reload = j == nact - 1;                 // (wave-uniform)
          float fb[2];
          auto rdb = [&](auto ic) {
            constexpr int I = decltype(ic)::value;
            if constexpr (I < NST) {
              constexpr int KS_ = I / 6, P_ = I % 6;
              fb[I & 1] = wino_lds_rd<(P_ * PLANE + KS_ * 4 * NQ) * 4>(baddr);
            }
          };
          auto step = [&](auto ic) {
            constexpr int I = decltype(ic)::value;
            constexpr int P_ = I % 6;
            {
              float& bq = fb[I & 1];
              if constexpr (I + 1 < NST) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(bq));
              else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq));
            }
            const float4 av = wv[I / 2];
            const float bv = fb[I & 1];
            M[0][P_] = __builtin_amdgcn_mfma_f32_16x16x4f32(wino_pick(av, (2 * I) & 3), bv, M[0][P_], 0, 0, 0);
            M[1][P_] = __builtin_amdgcn_mfma_f32_16x16x4f32(wino_pick(av, (2 * I + 1) & 3), bv, M[1][P_], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            rdb(std::integral_constant<int, I + 2>{});
            if constexpr (I & 1) {                           // wv[I / 2] has been consumed (by the group's last tile): the next layer's set takes its place
              if (NT == 1 || reload) {
                const wnm_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsn, von, w0n + (I / 2) * 1024, 0);
                wv[I / 2] = *reinterpret_cast<const float4*>(&t);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          };
          rdb(std::integral_constant<int, 0>{}); rdb(std::integral_constant<int, 1>{});
          wino_static_for<0, NST>(step);
        }
        if (j == 0) stamp(8);
        __syncthreads();                                     // every wave is done with the planes: the reduction area takes their place
        // ---- output transform of the partial sums -> RED[wave][jj][lane], jj = 8 tile + 2 i + o
        {
          float* rm = RED + (wave * 16) * 64 + lane;
#pragma unroll
          for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float s12 = M[tl][1][i] + M[tl][2][i], d12 = M[tl][1][i] - M[tl][2][i];
              const float s34 = M[tl][3][i] + M[tl][4][i], d34 = M[tl][3][i] - M[tl][4][i];
              rm[(8 * tl + 2 * i) * 64] = M[tl][0][i] + (s12 + s34);
              rm[(8 * tl + 2 * i + 1) * 64] = __builtin_fmaf(2.f, d34, d12) + M[tl][5][i];
            }
        }
        __syncthreads();
        if (j == 0) stamp(9);
        // ---- reduction over the eight K parts + bias + gate: wave w takes (i, o) = (w / 2, w % 2) of every lane
        {
          float vA = 0.f, vB = 0.f;
#pragma unroll
          for (int w = 0; w < 8; ++w) {
            vA += RED[(w * 16 + wave) * 64 + lane];
            vB += RED[(w * 16 + 8 + wave) * 64 + lane];
          }
          const int ch = 16 * r + 4 * k4 + (wave >> 1);
          const int m = 2 * col + (wave & 1);
          vA += gbias.x;
          vB += gbias.y;
          const int sb = __builtin_amdgcn_readfirstlane((int)(((long long)b * p.ag_bs + t0) * 4));
          wnm_st4_sc1(ag_rs, gate_tanh_sigmoid(vA, vB), (ch * p.ag_ld + m) * 4, sb);
        }
        if (j == 0) stamp(10);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                     // (also: the reduction area is free for the next tile's planes)
        if (tid == 0) __hip_atomic_store(p.fa + (gt * WNM_R + r) * WNM_FS, li + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (j == 0) stamp(11);
      }
    }
  }

  // ---- out = skip sum * mask (modules.py:175): rows 16 r + 4 k4 + i, columns t0 + 16 nt + col
  if (wave < 4 && rt2 == 1) {
    const int lane = threadIdx.x & 63, col = lane & 15, k4 = lane >> 4;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int t = (tile0 + j) * 32 + 16 * nt + col;
      if (j < nact && t < T) {
#pragma unroll
        for (int i = 0; i < 4; ++i) p.out[(long long)b * p.out_bs + (long long)(16 * r + 4 * k4 + i) * p.out_ld + t] = skip[j][i] * mk_rs[j];
      }
    }
  }
  // ---- leave: the last workgroup out clears the flags for the next launch (two-level exit count: wn_stack.hip)
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int sub = g & 15;
    const int in_sub = (nwg - sub + 15) >> 4, nsub = nwg < 16 ? nwg : 16;
    if (__hip_atomic_fetch_add(p.exited + 1 + sub, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_sub - 1) {
      if (__hip_atomic_fetch_add(p.exited, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsub - 1) {
        for (int i = 0; i < ntiles * WNM_R; ++i) {
          __hip_atomic_store(p.fa + i * WNM_FS, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(p.fx + i * WNM_FS, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (int i = 0; i < 17; ++i) __hip_atomic_store(p.exited + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// Mesh image of one in_layer from its F(2,5) image (pack_wn_f25_kernel: [pair 6][K half 2][k-step 24][product 6][lane 64][row tile 4]): a pure
// permutation.  [workgroup r 12][wave 8][load 18][lane 64][4]: component f = 4 load + j of lane (k4, rr) is U_prod[tile * H + 16 r + rr][24 wave + 4 ks + k4]
// with ks = f / 12, prod = (f % 12) / 2, tile = f % 2 (0: tanh row, 1: sigmoid row) - the order the stream consumes them in.
__global__ void pack_wn_mesh_kernel(const float* __restrict__ f25, float* __restrict__ img, int total) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int j = e & 3, lane = (e >> 2) & 63;
  int rest = e >> 8;
  const int l = rest % 18; rest /= 18;
  const int w = rest & 7, r = rest >> 3;
  const int f = 4 * l + j, ks = f / 12, prod = (f % 12) >> 1, tl = f & 1;
  const int c = 16 * r + (lane & 15), chan = 24 * w + 4 * ks + (lane >> 4);
  const int pi = c >> 5, rt = (tl << 1) | ((c & 31) >> 4), kh = chan / 96, ks24 = (chan % 96) >> 2;
  // (chan & 3) == lane >> 4 and c & 15 == lane & 15: the lane is the same in both images
  img[e] = f25[(((((long long)pi * 2 + kh) * WNF_KS + ks24) * 6 + prod) * 64 + lane) * 4 + rt];
}

// ------------------------------------------------------------------------------------------------ host side
bool wn_mesh_enabled() {
  static const bool on = wn_f25_enabled() && !(getenv("SVOC_WN_MESH") && atoi(getenv("SVOC_WN_MESH")) == 0);      // SVOC_WN_MESH=0: one launch per layer (wn_small.hip)
  return on;
}
// twelve workgroups each: the grid stays within half the CUs - and within half of what the occupancy calculator says the device holds of the kernel
// (asked per launch form, cached; the scratch area is sized by the CU count)
static int wnm_max_groups() { return device_cu_count() / 2 / WNM_R; }
static int wnm_capacity(int NT) {
  const void* k = NT == 1 ? (const void*)wn_mesh_f25_kernel<1> : (const void*)wn_mesh_f25_kernel<2>;
  return std::min(persist_capacity(k, 512, (size_t)WNM_LDS_FLOATS * sizeof(float)), device_cu_count()) / 2 / WNM_R;
}
static int wnm_max_tiles() { return 2 * wnm_max_groups(); }                 // up to two tiles per group
// scratch: x rows (two parities) | their packed edges (two parities) | acts rows | flags fa, fx (a line each) | exit counters [17] | (64-byte aligned) layer table
static size_t wnm_x_floats() { return (size_t)2 * WNF_H * 32 * wnm_max_tiles() + (size_t)2 * wnm_max_tiles() * 2 * WNF_H * 4; }      // rows + packed edges, two parities each
static size_t wnm_a_floats() { return (size_t)WNF_H * 32 * wnm_max_tiles(); }
static size_t wnm_flag_offset() { return (wnm_x_floats() + wnm_a_floats()) * sizeof(float); }
static size_t wnm_table_offset() { return (wnm_flag_offset() + ((size_t)2 * WNM_R * wnm_max_tiles() * WNM_FS + 32) * sizeof(int) + 63) / 64 * 64; }
size_t wn_mesh_scratch_bytes() { return wnm_table_offset() + WNM_MAXL * sizeof(WnMeshLayer); }
size_t wn_mesh_image_floats() { return (size_t)WNM_R * 8 * 18 * 256; }
// tiles per workgroup group for (B, T): 1 up to ten tiles, 2 (neighbouring tiles of one utterance) up to ten groups; 0 = the launch does not apply
static int wnm_tiles_per_group(int B, int T) {
  const long long vb = std::max(B, variant_batch(B)), ntx = (T + 31) / 32;
  if (vb * ntx <= wnm_max_groups()) return vb * ntx <= wnm_capacity(1) ? 1 : 0;
  if (vb * ((ntx + 1) / 2) <= wnm_max_groups()) return vb * ((ntx + 1) / 2) <= wnm_capacity(2) ? 2 : 0;
  return 0;
}
bool wn_mesh_supported(int H, int K, int dil_rate, int NL) {      // the module: whatever the shapes that will come
  return wn_mesh_enabled() && H == WNF_H && K == 5 && dil_rate == 1 && NL >= 2 && NL <= WNM_MAXL;
}
bool wn_mesh_applies(int H, int K, int dil_rate, int NL, int B, int T) {
  return wn_mesh_supported(H, K, dil_rate, NL) && !persist_disabled() && B > 0 && T > 0 && wnm_tiles_per_group(B, T) != 0;
}
int pack_wn_mesh(DevBuf& img, const float* f25, hipStream_t st) {
  if (!wn_mesh_enabled() || !f25) return SVOC_OK;
  const int total = (int)wn_mesh_image_floats();
  SVOC_TRY(img.ensure((size_t)(total + 1024) * sizeof(float)));
  hipLaunchKernelGGL(pack_wn_mesh_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, f25, img.f(), total);
  SVOC_HIP(hipGetLastError());
  return SVOC_OK;
}
// Zeroes the scratch area (pad columns, flags, counters) and writes the layer table (once, when the WN module is created)
int wn_mesh_prepare(float* scratch, const PackedConv* const* in_l, const float* const* wm, const float* const* wrs, int NL, hipStream_t st) {
  if (!scratch || NL > WNM_MAXL || NL < 2) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "wn_mesh_prepare: bad arguments");
  if (!async_error_word()) SVOC_FAIL(SVOC_ERR_NOMEM, "wn_mesh_prepare: no pinned host memory for the error word");      // allocated here, never inside a caller's stream capture
  WnMeshLayer t[WNM_MAXL] = {};
  for (int i = 0; i < NL; ++i) { t[i].wm = wm[i]; t[i].bias1 = in_l[i]->bias.f(); t[i].wrs = wrs[i]; t[i].rs_tiles = i == NL - 1 ? 12 : 24; }
  SVOC_HIP(hipMemsetAsync(scratch, 0, wnm_table_offset(), st));
  SVOC_HIP(hipMemcpyAsync(reinterpret_cast<char*>(scratch) + wnm_table_offset(), t, sizeof(t), hipMemcpyHostToDevice, st));
  SVOC_HIP(hipStreamSynchronize(st));                      // `t` lives on this stack frame
  return SVOC_OK;
}
// 1 = not eligible (the caller runs the per-layer chain)
int launch_wn_mesh_f25(const PackedConv* const* in_l, const PackedConv* const* rs_l, int NL, int H, const float* x, long long x_bs, int x_ld, float* out,
                       long long out_bs, int out_ld, const float* mask, long long mask_bs, float* scratch, int B, int T, hipStream_t st) {
  if (!scratch || !wn_mesh_applies(H, in_l[0]->ktaps, 1, NL, B, T)) return 1;
  for (int i = 0; i < NL; ++i) {
    const bool last = i == NL - 1;
    if (!in_l[i]->paired || in_l[i]->Cin != H || in_l[i]->Cout != 2 * H || in_l[i]->ktaps != 5 || in_l[i]->dil != 1 || rs_l[i]->Cin != H || rs_l[i]->ktaps != 1 ||
        (last ? rs_l[i]->Cout != H : rs_l[i]->Cout != 2 * H)) return 1;
  }
  WnMeshArgs a{};
  a.x = x; a.x_bs = x_bs; a.x_ld = x_ld;
  a.out = out; a.out_bs = out_bs; a.out_ld = out_ld;
  a.mask = mask; a.mask_bs = mask_bs;
  a.layers = reinterpret_cast<const WnMeshLayer*>(reinterpret_cast<const char*>(scratch) + wnm_table_offset());
  a.NL = NL; a.T = T; a.ntx = (T + 31) / 32; a.ntiles = a.ntx * B;
  const int NT = wnm_tiles_per_group(B, T);
  a.xg = scratch; a.xg_ld = 32 * a.ntx; a.xg_bs = (long long)H * a.xg_ld; a.xg_par = a.xg_bs * B;
  a.xh = scratch + (size_t)2 * H * 32 * wnm_max_tiles(); a.xh_par = (long long)a.ntx * B * 2 * H * 4;
  a.ag = scratch + wnm_x_floats(); a.ag_ld = 32 * a.ntx; a.ag_bs = (long long)H * a.ag_ld;
  if ((long long)a.ntx * B > wnm_max_tiles()) return 1;
  a.fa = reinterpret_cast<int*>(reinterpret_cast<char*>(scratch) + wnm_flag_offset());
  a.fx = a.fa + WNM_R * wnm_max_tiles() * WNM_FS;
  a.exited = a.fx + WNM_R * wnm_max_tiles() * WNM_FS;
  a.err = async_error_word();                                // pinned host memory (misc_kernels.hip): looked at by the next call
  if (!a.err) return 1;
  a.dbg = debug_stamp_buffer();
  a.timeout = persist_timeout_ticks();
  a.fault_tile = persist_fault_tile();
  double flops = 0, exec = 0;
  for (int i = 0; i < NL; ++i) {
    flops += (in_l[i]->flops_per_col + rs_l[i]->flops_per_col) * (double)B * (double)T;
    exec += (0.6 * in_l[i]->flops_per_col + rs_l[i]->flops_per_col) * (double)B * (double)T;
  }
  stats_add_conv(flops, 2 * NL, exec);
  int prof_idx = -1;
  if (prof_enabled()) {
    char d[160];
    snprintf(d, sizeof(d), "meshWN H%-4d k5  d1  N%-7d B%-3d %d layers, one launch (%d tile%s per group), F(2,5)", H, T, B, NL, NT, NT > 1 ? "s" : "");
    prof_idx = prof_begin(st, d, flops);
  }
  const unsigned grid = (unsigned)(((a.ntx + NT - 1) / NT) * B * WNM_R);
  if (NT == 1) {
    auto kern = wn_mesh_f25_kernel<1>;
    SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), (size_t)WNM_LDS_FLOATS * sizeof(float), st, a);
  } else {
    auto kern = wn_mesh_f25_kernel<2>;
    SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), (size_t)WNM_LDS_FLOATS * sizeof(float), st, a);
  }
  prof_end(st, prof_idx);
  SVOC_HIP(hipGetLastError());
  return SVOC_OK;
}
}  // namespace svoc
