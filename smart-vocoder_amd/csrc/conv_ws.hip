// Persistent, wave-specialised variant of the implicit-GEMM convolution (same math, packing and
// epilogues as conv_mfma.hip).
//
// Why: with identical workgroups the chip runs read -> MFMA -> write bulk-synchronously, and a
// wave that both streams activations and consumes weight fragments serialises them on its single in-order
// vmcnt queue.  Here one 8-wave workgroup per CU walks a list of output tiles and splits the roles:
//   waves 0-3  consumers : activation fragments from LDS, weight fragments straight from L2 (packed order, one
//                          group ahead, ping-pong registers) -> v_mfma_f32_32x32x2_f32 -> fused epilogue stores
//   waves 4-7  activation producers: wave 4+w owns the 32-channel chunks cc == w (mod 4): requests the
//                          [32][BN+halo] tile three chunks ahead (branch-free clamped loads), holds it in registers
//                          across two barriers and publishes it (leaky-relu / mask / zero padding applied on the way)
//                          one chunk before it is consumed; 4-slot LDS ring
// One raw s_barrier per chunk hands LDS tiles from producers to consumers.  conv_ws2_kernel (below) adds a second
// consumer set.  Both are opt-in experiments (SVOC_WS=1 / 2): correct, at parity with conv_mfma.hip at best.
#include "svoc_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace svoc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int WS_HALO_MAX = 56;
constexpr unsigned F_VECST_WS = 1u << 16;

__device__ __forceinline__ float ws_pick4(const float4& v, int s) { return s == 0 ? v.x : (s == 1 ? v.y : (s == 2 ? v.z : v.w)); }
__device__ __forceinline__ float ws_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// Workgroup barriers without the memory fence of __syncthreads(): a fence would make the consumers wait for their
// global stores (and the producers for their in-flight global loads) at every hand-off.  LDS data is published by
// waiting for the writer's own ds_writes (lgkmcnt) before the barrier.
__device__ __forceinline__ void ws_publish_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}
__device__ __forceinline__ void ws_consume_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own LDS reads have returned (they feed the MFMAs anyway)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Epilogue of one finished tile (bias is already inside acc).
template <int MR, int NR>
__device__ __forceinline__ void ws_epilogue(const ConvArgs& p, f32x16 (&acc)[MR][NR], int b, int mt0, int ncol0, int lane) {
  const int l31 = lane & 31, hi = lane >> 5;
  const float* maskb = p.mask ? p.mask + (long long)b * p.mask_bs : nullptr;
  const float* gaddb = p.gadd ? p.gadd + (long long)b * p.gadd_bs : nullptr;

  if (p.mode == EPI_PLAIN) {
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
      if (mt0 + mr >= p.mtiles) break;
      const int trow0 = (mt0 + mr) * 32;
      const bool sel = trow0 >= p.split_row;
      float* const oy = sel ? p.out[1].y : p.out[0].y;
      if (oy == nullptr) continue;
      const long long oy_bs = sel ? p.out[1].y_bs : p.out[0].y_bs;
      const int oy_ld = sel ? p.out[1].y_ld : p.out[0].y_ld;
      const float* const ores = sel ? p.out[1].res : p.out[0].res;
      const long long ores_bs = sel ? p.out[1].res_bs : p.out[0].res_bs;
      const int ores_ld = sel ? p.out[1].res_ld : p.out[0].res_ld;
      const unsigned fl = sel ? p.out[1].flags : p.out[0].flags;
      const float odiv = sel ? p.out[1].div : p.out[0].div;
      const int onrows = sel ? p.out[1].nrows : p.out[0].nrows;
      const int rbase = sel ? trow0 - p.split_row : trow0;
      const bool full_rows = rbase + 32 <= onrows;
      const bool simple = full_rows && gaddb == nullptr && (fl & ~(unsigned)(F_RES | F_ACC | F_DIV)) == 0;
      if (simple) {
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
          const int col = ncol0 + nr * 32 + l31;
          if (col >= p.Ncols) continue;
          float* ybase = oy + (long long)b * oy_bs + (long long)(rbase + 4 * hi) * oy_ld + col;
          float rv[16], yo[16];
          if (fl & F_RES) {
            const float* rp = ores + (long long)b * ores_bs + (long long)(rbase + 4 * hi) * ores_ld + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[r] = rp[(long long)((r & 3) + 8 * (r >> 2)) * ores_ld];
          }
          if (fl & F_ACC) {
#pragma unroll
            for (int r = 0; r < 16; ++r) yo[r] = ybase[(long long)((r & 3) + 8 * (r >> 2)) * oy_ld];
          }
          float vo[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[mr][nr][r];
            if (fl & F_RES) v = v + rv[r];
            if (fl & F_ACC) v = yo[r] + v;
            vo[r] = v;
          }
          // one uniform branch around all 16 divisions: inside the loop the compiler turns `if (flag) v /= d` into
          // an unconditional IEEE division sequence (~12 vector instructions per value) plus a select
          if (fl & F_DIV) {
            asm volatile("" ::: "memory");     // not speculatable: keeps the branch
#pragma unroll
            for (int r = 0; r < 16; ++r) vo[r] = vo[r] / odiv;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) ybase[(long long)((r & 3) + 8 * (r >> 2)) * oy_ld] = vo[r];
        }
        continue;
      }
#pragma unroll
      for (int nr = 0; nr < NR; ++nr) {
        const int col = ncol0 + nr * 32 + l31;
        if (col >= p.Ncols) continue;
        const float mk = maskb ? maskb[col] : 1.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lr = (r & 3) + 8 * (r >> 2) + 4 * hi;
          const int rr = rbase + lr;
          if (rr >= onrows) continue;
          float v = acc[mr][nr][r];
          if (gaddb) v += gaddb[(long long)rr * p.gadd_ld + (long long)col * p.gadd_ts];
          float* yp = oy + (long long)b * oy_bs + (long long)rr * oy_ld + col;
          if (fl & (F_RES | F_CPL_REV | F_CPL_FWD)) {
            const float rv = ores[(long long)b * ores_bs + (long long)rr * ores_ld + col];
            if (fl & F_RES) v = v + rv;
            else if (fl & F_CPL_REV) v = (rv - v * mk) * mk;
            else v = v * mk + rv * mk;
          }
          if (fl & F_ACC) v = *yp + v;
          if (fl & F_DIV) v = v / odiv;
          if (fl & F_OUTMASK) v *= mk;
          *yp = v;
        }
      }
    }
  } else if (p.mode == EPI_UPS) {
    const EpiOut& o = p.out[0];
    const int s = p.ups_s;
    float* yb = o.y + (long long)b * o.y_bs;
    const bool vec = (o.flags & F_VECST_WS) != 0;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
      if (mt0 + mr >= p.mtiles) break;
#pragma unroll
      for (int nr = 0; nr < NR; ++nr) {
        const int col = ncol0 + nr * 32 + l31;
        if (col >= p.Ncols) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row0 = (mt0 + mr) * 32 + 8 * q + 4 * hi;
          if (vec && row0 + 3 < o.nrows) {
            const int oc = row0 >> 3;
            const int n = col * 8 + (row0 & 7) - p.ups_pad;
            float4 v = make_float4(acc[mr][nr][4 * q], acc[mr][nr][4 * q + 1], acc[mr][nr][4 * q + 2], acc[mr][nr][4 * q + 3]);
            float* yp = yb + (long long)oc * o.y_ld + n;
            if (n >= 0 && n + 3 < p.Lout) {
              *reinterpret_cast<float4*>(yp) = v;
            } else {
              if (n >= 0 && n < p.Lout) yp[0] = v.x;
              if (n + 1 >= 0 && n + 1 < p.Lout) yp[1] = v.y;
              if (n + 2 >= 0 && n + 2 < p.Lout) yp[2] = v.z;
              if (n + 3 >= 0 && n + 3 < p.Lout) yp[3] = v.w;
            }
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int row = row0 + i;
              const int oc = row / s;
              const int n = col * s + (row - oc * s) - p.ups_pad;
              if (row < o.nrows && n >= 0 && n < p.Lout) yb[(long long)oc * o.y_ld + n] = acc[mr][nr][4 * q + i];
            }
          }
        }
      }
    }
  } else {
    if constexpr (MR % 2 == 0) {
      const int H = p.half_rows;
      const EpiOut& o = p.out[0];
      float lsum = 0.0f;
#pragma unroll
      for (int mr = 0; mr < MR; mr += 2) {
        if (mt0 + mr >= p.mtiles) break;
        const int pi = (mt0 + mr) >> 1;
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
          const int col = ncol0 + nr * 32 + l31;
          if (col >= p.Ncols) continue;
          const float mk = maskb ? maskb[col] : 1.0f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int chn = pi * 32 + rr;
            if (chn >= H) continue;
            float vA = acc[mr][nr][r];
            float vB = acc[mr + 1][nr][r];
            const long long yo = (long long)b * o.y_bs + (long long)chn * o.y_ld + col;
            if (p.mode == EPI_GATE) {
              if (gaddb) {
                vA += gaddb[(long long)chn * p.gadd_ld + (long long)col * p.gadd_ts];
                vB += gaddb[(long long)(H + chn) * p.gadd_ld + (long long)col * p.gadd_ts];
              }
              o.y[yo] = tanhf(vA) * ws_sigmoid(vB);
            } else if (p.mode == EPI_PROJ) {
              const float m = vA * mk, lg = vB * mk;
              const float e = p.eps ? p.eps[(long long)b * p.eps_bs + (long long)chn * p.eps_ld + col] : 0.0f;
              if (o.y) o.y[yo] = m;
              if (p.y2) p.y2[yo] = lg;
              if (p.y3) p.y3[yo] = (m + e * expf(lg) * p.noise_scale) * ((o.flags & F_OUTMASK) ? mk : 1.0f);   // F_OUTMASK: PosteriorEncoder's z
            } else {
              const float m = vA * mk, lg = vB * mk;
              const float x1 = o.res[(long long)b * o.res_bs + (long long)chn * o.res_ld + col];
              if (p.mode == EPI_CPL_FULL_REV) {
                o.y[yo] = (x1 - m) * expf(-lg) * mk;
              } else {
                o.y[yo] = m + x1 * expf(lg) * mk;
                lsum += lg;
              }
            }
          }
        }
      }
      if (p.mode == EPI_CPL_FULL_FWD && p.logdet) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lsum += __shfl_xor(lsum, off);
        if (lane == 0) atomicAdd(p.logdet + b, lsum);
      }
    }
  }
}

template <int WM, int WN, int MR, int NR>
__global__ void __launch_bounds__(512, 2) conv_ws_kernel(const ConvArgs p) {
  static_assert(WM * WN == 4, "four consumer waves");
  constexpr int BN = WN * NR * 32;
  constexpr int RING = 4;                                         // activation tiles in LDS (3 chunks of lookahead)
  constexpr int XREG = (KC * ((BN + WS_HALO_MAX) / 4) + 63) / 64;  // float4 per producer lane for one whole chunk
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int total_tiles = p.ntn * p.B;
  if ((int)blockIdx.x >= total_tiles) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xbuf_floats = KC * p.row_len;
  float* const XS = lds;                                         // [RING][KC][row_len]
  const int mblk = blockIdx.y;
  const int ntl = (total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // tiles of this workgroup
  const int total_cc = ntl * p.nchunks;                          // chunks this workgroup walks through

  if (wave >= 4) {
    // ======================================================================= activation producers
    // Wave 4+w owns the chunks cc == w (mod 4): it requests chunk cc at the start of chunk cc-3 (as soon as ring
    // slot cc%4 has been released), keeps it in registers across two barriers and publishes it during chunk cc-1.
    const int pw = wave - 4;
    const int R4 = p.row_len >> 2;
    const int xtotal = KC * R4;
    const int wc0 = lane / R4, wg0 = lane - wc0 * R4;
    const int xdc = 64 / R4, xdg = 64 - xdc * R4;
    const float slope = p.pre_slope;
    const bool act = slope != 1.0f;

    auto chunk_coords = [&](int cc, int& b, int& n0, int& ch) {
      const int ti = cc / p.nchunks;
      ch = cc - ti * p.nchunks;
      const int tile = blockIdx.x + ti * gridDim.x;
      b = tile / p.ntn;
      n0 = (tile - b * p.ntn) * BN;
    };
    // Branch-free request: clamp (channel, time) to a valid address and always issue the 16-byte load, so the XREG
    // loads of a chunk go out back-to-back; the zero padding is applied when the tile is written to LDS.
    auto x_issue = [&](int b, int n0, int ch, float4(&xv)[XREG]) {
      const float* xb = p.x + (long long)b * p.x_bs;
      const int xs_start = n0 + p.xoff0;
      const int c0 = ch * KC;
      int wc = wc0, wg = wg0;
      if (p.vec4) {
#pragma unroll
        for (int u = 0; u < XREG; ++u) {
          const int gc = min(c0 + wc, p.Cin - 1);
          int t = xs_start + 4 * wg;
          t = (t >= 0 && t < p.Lin) ? t : 0;
          xv[u] = *reinterpret_cast<const float4*>(xb + (long long)gc * p.x_ld + t);
          wc += xdc; wg += xdg;
          if (wg >= R4) { wg -= R4; ++wc; }
        }
      } else {
#pragma unroll
        for (int u = 0; u < XREG; ++u) {
          const int gc = c0 + wc;
          const int t = xs_start + 4 * wg;
          xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (lane + u * 64 < xtotal && gc < p.Cin && t + 3 >= 0 && t < p.Lin) {
            const float* row = xb + (long long)gc * p.x_ld;
            if (t >= 0 && t < p.Lin) xv[u].x = row[t];
            if (t + 1 >= 0 && t + 1 < p.Lin) xv[u].y = row[t + 1];
            if (t + 2 >= 0 && t + 2 < p.Lin) xv[u].z = row[t + 2];
            if (t + 3 >= 0 && t + 3 < p.Lin) xv[u].w = row[t + 3];
          }
          wc += xdc; wg += xdg;
          if (wg >= R4) { wg -= R4; ++wc; }
        }
      }
    };
    auto x_write = [&](float* buf, int b, int n0, int ch, const float4(&xv)[XREG]) {
      const float* mb = p.in_mask ? p.in_mask + (long long)b * p.in_mask_bs : nullptr;
      const int xs_start = n0 + p.xoff0;
      const int c0 = ch * KC;
      int wc = wc0, wg = wg0;
#pragma unroll
      for (int u = 0; u < XREG; ++u) {
        if (lane + u * 64 < xtotal) {
          const int t = xs_start + 4 * wg;
          const bool cok = c0 + wc < p.Cin;
          float4 q = xv[u];
          q.x = (cok && t >= 0 && t < p.Lin) ? q.x : 0.f;
          q.y = (cok && t + 1 >= 0 && t + 1 < p.Lin) ? q.y : 0.f;
          q.z = (cok && t + 2 >= 0 && t + 2 < p.Lin) ? q.z : 0.f;
          q.w = (cok && t + 3 >= 0 && t + 3 < p.Lin) ? q.w : 0.f;
          if (act) {
            q.x = q.x > 0.f ? q.x : q.x * slope;
            q.y = q.y > 0.f ? q.y : q.y * slope;
            q.z = q.z > 0.f ? q.z : q.z * slope;
            q.w = q.w > 0.f ? q.w : q.w * slope;
          }
          if (mb) {
            q.x *= (t >= 0 && t < p.Lin) ? mb[t] : 0.f;
            q.y *= (t + 1 >= 0 && t + 1 < p.Lin) ? mb[t + 1] : 0.f;
            q.z *= (t + 2 >= 0 && t + 2 < p.Lin) ? mb[t + 2] : 0.f;
            q.w *= (t + 3 >= 0 && t + 3 < p.Lin) ? mb[t + 3] : 0.f;
          }
          *reinterpret_cast<float4*>(buf + wc * p.row_len + 4 * wg) = q;
        }
        wc += xdc; wg += xdg;
        if (wg >= R4) { wg -= R4; ++wc; }
      }
    };

    float4 xv[XREG];
    // prologue: every producer wave loads and publishes its first chunk (cc = pw), chunks 0..3 fill the ring
    int mine = pw;                       // the chunk currently held / next to publish by this wave
    if (mine < total_cc) {
      int b, n0, ch;
      chunk_coords(mine, b, n0, ch);
      x_issue(b, n0, ch, xv);
      x_write(XS + (mine % RING) * xbuf_floats, b, n0, ch, xv);
    }
    mine += 4;
    bool holding = false;
    long long t_iss = 0, t_pub = 0, t_pbar = 0;
    ws_publish_barrier();
    for (int cc = 0; cc < total_cc; ++cc) {
      // during chunk cc the consumers read slot cc%4; slots of chunks cc+1..cc+3 are free to fill
      if (mine < total_cc) {
        int b, n0, ch;
        if (!holding && mine - 3 <= cc) {          // slot (mine%4) was released by the barrier that ended chunk mine-4
          const long long ta = __builtin_readcyclecounter();
          chunk_coords(mine, b, n0, ch);
          x_issue(b, n0, ch, xv);
          holding = true;
          t_iss += __builtin_readcyclecounter() - ta;
        }
        if (holding && mine - 1 <= cc) {           // publish one chunk before it is consumed
          chunk_coords(mine, b, n0, ch);
          const long long tb = __builtin_readcyclecounter();
          x_write(XS + (mine % RING) * xbuf_floats, b, n0, ch, xv);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          t_pub += __builtin_readcyclecounter() - tb;
          holding = false;
          mine += 4;
        }
      }
      const long long tc = __builtin_readcyclecounter();
      ws_publish_barrier();
      t_pbar += __builtin_readcyclecounter() - tc;
    }
    if (p.dbg && tid == 256) {
      long long* d = p.dbg + 4 * (blockIdx.x + (long long)gridDim.x * blockIdx.y) + 4 * (long long)gridDim.x * gridDim.y;
      d[0] = 0; d[1] = t_iss; d[2] = t_iss + t_pub; d[3] = t_iss + t_pub + t_pbar;
    }
    return;
  }

  // ========================================================================= consumers
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, hi = lane >> 5;
  const int mt0 = (mblk * WM + wm) * MR;
  const bool m_ok = mt0 < p.mtiles;
  const int bcol = wn * NR * 32 + l31 - p.pad - p.xoff0;

  float biasr[MR][16];
#pragma unroll
  for (int mr = 0; mr < MR; ++mr)
#pragma unroll
    for (int i = 0; i < 16; ++i) biasr[mr][i] = p.bias[min(mt0 + mr, p.mtiles - 1) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi];
  f32x16 acc[MR][NR];
#pragma unroll
  for (int mr = 0; mr < MR; ++mr)
#pragma unroll
    for (int nr = 0; nr < NR; ++nr)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mr][nr][i] = biasr[mr][i];

  // weight fragment stream straight from L2 (packed order), one group ahead; wraps to group 0 at the tile end
  const float4* wp4 = reinterpret_cast<const float4*>(p.wp);
  long long abase[MR];
  float4 a0[MR], a1[MR];                 // ping-pong weight fragments (a0 = group to run next)
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) {
    abase[mr] = (long long)min(mt0 + mr, p.mtiles - 1) * p.ksg_total * 64 + lane;
    a0[mr] = wp4[abase[mr]];
  }
  const int ngroups = p.ktaps * (KC / 8);

  ws_consume_barrier();   // prologue barrier (matches the producers')

  int tile = blockIdx.x;
  int cc = 0;
  long long t_bar = 0, t_mma = 0, t_epi = 0;
  const long long t_begin = __builtin_readcyclecounter();
  while (true) {
    const int b = tile / p.ntn;
    const int n0 = (tile - b * p.ntn) * BN;
    const int ncol0 = n0 + wn * NR * 32;
    const bool wave_active = m_ok && ncol0 < p.Ncols;
    int ksg = 0;
    for (int ch = 0; ch < p.nchunks; ++ch) {
      const long long t0 = __builtin_readcyclecounter();
      if (wave_active) {
        const float* xs = XS + (cc % RING) * xbuf_floats;
        const float* bp = xs + hi * p.row_len + bcol;
        float b0[4][NR], b1[4][NR];             // ping-pong activation fragments
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int nr = 0; nr < NR; ++nr) b0[s][nr] = bp[(2 * s) * p.row_len + nr * 32];
        int g = 0;
        // One group = 4 k-steps = 4*MR*NR MFMAs.  The next group's fragment requests are issued after the first
        // k-step's MFMAs so that they overlap MFMA execution (a lone wave per SIMD has nothing else to fill the
        // pipe); two register sets alternate, so no moves.  ngroups is a multiple of 4.
        auto run_group = [&](float4(&ac)[MR], float(&bc)[4][NR], float4(&an)[MR], float(&bn)[4][NR], bool last_group) {
#pragma unroll
          for (int mr = 0; mr < MR; ++mr) {
            const float av = ac[mr].x;
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bc[0][nr], acc[mr][nr], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          ++ksg;
          const int kn = ksg < p.ksg_total ? ksg : 0;      // wraps to group 0 = first group of the next tile
#pragma unroll
          for (int mr = 0; mr < MR; ++mr) an[mr] = wp4[abase[mr] + (long long)kn * 64];
          const float* bpn = (g == KC / 8 - 1) ? bp + p.dil - (KC - 8) * p.row_len : bp + 8 * p.row_len;
          if (!last_group) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
              for (int nr = 0; nr < NR; ++nr) bn[s][nr] = bpn[(2 * s) * p.row_len + nr * 32];
          }
          bp = bpn;
          g = (g + 1) & (KC / 8 - 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int s = 1; s < 4; ++s) {
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) {
              const float av = ws_pick4(ac[mr], s);
#pragma unroll
              for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bc[s][nr], acc[mr][nr], 0, 0, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        };
        for (int gi = 0; gi < ngroups; gi += 2) {
          run_group(a0, b0, a1, b1, false);
          run_group(a1, b1, a0, b0, gi + 2 >= ngroups);
        }
      }
      const long long t1 = __builtin_readcyclecounter();
      ws_consume_barrier();
      const long long t2 = __builtin_readcyclecounter();
      t_mma += t1 - t0; t_bar += t2 - t1;
      ++cc;
    }
    const long long t3 = __builtin_readcyclecounter();
    if (wave_active) {
      int opq = 0;
      asm volatile("" : "+s"(opq));      // keep tile-invariant epilogue address math out of the persistent loop
      ws_epilogue<MR, NR>(p, acc, b, mt0 + opq, ncol0, lane);
#pragma unroll
      for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[mr][nr][i] = biasr[mr][i];
    }
    t_epi += __builtin_readcyclecounter() - t3;
    tile += gridDim.x;
    if (tile >= total_tiles) break;
  }
  if (p.dbg && tid == 0) {
    long long* d = p.dbg + 4 * (blockIdx.x + (long long)gridDim.x * blockIdx.y);
    d[0] = 0; d[1] = t_bar; d[2] = t_bar + t_mma; d[3] = t_bar + t_mma + t_epi;   // decoded as (barrier wait, MFMA, epilogue)
    (void)t_begin;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Two consumer sets per workgroup ("WS2").
//
// Identical co-resident workgroups drift into lockstep, so the staging and epilogue phases of the
// plain kernel are never covered by another workgroup's MFMAs.  Here the coordination is explicit: one 12-wave
// workgroup per CU, three waves per SIMD:
//   waves 0-3  consumer set A : even tiles of the workgroup's tile list
//   waves 4-7  consumer set B : odd tiles, running nchunks/2 channel chunks behind set A
//   waves 8-9  activation producers for set A, waves 10-11 for set B (each pair splits every chunk; a chunk is
//              requested two steps ahead, held in registers across a barrier and published one step ahead)
// Every step (one s_barrier) each set multiplies one 32-channel chunk.  When a set finishes a tile its epilogue
// runs at the start of the next step, while the other set is in the middle of its own tile: the MFMA pipe of each
// SIMD always has one wave with work.  Bias lives in LDS so that a consumer fits 168 registers.
// lane: 0..127 (the two producer waves of a set split every chunk between them)
template <int XREG>
__device__ __forceinline__ void ws2_x_issue(const ConvArgs& p, int b, int n0, int ch, int lane, float4 (&xv)[XREG]) {
  const int R4 = p.row_len >> 2;
  const int wc0 = lane / R4, wg0 = lane - wc0 * R4;
  const int xdc = 128 / R4, xdg = 128 - xdc * R4;
  const float* xb = p.x + (long long)b * p.x_bs;
  const int xs_start = n0 + p.xoff0;
  const int c0 = ch * KC;
  int wc = wc0, wg = wg0;
  if (p.vec4) {
#pragma unroll
    for (int u = 0; u < XREG; ++u) {
      const int gc = min(c0 + wc, p.Cin - 1);
      int t = xs_start + 4 * wg;
      t = (t >= 0 && t < p.Lin) ? t : 0;              // t, x_ld are multiples of 4: the float4 stays inside the row
      xv[u] = *reinterpret_cast<const float4*>(xb + (long long)gc * p.x_ld + t);
      wc += xdc; wg += xdg;
      if (wg >= R4) { wg -= R4; ++wc; }
    }
  } else {
    const int xtotal = KC * R4;
#pragma unroll
    for (int u = 0; u < XREG; ++u) {
      const int gc = c0 + wc;
      const int t = xs_start + 4 * wg;
      xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lane + u * 128 < xtotal && gc < p.Cin && t + 3 >= 0 && t < p.Lin) {
        const float* row = xb + (long long)gc * p.x_ld;
        if (t >= 0 && t < p.Lin) xv[u].x = row[t];
        if (t + 1 >= 0 && t + 1 < p.Lin) xv[u].y = row[t + 1];
        if (t + 2 >= 0 && t + 2 < p.Lin) xv[u].z = row[t + 2];
        if (t + 3 >= 0 && t + 3 < p.Lin) xv[u].w = row[t + 3];
      }
      wc += xdc; wg += xdg;
      if (wg >= R4) { wg -= R4; ++wc; }
    }
  }
}

template <int XREG>
__device__ __forceinline__ void ws2_x_write(const ConvArgs& p, float* buf, int b, int n0, int ch, int lane, const float4 (&xv)[XREG]) {
  const int R4 = p.row_len >> 2;
  const int xtotal = KC * R4;
  const int wc0 = lane / R4, wg0 = lane - wc0 * R4;
  const int xdc = 128 / R4, xdg = 128 - xdc * R4;
  const float* mb = p.in_mask ? p.in_mask + (long long)b * p.in_mask_bs : nullptr;
  const int xs_start = n0 + p.xoff0;
  const int c0 = ch * KC;
  const float slope = p.pre_slope;
  const bool act = slope != 1.0f;
  int wc = wc0, wg = wg0;
#pragma unroll
  for (int u = 0; u < XREG; ++u) {
    if (lane + u * 128 < xtotal) {
      const int t = xs_start + 4 * wg;
      const bool cok = c0 + wc < p.Cin;
      float4 q = xv[u];
      q.x = (cok && t >= 0 && t < p.Lin) ? q.x : 0.f;
      q.y = (cok && t + 1 >= 0 && t + 1 < p.Lin) ? q.y : 0.f;
      q.z = (cok && t + 2 >= 0 && t + 2 < p.Lin) ? q.z : 0.f;
      q.w = (cok && t + 3 >= 0 && t + 3 < p.Lin) ? q.w : 0.f;
      if (act) {
        q.x = q.x > 0.f ? q.x : q.x * slope;
        q.y = q.y > 0.f ? q.y : q.y * slope;
        q.z = q.z > 0.f ? q.z : q.z * slope;
        q.w = q.w > 0.f ? q.w : q.w * slope;
      }
      if (mb) {
        q.x *= (t >= 0 && t < p.Lin) ? mb[t] : 0.f;
        q.y *= (t + 1 >= 0 && t + 1 < p.Lin) ? mb[t + 1] : 0.f;
        q.z *= (t + 2 >= 0 && t + 2 < p.Lin) ? mb[t + 2] : 0.f;
        q.w *= (t + 3 >= 0 && t + 3 < p.Lin) ? mb[t + 3] : 0.f;
      }
      *reinterpret_cast<float4*>(buf + wc * p.row_len + 4 * wg) = q;
    }
    wc += xdc; wg += xdg;
    if (wg >= R4) { wg -= R4; ++wc; }
  }
}

template <int WM, int WN, int MR, int NR>
__global__ void __launch_bounds__(768, 3) conv_ws2_kernel(const ConvArgs p) {
  static_assert(WM * WN == 4, "four consumer waves per set");
  constexpr int BN = WN * NR * 32;
  constexpr int MTB = WM * MR;
  constexpr int XREG = (KC * ((BN + WS_HALO_MAX) / 4) + 127) / 128;   // float4 per producer lane for half a chunk
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int total_tiles = p.ntn * p.B;
  if ((int)blockIdx.x >= total_tiles) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xbuf_floats = KC * p.row_len;
  float* const XS = lds;                                  // [2 sets][2][KC][row_len]
  float* const BS = lds + 4 * xbuf_floats;                // [MTB*32] bias of this row block
  const int mblk = blockIdx.y;
  const int gx = gridDim.x;
  const int nch = p.nchunks;
  const int n_all = (total_tiles - (int)blockIdx.x + gx - 1) / gx;   // tiles of this workgroup
  const int off = nch >> 1;                                          // set B runs half a tile behind set A
  const int LA = ((n_all + 1) >> 1) * nch, LB = (n_all >> 1) * nch;  // chunks per set
  const int S = max(LA, LB > 0 ? LB + off : 0);                      // steps

  if (tid < MTB * 32) BS[tid] = p.bias[min(mblk * MTB * 32 + tid, p.mtiles * 32 - 1)];

  if (wave >= 8) {
    // ======================================================================= activation producers
    // The two waves of a set split every chunk (128 lanes).  Chunk i is requested during step i-2 (right after
    // chunk i-1 has been written), held in registers across one barrier and published during step i-1.
    const int ps = (wave - 8) >> 1;
    const int pl = (wave & 1) * 64 + lane;
    const int L = ps ? LB : LA, offp = ps ? off : 0;
    float* const XSp = XS + ps * 2 * xbuf_floats;
    auto coords = [&](int i, int& b, int& n0, int& ch) {
      const int ti = i / nch;
      ch = i - ti * nch;
      const int tile = blockIdx.x + (2 * ti + ps) * gx;
      b = tile / p.ntn;
      n0 = (tile - b * p.ntn) * BN;
    };
    float4 xv[XREG];
    int b, n0, ch;
    if (L > 0) { coords(0, b, n0, ch); ws2_x_issue<XREG>(p, b, n0, ch, pl, xv); ws2_x_write<XREG>(p, XSp, b, n0, ch, pl, xv); }
    if (L > 1) { coords(1, b, n0, ch); ws2_x_issue<XREG>(p, b, n0, ch, pl, xv); }
    ws_publish_barrier();
    long long t_work = 0, t_pbar = 0;
    for (int step = 0; step < S; ++step) {
      const long long ta = __builtin_readcyclecounter();
      const int i = step - offp + 1;          // the chunk consumed in the next step
      if (i >= 1 && i < L) {
        coords(i, b, n0, ch);
        ws2_x_write<XREG>(p, XSp + (i & 1) * xbuf_floats, b, n0, ch, pl, xv);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (i + 1 < L) { coords(i + 1, b, n0, ch); ws2_x_issue<XREG>(p, b, n0, ch, pl, xv); }
      }
      const long long tb = __builtin_readcyclecounter();
      ws_publish_barrier();
      t_work += tb - ta; t_pbar += __builtin_readcyclecounter() - tb;
    }
    if (p.dbg && tid == 512) {
      long long* d = p.dbg + 4 * (blockIdx.x + (long long)gx * blockIdx.y) + 8 * (long long)gx * gridDim.y;
      d[0] = 0; d[1] = t_work; d[2] = t_work; d[3] = t_work + t_pbar;       // decoded as (write+issue, 0, barrier wait)
    }
    return;
  }

  // ========================================================================= consumers
  const int cs = wave >> 2, cw = wave & 3;
  const int L = cs ? LB : LA, offc = cs ? off : 0;
  const float* const XSc = XS + cs * 2 * xbuf_floats;
  const int wm = cw / WN, wn = cw % WN;
  const int l31 = lane & 31, hi = lane >> 5;
  const int mt0 = (mblk * WM + wm) * MR;
  const bool m_ok = mt0 < p.mtiles;
  const int bcol = wn * NR * 32 + l31 - p.pad - p.xoff0;
  const float* const bsw = BS + wm * MR * 32 + 4 * hi;

  f32x16 acc[MR][NR];
  const float4* wp4 = reinterpret_cast<const float4*>(p.wp);
  long long abase[MR];
  float4 a0[MR], a1[MR];
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) {
    abase[mr] = (long long)min(mt0 + mr, p.mtiles - 1) * p.ksg_total * 64 + lane;
    a0[mr] = wp4[abase[mr]];
  }
  const int ngroups = p.ktaps * (KC / 8);

  ws_consume_barrier();   // prologue barrier: bias and the first chunks are in LDS

#define SVOC_WS2_ACC_INIT()                                           \
  _Pragma("unroll") for (int mr = 0; mr < MR; ++mr)                   \
  _Pragma("unroll") for (int i = 0; i < 16; ++i) {                    \
    const float bv = bsw[mr * 32 + (i & 3) + 8 * (i >> 2)];           \
    _Pragma("unroll") for (int nr = 0; nr < NR; ++nr) acc[mr][nr][i] = bv; \
  }
  SVOC_WS2_ACC_INIT();

  int tile = blockIdx.x + cs * gx;
  int ch = 0, ksg = 0;
  int cb = 0, cncol0 = 0;             // coordinates of the tile being accumulated
  bool cactive = false, pend = false;
  long long t_bar = 0, t_mma = 0, t_epi = 0;
  for (int step = 0; step < S; ++step) {
    const int i = step - offc;
    const long long t0 = __builtin_readcyclecounter();
    long long t1 = t0;
    if (i >= 0 && i < L) {
      if (pend) {                      // epilogue of the previous tile, covered by the other set's MFMAs
        if (cactive) {
          int opq = 0;
          asm volatile("" : "+s"(opq));
          ws_epilogue<MR, NR>(p, acc, cb, mt0 + opq, cncol0, lane);
          SVOC_WS2_ACC_INIT();
        }
        pend = false;
      }
      t1 = __builtin_readcyclecounter();
      if (ch == 0) {
        cb = tile / p.ntn;
        cncol0 = (tile - cb * p.ntn) * BN + wn * NR * 32;
        cactive = m_ok && cncol0 < p.Ncols;
        ksg = 0;
      }
      if (cactive) {
        const float* xs = XSc + (i & 1) * xbuf_floats;
        const float* bp = xs + hi * p.row_len + bcol;
        float b0[4][NR], b1[4][NR];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int nr = 0; nr < NR; ++nr) b0[s][nr] = bp[(2 * s) * p.row_len + nr * 32];
        int g = 0;
        auto run_group = [&](float4(&ac)[MR], float(&bc)[4][NR], float4(&an)[MR], float(&bn)[4][NR], bool last_group) {
#pragma unroll
          for (int mr = 0; mr < MR; ++mr) {
            const float av = ac[mr].x;
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bc[0][nr], acc[mr][nr], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          ++ksg;
          const int kn = ksg < p.ksg_total ? ksg : 0;
#pragma unroll
          for (int mr = 0; mr < MR; ++mr) an[mr] = wp4[abase[mr] + (long long)kn * 64];
          const float* bpn = (g == KC / 8 - 1) ? bp + p.dil - (KC - 8) * p.row_len : bp + 8 * p.row_len;
          if (!last_group) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
              for (int nr = 0; nr < NR; ++nr) bn[s][nr] = bpn[(2 * s) * p.row_len + nr * 32];
          }
          bp = bpn;
          g = (g + 1) & (KC / 8 - 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int s = 1; s < 4; ++s) {
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) {
              const float av = ws_pick4(ac[mr], s);
#pragma unroll
              for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bc[s][nr], acc[mr][nr], 0, 0, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        };
        for (int gi = 0; gi < ngroups; gi += 2) {
          run_group(a0, b0, a1, b1, false);
          run_group(a1, b1, a0, b0, gi + 2 >= ngroups);
        }
      }
      if (++ch == nch) { ch = 0; pend = true; tile += 2 * gx; }
    }
    const long long t2 = __builtin_readcyclecounter();
    ws_consume_barrier();
    t_epi += t1 - t0; t_mma += t2 - t1; t_bar += __builtin_readcyclecounter() - t2;
  }
  if (pend && cactive) ws_epilogue<MR, NR>(p, acc, cb, mt0, cncol0, lane);
  if (p.dbg && (tid == 0 || tid == 256)) {
    long long* d = p.dbg + 4 * (blockIdx.x + (long long)gx * blockIdx.y) + (tid ? 4 : 0) * (long long)gx * gridDim.y;
    d[0] = 0; d[1] = t_bar; d[2] = t_bar + t_mma; d[3] = t_bar + t_mma + t_epi;   // decoded as (barrier wait, MFMA, epilogue)
  }
#undef SVOC_WS2_ACC_INIT
}


namespace {
int ws_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <int WM, int WN, int MR, int NR>
int launch_ws_cfg(ConvArgs& a, int B, hipStream_t st) {
  constexpr int BN = WN * NR * 32;
  constexpr int MTB = WM * MR;
  auto kern = conv_ws_kernel<WM, WN, MR, NR>;
  static bool attr_set = false;
  if (!attr_set) {
    SVOC_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const size_t lds = 4 * (size_t)KC * a.row_len * sizeof(float);
  if (a.row_len > BN + WS_HALO_MAX || lds > 160 * 1024) return 1;   // caller falls back to the non-persistent kernel
  a.ntn = (a.Ncols + BN - 1) / BN;
  a.B = B;
  const int gy = (a.mtiles + MTB - 1) / MTB;
  const long long total = (long long)a.ntn * B;
  const long long slots = std::max<long long>(1, ws_num_cus() / gy);
  const long long tpb = (total + slots - 1) / slots;
  const int gx = (int)((total + tpb - 1) / tpb);
  hipLaunchKernelGGL(kern, dim3(gx, gy, 1), dim3(512), lds, st, a);
  SVOC_HIP(hipGetLastError());
  return SVOC_OK;
}
template <int WM, int WN, int MR, int NR>
int launch_ws2_cfg(ConvArgs& a, int B, hipStream_t st) {
  constexpr int BN = WN * NR * 32;
  constexpr int MTB = WM * MR;
  auto kern = conv_ws2_kernel<WM, WN, MR, NR>;
  static bool attr_set = false;
  if (!attr_set) {
    SVOC_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const size_t lds = (4 * (size_t)KC * a.row_len + MTB * 32) * sizeof(float);
  if (a.row_len > BN + WS_HALO_MAX || lds > 160 * 1024 || a.nchunks < 2) return 1;
  const int ntn = (a.Ncols + BN - 1) / BN;
  const int gy = (a.mtiles + MTB - 1) / MTB;
  const long long total = (long long)ntn * B;
  const long long slots = std::max<long long>(1, ws_num_cus() / gy);
  if (total < 4 * slots) return 1;                       // too few tiles for a persistent pipeline to pay
  long long tpb = (total + slots - 1) / slots;
  tpb += tpb & 1;                                         // both consumer sets get the same number of tiles
  const int gx = (int)((total + tpb - 1) / tpb);
  a.ntn = ntn;
  a.B = B;
  hipLaunchKernelGGL(kern, dim3(gx, gy, 1), dim3(768), lds, st, a);
  SVOC_HIP(hipGetLastError());
  return SVOC_OK;
}
}  // namespace

// returns SVOC_OK, a negative error, or 1 when this launch is not eligible
int launch_conv_ws(ConvArgs& a, int B, int WM, int WN, int MR, int NR, hipStream_t st) {
#define SVOC_WS(wm, wn, mr, nr) if (WM == wm && WN == wn && MR == mr && NR == nr) return launch_ws_cfg<wm, wn, mr, nr>(a, B, st)
  SVOC_WS(2, 2, 2, 2);
  SVOC_WS(2, 2, 2, 1);
  SVOC_WS(2, 2, 1, 1);
  SVOC_WS(1, 4, 1, 1);
#undef SVOC_WS
  return 1;
}

int launch_conv_ws2(ConvArgs& a, int B, int WM, int WN, int MR, int NR, hipStream_t st) {
  if (WM == 2 && WN == 2 && MR == 2 && NR == 2) return launch_ws2_cfg<2, 2, 2, 2>(a, B, st);
  return 1;
}

}  // namespace svoc
