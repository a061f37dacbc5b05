// Implicit-GEMM 1-D convolution on the gfx950 FP32 matrix cores.
//
// One kernel family covers every convolution of the SMART-Vocoder inference
// path (reference models.py:32-33,120-135; modules.py:127-146,190-207,318-320):
//   D[M = out-channel rows][N = time] = sum_{tap j, in-channel c} W[row][c][j] * act(x[c][n + j*dil - pad])
// computed with v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).
//
// Mapping to the hardware
//   * M = output channels, N = time: the MFMA C/D layout puts 32 consecutive
//     time steps of one channel in the 32 lanes of a half-wave, so every
//     accumulator register stores as 128-byte contiguous NCW segments.
//   * B operand (activations): a [KC=32 channels][BN + dilation halo] tile is
//     staged ONCE per input-channel chunk into LDS with 16-byte coalesced
//     loads; the leaky-relu that precedes the conv in the reference
//     (modules.py:212,216; models.py:147) and the optional x*x_mask are applied
//     while staging, and out-of-range positions are written as zeros (the
//     conv's own zero padding, commons.py:14-15).  All k taps re-read the same
//     tile at shifted columns (ds_read_b32, 32 consecutive lanes = 32
//     consecutive banks, conflict-free).
//   * A operand (weights): repacked at load time into MFMA fragment order
//     (pack.hip) so one global_load_dwordx4 per lane feeds four k-steps; the
//     stream is shared by every time tile of every utterance, stays L2
//     resident, and is prefetched one group ahead in registers.
//   * Each wave owns MR x NR 32x32 accumulator tiles; fused epilogues (bias,
//     residual, MRF accumulate, masks, WN gate, reparameterisation, coupling,
//     polyphase ConvTranspose scatter) run from the accumulators.
#include "svoc_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace svoc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr unsigned F_VECST = 1u << 16;   // internal: float4 stores legal for EPI_UPS


__device__ __forceinline__ float pick4(const float4& v, int s) {
  return s == 0 ? v.x : (s == 1 ? v.y : (s == 2 ? v.z : v.w));
}

// One output tile: workgroup (bx, by, bz) = (time tile, row block, batch element) of problem p.
// KS (K split, short inputs): the four waves of the workgroup share ONE (32*MR) x (32*NR) tile and each multiplies over a
// quarter of K - wave w takes the 8-channel group w of every (32-channel chunk, tap) - then the partial accumulators
// are summed through LDS and wave 0 runs the epilogue.  A tile's serial MFMA chain is 4x shorter and there are 4x more
// workgroups than with a 64x64 four-wave tile: at 1 x 200 frames the decoder's C=256 stage has 1600 columns, 100
// conventional workgroups each chaining 1408 MFMAs (117 us per convolution) against 400 workgroups chaining 352.
template <int WM, int WN, int MR, int NR, bool KS = false>
__device__ __forceinline__ void conv_tile(const ConvArgs& p, const int bx, const int by, const int bz, const long long dbg_lin) {
  constexpr int NT = WM * WN * 64;
  constexpr int BN = (KS ? 1 : WN) * NR * 32;
  extern __shared__ __attribute__((aligned(16))) float xs[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = KS ? 0 : wave / WN, wn = KS ? 0 : wave % WN;
  const int l31 = lane & 31, hi = lane >> 5;
  const int b = bz;
  const int n0 = bx * BN;
  const int mt0 = (by * (KS ? 1 : WM) + wm) * MR;
  const int ncol0 = n0 + wn * NR * 32;
  const bool wave_active = (mt0 < p.mtiles) && (ncol0 < p.Ncols);

  f32x16 acc[MR][NR];
#pragma unroll
  for (int mr = 0; mr < MR; ++mr)
#pragma unroll
    for (int nr = 0; nr < NR; ++nr)
#pragma unroll
      for (int i = 0; i < 16; ++i)   // bias folded into the accumulator (no dependent loads in the epilogue)
        acc[mr][nr][i] = (KS && wave != 0) ? 0.f : p.bias[min(mt0 + mr, p.mtiles - 1) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi];

  // ---- weight fragment stream (prefetched one group of 4 k-steps ahead)
  const float4* wp4 = reinterpret_cast<const float4*>(p.wp);
  long long abase[MR];
  float4 a_cur[MR], a_nxt[MR];
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) {
    const int mt = min(mt0 + mr, p.mtiles - 1);
    abase[mr] = (long long)mt * p.ksg_total * 64 + lane;
    a_nxt[mr] = wp4[abase[mr]];
  }
  int ksg = 0;
  const int ksg_last = p.ksg_total - 1;

  const int R4 = p.row_len >> 2;
  const int xs_start = n0 + p.xoff0;
  const float* xb = p.x + (long long)b * p.x_bs;
  const float* mb = p.in_mask ? p.in_mask + (long long)b * p.in_mask_bs : nullptr;
  const float slope = p.pre_slope;
  const bool act = slope != 1.0f;
  const float* bp0 = xs + hi * p.row_len + (wn * NR * 32 + l31 - p.pad - p.xoff0);
  constexpr int SU = (MR * NR >= 8) ? 6 : 9;   // float4 staging loads in flight per thread
  const int stage_total = p.kcs * R4;
  const int wc0 = tid / R4, wg0 = tid - wc0 * R4;
  const int stage_dc = NT / R4, stage_dg = NT - stage_dc * R4;

  long long tstamp[4] = {0, 0, 0, 0};
  auto dbg_clock = [&]() -> long long { return p.dbg_wall ? (long long)wall_clock64() : (long long)__builtin_readcyclecounter(); };
  if (p.dbg) tstamp[0] = dbg_clock();
  const int sub_per_stage = p.kcs / KC;
  for (int ch = 0; ch < p.nchunks; ch += sub_per_stage) {
    __syncthreads();
    // ---- stage [KC][row_len] activations, zero-filled outside [0,Lin) and beyond Cin.  SU float4 groups per
    // thread are requested back-to-back before any is consumed, so one memory latency covers SU loads.
    const int c0 = ch * KC;
    {
      int wc = wc0, wg = wg0;   // (channel, float4-group) walker for idx = tid + i*NT, no divisions
      for (int base = tid; base < stage_total; base += NT * SU) {
        float4 v[SU];
        const int wc_s = wc, wg_s = wg;
        if (p.vec4) {
          // branch-free: clamp to a valid address and always issue the 16-byte load, so the SU loads go out
          // back-to-back; the zero padding is applied below when the tile is written to LDS
#pragma unroll
          for (int u = 0; u < SU; ++u) {
            const int gc = min(c0 + wc, p.Cin - 1);
            int t = xs_start + 4 * wg;
            t = (t >= 0 && t < p.Lin) ? t : 0;
            v[u] = *reinterpret_cast<const float4*>(xb + (long long)gc * p.x_ld + t);
            wc += stage_dc;
            wg += stage_dg;
            if (wg >= R4) { wg -= R4; ++wc; }
          }
        } else {
#pragma unroll
          for (int u = 0; u < SU; ++u) {
            const int gc = c0 + wc;
            const int t = xs_start + 4 * wg;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (base + u * NT < stage_total && gc < p.Cin && t + 3 >= 0 && t < p.Lin) {
              const float* row = xb + (long long)gc * p.x_ld;
              if (t >= 0 && t < p.Lin) v[u].x = row[t];
              if (t + 1 >= 0 && t + 1 < p.Lin) v[u].y = row[t + 1];
              if (t + 2 >= 0 && t + 2 < p.Lin) v[u].z = row[t + 2];
              if (t + 3 >= 0 && t + 3 < p.Lin) v[u].w = row[t + 3];
            }
            wc += stage_dc;
            wg += stage_dg;
            if (wg >= R4) { wg -= R4; ++wc; }
          }
        }
        int wc2 = wc_s, wg2 = wg_s;   // second pass of the same walk: activation, mask, LDS write
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          if (base + u * NT < stage_total) {
            float4 q = v[u];
            {
              const int t = xs_start + 4 * wg2;
              const bool cok = c0 + wc2 < p.Cin;
              q.x = (cok && t >= 0 && t < p.Lin) ? q.x : 0.f;
              q.y = (cok && t + 1 >= 0 && t + 1 < p.Lin) ? q.y : 0.f;
              q.z = (cok && t + 2 >= 0 && t + 2 < p.Lin) ? q.z : 0.f;
              q.w = (cok && t + 3 >= 0 && t + 3 < p.Lin) ? q.w : 0.f;
            }
            if (act) {
              q.x = q.x > 0.f ? q.x : q.x * slope;
              q.y = q.y > 0.f ? q.y : q.y * slope;
              q.z = q.z > 0.f ? q.z : q.z * slope;
              q.w = q.w > 0.f ? q.w : q.w * slope;
            }
            if (mb) {
              const int t = xs_start + 4 * wg2;
              q.x *= (t >= 0 && t < p.Lin) ? mb[t] : 0.f;
              q.y *= (t + 1 >= 0 && t + 1 < p.Lin) ? mb[t + 1] : 0.f;
              q.z *= (t + 2 >= 0 && t + 2 < p.Lin) ? mb[t + 2] : 0.f;
              q.w *= (t + 3 >= 0 && t + 3 < p.Lin) ? mb[t + 3] : 0.f;
            }
            *reinterpret_cast<float4*>(xs + wc2 * p.row_len + 4 * wg2) = q;
          }
          wc2 += stage_dc;
          wg2 += stage_dg;
          if (wg2 >= R4) { wg2 -= R4; ++wc2; }
        }
      }
    }
    __syncthreads();
    if (p.dbg && ch == 0) tstamp[1] = dbg_clock();

    if constexpr (KS) {
      if (wave_active) {
        // this wave's groups: (sub-chunk q, tap j) -> packed group ((ch + q) * ktaps + j) * 4 + wave, LDS rows
        // q*32 + 8*wave + 2s + hi at column offset j*dil; one group ahead in registers
        const int nsub = min(sub_per_stage, p.nchunks - ch);
        const int ng = nsub * p.ktaps;
        const float* bpw = bp0 + 8 * wave * p.row_len;
        float4 an[MR];
        float bn[4][NR];
        auto req = [&](int q, int j) {
          const long long kg = ((long long)(ch + q) * p.ktaps + j) * 4 + wave;
#pragma unroll
          for (int mr = 0; mr < MR; ++mr) an[mr] = wp4[abase[mr] + kg * 64];
          const float* bq = bpw + q * KC * p.row_len + j * p.dil;
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) bn[s][nr] = bq[(2 * s) * p.row_len + nr * 32];
        };
        req(0, 0);
        int q = 0, j = 0;
        for (int g = 0; g < ng; ++g) {
          float4 ac[MR];
          float bc[4][NR];
#pragma unroll
          for (int mr = 0; mr < MR; ++mr) ac[mr] = an[mr];
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) bc[s][nr] = bn[s][nr];
          if (++j == p.ktaps) { j = 0; ++q; }
          if (g + 1 < ng) req(q, j);
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) {
              const float av = pick4(ac[mr], s);
#pragma unroll
              for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bc[s][nr], acc[mr][nr], 0, 0, 0);
            }
        }
      }
    } else
    if (wave_active) {
      // Group-level software pipeline (a group = 4 k-steps = 8 input channels of one tap): at the top of
      // iteration gi the fragments of group gi (requested one iteration earlier) are moved into the "cur"
      // registers, then the weight fragments (global/L2) and activation fragments (LDS) of group gi+1 are
      // requested, then the MFMAs of group gi run.  sched_barrier pins the requests above the MFMAs; the
      // weight stream keeps running ahead across the chunk barrier.
      const int nsub = min(sub_per_stage, p.nchunks - ch);
      const int ngroups = nsub * p.ktaps * (KC / 8);
      const int groups_per_sub = p.ktaps * (KC / 8);
      int gsub = 0;                       // groups done in the current 32-channel sub-chunk
      const float* bp = bp0;
      float b0[4][NR], b1[4][NR];        // ping-pong activation fragments (no register moves); a_nxt/a_cur likewise
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) b0[s][nr] = bp[(2 * s) * p.row_len + nr * 32];
      int g = 0;
      // The next group's fragment requests are issued after the first k-step's MFMAs so that their issue overlaps
      // MFMA execution instead of preceding it.  ngroups is a multiple of 4, so two groups per trip is exact.
      auto run_group = [&](float4(&ac)[MR], float(&bc)[4][NR], float4(&an)[MR], float(&bn)[4][NR], bool last_group) {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
          const float av = ac[mr].x;
#pragma unroll
          for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bc[0][nr], acc[mr][nr], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        ++ksg;
        const int kn = ksg < ksg_last ? ksg : ksg_last;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) an[mr] = wp4[abase[mr] + (long long)kn * 64];
        // next group: +8 channels; after 32 channels next tap; after the last tap the next 32-channel sub-chunk
        ++gsub;
        const float* bpn = (g != KC / 8 - 1) ? bp + 8 * p.row_len
                           : (gsub != groups_per_sub ? bp + p.dil - (KC - 8) * p.row_len
                                                     : bp + 8 * p.row_len - (p.ktaps - 1) * p.dil);
        if (gsub == groups_per_sub) gsub = 0;
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
            const float av = pick4(ac[mr], s);
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bc[s][nr], acc[mr][nr], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      for (int gi = 0; gi < ngroups; gi += 2) {
        run_group(a_nxt, b0, a_cur, b1, false);
        run_group(a_cur, b1, a_nxt, b0, gi + 2 >= ngroups);
      }
    }
  }

  if (p.dbg) tstamp[2] = dbg_clock();
  if constexpr (KS) {
    // sum the four K-quarters: waves 1..3 park their accumulators in LDS (the staging tile is dead), wave 0 adds them
    __syncthreads();
    float* red = xs;                                       // [3][MR*NR][16][64]
    if (wave != 0) {
#pragma unroll
      for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
          for (int i = 0; i < 16; ++i) red[(((wave - 1) * MR * NR + mr * NR + nr) * 16 + i) * 64 + lane] = acc[mr][nr][i];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[mr][nr][i] += red[((w * MR * NR + mr * NR + nr) * 16 + i) * 64 + lane];
  }
  if (!wave_active) return;

  auto run_epilogue = [&]() {
  // ------------------------------------------------------------------ epilogues
  const float* maskb = p.mask ? p.mask + (long long)b * p.mask_bs : nullptr;
  const float* gaddb = p.gadd ? p.gadd + (long long)b * p.gadd_bs : nullptr;

  if (p.mode == EPI_PLAIN) {
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
      if (mt0 + mr >= p.mtiles) break;
      // split_row is a multiple of 32, so the output set is uniform per accumulator tile
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
        // fast paths (decoder ResBlocks): every load of an accumulator tile is issued before its first use
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
          const int col = ncol0 + nr * 32 + l31;
          if (col >= p.Ncols) continue;
          float* ybase = oy + (long long)b * oy_bs + (long long)(rbase + 4 * hi) * oy_ld + col;
          float rv[16], yo[16];
          if (fl & F_RES) {
            const float* rbase_p = ores + (long long)b * ores_bs + (long long)(rbase + 4 * hi) * ores_ld + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[r] = rbase_p[(long long)((r & 3) + 8 * (r >> 2)) * ores_ld];
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
          if (fl & F_LOGCLAMP) v = logf(fmaxf(v, p.log_clamp));
          *yp = v;
        }
      }
    }
    return;
  }

  if (p.mode == EPI_UPS) {
    const EpiOut& o = p.out[0];
    const int s = p.ups_s;
    float* yb = o.y + (long long)b * o.y_bs;
    const bool vec = (o.flags & F_VECST) != 0;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
      if (mt0 + mr >= p.mtiles) break;
#pragma unroll
      for (int nr = 0; nr < NR; ++nr) {
        const int col = ncol0 + nr * 32 + l31;
        if (col >= p.Ncols) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row0 = (mt0 + mr) * 32 + 8 * q + 4 * hi;   // rows row0..row0+3 are accumulator regs 4q..4q+3
          if (vec && row0 + 3 < o.nrows) {   // s == 8: the four rows are four consecutive output samples of one channel
            const int oc = row0 >> 3;
            const int n = col * 8 + (row0 & 7) - p.ups_pad;
            float4 v;
            v.x = acc[mr][nr][4 * q + 0];
            v.y = acc[mr][nr][4 * q + 1];
            v.z = acc[mr][nr][4 * q + 2];
            v.w = acc[mr][nr][4 * q + 3];
            float* yp = yb + (long long)oc * o.y_ld + n;
            if (n >= 0 && n + 3 < p.Lout) {
              *reinterpret_cast<float4*>(yp) = v;
            } else {
              if (n >= 0 && n < p.Lout) yp[0] = v.x;
              if (n + 1 >= 0 && n + 1 < p.Lout) yp[1] = v.y;
              if (n + 2 >= 0 && n + 2 < p.Lout) yp[2] = v.z;
              if (n + 3 >= 0 && n + 3 < p.Lout) yp[3] = v.w;
            }
          } else if (s == 2 && row0 + 3 < o.nrows) {   // the four rows are two consecutive samples of two channels
            const int oc = row0 >> 1;
            const int n = col * 2 - p.ups_pad;
            float* yp = yb + (long long)oc * o.y_ld + n;
            if (n >= 0 && n + 1 < p.Lout) {
              yp[0] = acc[mr][nr][4 * q + 0]; yp[1] = acc[mr][nr][4 * q + 1];
              yp[o.y_ld] = acc[mr][nr][4 * q + 2]; yp[o.y_ld + 1] = acc[mr][nr][4 * q + 3];
            } else {
              if (n >= 0 && n < p.Lout) { yp[0] = acc[mr][nr][4 * q + 0]; yp[o.y_ld] = acc[mr][nr][4 * q + 2]; }
              if (n + 1 >= 0 && n + 1 < p.Lout) { yp[1] = acc[mr][nr][4 * q + 1]; yp[o.y_ld + 1] = acc[mr][nr][4 * q + 3]; }
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
    return;
  }

  // ---- paired-tile modes: accumulator tiles (mr, mr+1) hold the two halves of the same channels
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
            o.y[yo] = gate_tanh_sigmoid(vA, vB);
          } else if (p.mode == EPI_MAG) {
            o.y[yo] = sqrtf(vA * vA + vB * vB + p.mag_eps);
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
  };
  run_epilogue();
  if (p.dbg && threadIdx.x == 0) {
    tstamp[3] = dbg_clock();
    long long* d = p.dbg + 4 * dbg_lin;
    d[0] = tstamp[0]; d[1] = tstamp[1]; d[2] = tstamp[2]; d[3] = tstamp[3];
  }
}

template <int WM, int WN, int MR, int NR>
__global__ void __launch_bounds__(WM* WN * 64, (MR * NR >= 8 ? 2 : 3)) conv_mfma_kernel(const ConvArgs p) {
  const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const int tl = xcd_linear(lin, gridDim.x * gridDim.y * gridDim.z, p.xcd);
  const int t = tl / (int)gridDim.x;
  const int bz = t / (int)gridDim.y;
  conv_tile<WM, WN, MR, NR>(p, tl - t * (int)gridDim.x, t - bz * (int)gridDim.y, bz, lin);
}

// K-split variant for launches that cannot fill the chip otherwise (see conv_tile): 4 waves, one (32*MR) x 32 tile.
template <int MR>
__global__ void __launch_bounds__(256, 3) conv_ksplit_kernel(const ConvArgs p) {
  const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const int tl = xcd_linear(lin, gridDim.x * gridDim.y * gridDim.z, p.xcd);
  const int t = tl / (int)gridDim.x;
  const int bz = t / (int)gridDim.y;
  conv_tile<2, 2, MR, 1, true>(p, tl - t * (int)gridDim.x, t - bz * (int)gridDim.y, bz, lin);
}

// Several independent convolutions of one tile shape in ONE launch (the three MRF chains' step-i convolutions):
// workgroups are numbered problem by problem, longest tiles first, so the short ones fill the tail of the long ones
// and there is one tail per launch instead of one per convolution.
template <int WM, int WN, int MR, int NR>
__global__ void __launch_bounds__(WM* WN * 64, (MR * NR >= 8 ? 2 : 3)) conv_group_kernel(const ConvGroup g) {
  const int lin = blockIdx.x;
  int pi = 0;
  if (lin >= g.end[0]) pi = 1;
  if (lin >= g.end[1]) pi = 2;
  const int first = pi == 0 ? 0 : g.end[pi - 1];
  const ConvArgs& p = g.a[pi];
  const int local = xcd_linear(lin - first, g.end[pi] - first, p.xcd);
  const int t = local / p.ntn;
  const int bz = t / p.gy;
  conv_tile<WM, WN, MR, NR>(p, local - t * p.ntn, t - bz * p.gy, bz, lin);
}

// ------------------------------------------------------------------ host side
namespace {

struct TileCfg { int WM, WN, MR, NR; int ks = 0; };
constexpr TileCfg CFG_B{2, 2, 2, 2};   // 128 x 128
constexpr TileCfg CFG_C{2, 2, 1, 4};   //  64 x 256
constexpr TileCfg CFG_D2{1, 4, 1, 2};  //  32 x 256
constexpr TileCfg CFG_E{2, 2, 2, 1};   // 128 x  64  (short sequences, paired)
constexpr TileCfg CFG_F{2, 2, 1, 1};   //  64 x  64  (short sequences)
constexpr TileCfg CFG_G{1, 4, 1, 1};   //  32 x 128  (short sequences, odd tile counts)

template <int MR>
int launch_ks(const ConvArgs& a, int B, hipStream_t st) {
  auto kern = conv_ksplit_kernel<MR>;
  SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
  const size_t lds = std::max((size_t)a.kcs * a.row_len * sizeof(float), (size_t)3 * MR * 16 * 64 * sizeof(float));
  if (lds > 160 * 1024) SVOC_FAIL(SVOC_ERR_UNSUPPORTED, "conv LDS tile of %zu bytes exceeds 160 KiB (kernel %d taps, dilation %d)", lds, a.ktaps, a.dil);
  dim3 grid((a.Ncols + 31) / 32, (a.mtiles + MR - 1) / MR, B);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
  SVOC_HIP(hipGetLastError());
  return SVOC_OK;
}

template <int WM, int WN, int MR, int NR>
int launch_cfg(const ConvArgs& a, int B, hipStream_t st) {
  constexpr int BN = WN * NR * 32;
  auto kern = conv_mfma_kernel<WM, WN, MR, NR>;
  SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
  const size_t lds = (size_t)a.kcs * a.row_len * sizeof(float);
  if (lds > 160 * 1024) SVOC_FAIL(SVOC_ERR_UNSUPPORTED, "conv LDS tile of %zu bytes exceeds 160 KiB (kernel %d taps, dilation %d)", lds, a.ktaps, a.dil);
  dim3 grid((a.Ncols + BN - 1) / BN, (a.mtiles + WM * MR - 1) / (WM * MR), B);
  hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, st, a);
  SVOC_HIP(hipGetLastError());
  return SVOC_OK;
}

}  // namespace

namespace {
// Completes `a` from the packed convolution and picks the tile configuration.
int prepare_conv(const PackedConv& pc, ConvArgs& a, int B, TileCfg& c) {
  a.wp = pc.wp.f();
  a.bias = pc.bias.f();
  a.Cin = pc.Cin;
  a.nchunks = pc.CinP / KC;
  a.ktaps = pc.ktaps;
  a.dil = pc.dil;
  a.pad = pc.pad;
  a.mtiles = pc.mtiles;
  a.ksg_total = pc.ksg_total;
  a.half_rows = pc.half_rows;
  a.ups_s = pc.ups_s;
  a.ups_pad = pc.ups_pad;
  a.vec4 = ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (a.x_ld & 3) == 0 && (a.x_bs & 3) == 0) ? 1 : 0;
  a.xcd = xcd_mapping_enabled();
  if (a.mode == EPI_UPS) {
    const EpiOut& o = a.out[0];
    if (pc.ups_s == 8 && (pc.ups_pad & 3) == 0 && (reinterpret_cast<uintptr_t>(o.y) & 15) == 0 && (o.y_ld & 3) == 0 &&
        (o.y_bs & 3) == 0)
      a.out[0].flags |= F_VECST;
  }
  const bool needs_pair = a.mode >= EPI_GATE;
  if (needs_pair && !pc.paired) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "paired epilogue on an unpaired convolution");

  // Tile configuration.  Rows per workgroup must not leave whole waves idle (mt % WM*MR); among the eligible
  // shapes take the largest one that still yields at least one workgroup per CU, otherwise the one with the
  // most workgroups: a big tile is a long chain of dependent MFMAs, which is pure latency when it cannot be
  // overlapped with other tiles (small batches / short utterances).
  const int mt = pc.mtiles;
  const int ncu = device_cu_count();
  const int Bv = variant_batch(B);                       // the batch the variant is chosen for (svoc_set_variant_batch)
  TileCfg cand[6];
  int ncand = 0;
  if (needs_pair) { cand[ncand++] = CFG_B; cand[ncand++] = CFG_E; }
  else {
    // (a 256x128 tile - 8 accumulator tiles per wave, occupancy 2, register spills - measured slower than two 128x128
    // row blocks for every C=256 layer, profiles/r01_c_*; removed in round 3)
    if (mt % 4 == 0) cand[ncand++] = CFG_B;
    if (mt % 2 == 0) cand[ncand++] = CFG_C;
    if (mt % 2 != 0) cand[ncand++] = CFG_D2;
    if (mt % 2 == 0) cand[ncand++] = CFG_F;
    cand[ncand++] = CFG_G;
  }
  c = cand[0];
  long long best = -1;
  for (int i = 0; i < ncand; ++i) {
    const int bn = cand[i].WN * cand[i].NR * 32, bm = cand[i].WM * cand[i].MR;
    const long long nb = (long long)((a.Ncols + bn - 1) / bn) * ((mt + bm - 1) / bm) * Bv;
    if (nb >= ncu) { c = cand[i]; best = nb; break; }
    if (nb > best) { c = cand[i]; best = nb; }
  }

  {   // short inputs: when even the smallest tile leaves most CUs without a workgroup, split K over the four waves
    static const bool ks_on = !(getenv("SVOC_KSPLIT") && atoi(getenv("SVOC_KSPLIT")) == 0);
    const int mrk = needs_pair ? 2 : 1;
    const long long nbk = (long long)((a.Ncols + 31) / 32) * ((mt + mrk - 1) / mrk) * Bv;
    const int groups = a.nchunks * pc.ktaps;             // per wave: groups of 4 k-steps
    // (measured: extending this to "fewer than two workgroups per CU" is neutral at 1 x 200 and 7 % slower at 4 x 512)
    if (ks_on && best < (long long)ncu && nbk > best && groups >= 2 && a.mode != EPI_UPS && a.mode != EPI_MAG) {
      c = TileCfg{1, 1, mrk, 1, 1};
    }
  }
  const int BN = c.WN * c.NR * 32;
  const int off_first = -pc.pad, off_last = (pc.ktaps - 1) * pc.dil - pc.pad;
  const int minoff = std::min(off_first, off_last), maxoff = std::max(off_first, off_last);
  a.xoff0 = minoff & ~3;                                  // floor to a multiple of 4 (two's complement)
  a.row_len = round_up(BN + maxoff - a.xoff0, 4);
  {   // input channels staged per memory round trip: as many 32-channel chunks as fit the per-block LDS budget
    // ... and one batch of staging loads (9 float4 per thread x 256 threads)
    const int budget = ((c.MR * c.NR >= 8 && !c.ks) ? 6 : 9) * 256 * 16;
    const int per32 = KC * a.row_len * (int)sizeof(float);
    const int nfit = std::max(1, budget / per32);
    a.kcs = KC * std::min(a.nchunks, nfit);
  }
  a.ntn = (a.Ncols + BN - 1) / BN;
  a.gy = c.ks ? (a.mtiles + c.MR - 1) / c.MR : (a.mtiles + c.WM * c.MR - 1) / (c.WM * c.MR);
  a.B = B;
  return SVOC_OK;
}
}  // namespace

int launch_conv(const PackedConv& pc, ConvArgs a, int B, hipStream_t st) {
  if (B <= 0 || a.Ncols <= 0) return SVOC_OK;
  TileCfg c{};
  SVOC_TRY(prepare_conv(pc, a, B, c));

  const double flops = pc.flops_per_col * (double)B * (double)(pc.transposed ? a.Lin : a.Ncols);
  stats_add_conv(flops);
  int prof_idx = -1;
  if (prof_enabled()) {
    char d[160];
    snprintf(d, sizeof(d), "%s Ci%-4d Co%-4d k%-2d d%-2d N%-7d B%-3d %dx%dx%dx%d%s m%d", pc.transposed ? "convT" : "conv ", pc.Cin, pc.Cout,
             pc.transposed ? pc.ktaps * pc.ups_s : pc.ktaps, pc.dil, a.Ncols, B, c.WM, c.WN, c.MR, c.NR, c.ks ? " ksplit" : "", a.mode);
    prof_idx = prof_begin(st, d, flops);
  }
  struct ProfEnd { hipStream_t st; int i; ~ProfEnd() { prof_end(st, i); } } prof_end_guard{st, prof_idx};

  if (c.ks) return c.MR == 2 ? launch_ks<2>(a, B, st) : launch_ks<1>(a, B, st);
#define SVOC_LAUNCH(C) if (c.WM == C.WM && c.WN == C.WN && c.MR == C.MR && c.NR == C.NR) return launch_cfg<C.WM, C.WN, C.MR, C.NR>(a, B, st)
  SVOC_LAUNCH(CFG_B);
  SVOC_LAUNCH(CFG_C);
  SVOC_LAUNCH(CFG_D2);
  SVOC_LAUNCH(CFG_E);
  SVOC_LAUNCH(CFG_F);
  SVOC_LAUNCH(CFG_G);
#undef SVOC_LAUNCH
  SVOC_FAIL(SVOC_ERR_UNSUPPORTED, "no tile configuration");
}

// n <= 3 independent convolutions in one launch (all must select the 128x128 tile); returns 1 when the group is not
// eligible, in which case the caller launches them one by one.  Problems should be ordered longest tile first.
int launch_conv_group(const PackedConv* const* pcs, const ConvArgs* as, int n, int B, hipStream_t st) {
  static const bool enabled = !(getenv("SVOC_GROUP") && atoi(getenv("SVOC_GROUP")) == 0);
  if (!enabled || n < 2 || n > 3 || B <= 0) return 1;
  ConvGroup g{};
  size_t lds = 0;
  double flops = 0;
  long long total = 0;
  for (int i = 0; i < n; ++i) {
    g.a[i] = as[i];
    if (g.a[i].Ncols <= 0 || g.a[i].mode != EPI_PLAIN) return 1;
    TileCfg c{};
    SVOC_TRY(prepare_conv(*pcs[i], g.a[i], B, c));
    if (!(c.WM == CFG_B.WM && c.WN == CFG_B.WN && c.MR == CFG_B.MR && c.NR == CFG_B.NR)) return 1;
    lds = std::max(lds, (size_t)g.a[i].kcs * g.a[i].row_len * sizeof(float));
    total += (long long)g.a[i].ntn * g.a[i].gy * B;
    if (total > 0x7fffffffLL) return 1;
    g.end[i] = (int)total;
    flops += pcs[i]->flops_per_col * (double)B * (double)g.a[i].Ncols;
  }
  for (int i = n; i < 3; ++i) g.end[i] = 0x7fffffff;
  if (lds > 160 * 1024) return 1;
  stats_add_conv(flops, n);
  int prof_idx = -1;
  if (prof_enabled()) {
    char d[160];
    snprintf(d, sizeof(d), "group Ci%-4d Co%-4d k%d/%d/%d N%-7d B%-3d", pcs[0]->Cin, pcs[0]->Cout, pcs[0]->ktaps, pcs[1]->ktaps,
             n > 2 ? pcs[2]->ktaps : 0, g.a[0].Ncols, B);
    prof_idx = prof_begin(st, d, flops);
  }
  auto kern = conv_group_kernel<CFG_B.WM, CFG_B.WN, CFG_B.MR, CFG_B.NR>;
  SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
  hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), lds, st, g);
  prof_end(st, prof_idx);
  SVOC_HIP(hipGetLastError());
  return SVOC_OK;
}

}  // namespace svoc
