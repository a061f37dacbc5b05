// Module graph of the SMART-Vocoder inference path, expressed as launch plans over
// the MFMA convolution kernel (conv_mfma.hip) and the small kernels (misc_kernels.hip).
// Every object mirrors one reference class (file:line in include/svoc.h); the host
// code here only sequences launches on the caller's stream.
#include "svoc_internal.h"

#include <cstring>
#include <cstdlib>
#include <new>

namespace svoc {

static ConvArgs mk_args() {
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.pre_slope = 1.0f;
  a.split_row = 1 << 30;
  a.mode = EPI_PLAIN;
  a.out[0].nrows = 1 << 30;
  a.out[1].nrows = 1 << 30;
  return a;
}
static void set_in(ConvArgs& a, const float* x, long long bs, int ld, int Lin) { a.x = x; a.x_bs = bs; a.x_ld = ld; a.Lin = Lin; }
static void set_out(EpiOut& o, float* y, long long bs, int ld, int nrows, unsigned flags = 0) {
  o.y = y; o.y_bs = bs; o.y_ld = ld; o.nrows = nrows; o.flags = flags; o.div = 1.0f;
}
static void set_res(EpiOut& o, const float* r, long long bs, int ld) { o.res = r; o.res_bs = bs; o.res_ld = ld; }
static inline int pad4(int n) { return round_up(n, 4); }
// Row stride of the decoder's stage tensors (padding it by a few cache lines so that the 32 channel rows of a tile
// do not map to the same HBM channels was measured neutral in round 1 and removed).
// (+ 20 floats since round 4: a row in window-major order holds whole q blocks of 4 x dilation samples, up to 4 * 5 - 1 more than L.)
static inline int stage_ld(int n) { return round_up(n, 4) + 20; }

// =================================================================== WN (modules.py:111-185)
struct WNStack {
  int H = 0, K = 0, DR = 1, NL = 0, gin = 0;
  std::vector<std::unique_ptr<PackedConv>> in_l, rs_l;
  std::vector<std::unique_ptr<DevBuf>> in_f25;            // in_layers in Winograd F(2,5) form where wn_fused.hip serves the shape
  std::vector<std::unique_ptr<DevBuf>> rs16;              // res_skip layers as 16x16x4 A operands (wn_small.hip: short inputs; the last one: wn_mesh.hip)
  std::vector<std::unique_ptr<DevBuf>> in_mesh;           // in_layers in wn_mesh.hip's order (a permutation of in_f25)
  DevBuf mesh_ws;                                           // wn_mesh.hip: x / acts hand-over rows, flags, error word, layer table
  std::unique_ptr<PackedConv> cond;
  DevBuf ws;
  DevBuf stack_ws;                                          // wn_stack.hip: halo buffer + per-tile layer counters + error word

  int create(int hidden, int k, int dr, int nl, int gin_, const TensorTable& tab, const std::string& prefix, hipStream_t st) {
    if (hidden <= 0 || k <= 0 || (k % 2) == 0 || nl <= 0 || dr <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "WN: bad hyper-parameters");
    H = hidden; K = k; DR = dr; NL = nl; gin = gin_;
    int d = 1;
    for (int i = 0; i < NL; ++i) {
      PackSpec sp{}; sp.Cin = H; sp.Cout = 2 * H; sp.K = K; sp.dil = d; sp.paired = true;
      in_l.emplace_back(new PackedConv());
      SVOC_TRY(pack_conv_named(*in_l.back(), sp, tab, prefix + "in_layers." + std::to_string(i), st));
      in_f25.emplace_back(new DevBuf());
      SVOC_TRY(pack_wn_f25_named(*in_f25.back(), H, K, d, tab, prefix + "in_layers." + std::to_string(i), st));
      PackSpec rp{}; rp.Cin = H; rp.K = 1;
      if (i < NL - 1) { rp.Cout = 2 * H; rp.split_at = H; } else { rp.Cout = H; }
      rs_l.emplace_back(new PackedConv());
      SVOC_TRY(pack_conv_named(*rs_l.back(), rp, tab, prefix + "res_skip_layers." + std::to_string(i), st));
      rs16.emplace_back(new DevBuf());
      if (K == 5 && d == 1) SVOC_TRY(pack_wn_rs16_named(*rs16.back(), H, rp.Cout, tab, prefix + "res_skip_layers." + std::to_string(i), st));
      in_mesh.emplace_back(new DevBuf());
      if (K == 5 && d == 1) SVOC_TRY(pack_wn_mesh(*in_mesh.back(), in_f25.back()->f(), st));
      d *= DR;
    }
    if (wn_mesh_supported(H, K, DR, NL)) {      // (conditioned calls - g given - take the per-layer chain)
      const PackedConv* il[16]; const float* wm[16]; const float* wr[16];
      bool all = NL <= 16;
      for (int i = 0; i < NL && all; ++i) { il[i] = in_l[i].get(); wm[i] = in_mesh[i]->f(); wr[i] = rs16[i]->f(); all = wm[i] != nullptr && wr[i] != nullptr; }
      if (all) {
        SVOC_TRY(mesh_ws.ensure(wn_mesh_scratch_bytes()));
        SVOC_TRY(wn_mesh_prepare(mesh_ws.f(), il, wm, wr, NL, st));
      }
    }
    if (gin > 0) {
      PackSpec cp{}; cp.Cin = gin; cp.Cout = 2 * H * NL; cp.K = 1;
      cond.reset(new PackedConv());
      SVOC_TRY(pack_conv_named(*cond, cp, tab, prefix + "cond_layer", st));
    }
    if (wn_stack_applies(H, K, DR, NL, 1, 32 * device_cu_count())) {      // (a shape that passes: the scratch is sized by the CU count only)
      const PackedConv* il[16]; const PackedConv* rl[16]; const float* wf[16];
      bool all = NL <= 16;
      for (int i = 0; i < NL && all; ++i) { il[i] = in_l[i].get(); rl[i] = rs_l[i].get(); wf[i] = in_f25[i]->f(); all = wf[i] != nullptr; }
      if (all) {
        SVOC_TRY(stack_ws.ensure(wn_stack_scratch_bytes()));
        SVOC_TRY(wn_stack_prepare(stack_ws.f(), il, rl, wf, NL, st));
      }
    }
    return SVOC_OK;
  }

  size_t need(int B, int T, int g_T) const {
    return (size_t)(4LL * H * pad4(T) * B + 2LL * H * NL * (g_T > 0 ? pad4(g_T) : 0) * B) * sizeof(float);
  }
  int reserve(int B, int T, int g_T) { return ws.ensure(need(B, T, g_T)); }

  // The weight images an unconditioned forward(B, T) streams, by the launch form it takes (0: the persistent mesh launch, 1: the short-input chain,
  // 2: the stack launch / one fused kernel per layer) - for the prefetch pass at the head of Synth::body
  int weight_path(int B, int T) const {
    if (mesh_ws.p && wn_layer_prefers_unfused(B, T) && wn_mesh_applies(H, K, DR, NL, B, T)) return 0;
    if (wn_small_enabled() && wn_layer_prefers_unfused(B, T) && K == 5 && DR == 1) return 1;
    return 2;
  }
  void weight_buffers(int path, std::vector<std::pair<const void*, size_t>>& out) const {
    for (int i = 0; i < NL; ++i) {
      if (path == 0) { out.push_back({in_mesh[i]->p, in_mesh[i]->bytes}); out.push_back({rs16[i]->p, rs16[i]->bytes}); }
      else if (path == 1) { out.push_back({in_f25[i]->p, in_f25[i]->bytes}); out.push_back({rs16[i]->p, rs16[i]->bytes}); if (i == NL - 1) out.push_back({rs_l[i]->wp.p, rs_l[i]->wp.bytes}); }
      else { out.push_back({in_f25[i]->p ? in_f25[i]->p : in_l[i]->wp.p, in_f25[i]->p ? in_f25[i]->bytes : in_l[i]->wp.bytes}); out.push_back({rs_l[i]->wp.p, rs_l[i]->wp.bytes}); }
    }
  }

  // x (already masked by the caller, as the reference's callers do) -> out; both [B][H][ld]
  int forward(hipStream_t st, const float* x, long long x_bs, int x_ld, const float* mask, long long mask_bs,
              const float* g, int g_T, float* out, long long out_bs, int out_ld, int B, int T) {
    SVOC_TRY(async_error_check());                          // a persistent launch of an earlier call that gave up a wait (svoc.h svoc_check_async_error)
    const int Tp = pad4(T);
    const long long per = (long long)H * Tp;
    const int gTp = g ? pad4(g_T) : 0;
    const long long gper = (long long)2 * H * NL * gTp;
    SVOC_TRY(ws.ensure(need(B, T, g ? g_T : 0)));
    float* xa = ws.f();
    float* xb = xa + per * B;
    float* acts = xb + per * B;
    float* acts2 = acts + per * B;
    float* gc = acts2 + per * B;
    if (g) {
      if (!cond) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "WN: g given but the module has gin_channels == 0");
      if (g_T != 1 && g_T != T) SVOC_FAIL(SVOC_ERR_SHAPE, "WN: g must have 1 or T=%d frames, got %d", T, g_T);
      ConvArgs a = mk_args();
      set_in(a, g, (long long)gin * g_T, g_T, g_T);
      a.Ncols = g_T;
      set_out(a.out[0], gc, gper, gTp, 2 * H * NL);
      SVOC_TRY(launch_conv(*cond, a, B, st));
    }
    // x is ping-ponged between xa and xb: layer i reads `src` (with a (k-1)d/2 halo) and writes `dst`
    const float* src = x; long long src_bs = x_bs; int src_ld = x_ld;
    // Short inputs (wn_small.hip): ONE launch per layer - the previous layer's res_skip at the head of the kernel that computes the
    // F(2,5) in_layer and the gate - and the last layer's res_skip as a convolution behind the chain
    // ... the shortest of them (at most ncu / 24 column tiles) as ONE persistent launch (wn_mesh.hip)
    if (!g && mesh_ws.p && wn_layer_prefers_unfused(B, T) && wn_mesh_applies(H, K, DR, NL, B, T)) {
      const PackedConv* il[16]; const PackedConv* rl[16];
      for (int i = 0; i < NL; ++i) { il[i] = in_l[i].get(); rl[i] = rs_l[i].get(); }
      const int r = launch_wn_mesh_f25(il, rl, NL, H, src, src_bs, src_ld, out, out_bs, out_ld, mask, mask_bs, mesh_ws.f(), B, T, st);
      if (r < 0) return r;
      if (r == 0) return SVOC_OK;
    }
    {
      bool chain = wn_small_enabled() && wn_layer_prefers_unfused(B, T) && K == 5 && DR == 1;
      for (int i = 0; i < NL && chain; ++i) chain = in_f25[i]->p != nullptr && (i == NL - 1 || rs16[i]->p != nullptr);
      if (chain) {
        float* ab[2] = {acts, acts2};
        for (int i = 0; i < NL; ++i) {
          float* dst = (i & 1) ? xb : xa;
          const float* gl = g ? gc + (long long)i * 2 * H * gTp : nullptr;
          const int r = launch_wn_small_layer(*in_l[i], in_f25[i]->f(), i > 0 ? rs16[i - 1]->f() : nullptr, i > 0 ? rs_l[i - 1]->flops_per_col : 0.0,
                                              src, src_bs, src_ld, i > 0 ? ab[(i - 1) & 1] : nullptr, per, Tp, dst, per, Tp, out, out_bs, out_ld,
                                              ab[i & 1], per, Tp, mask, mask_bs, gl, gper, gTp, g_T == 1 ? 0 : 1, i == 1 ? 1 : 0, B, T, st);
          if (r != 0) SVOC_FAIL(SVOC_ERR_UNSUPPORTED, "WN: the short-input layer kernel refused a shape it had accepted");
          if (i > 0) { src = dst; src_bs = per; src_ld = Tp; }
        }
        ConvArgs a = mk_args();                             // res_skip of the last layer: out = (out + rs) * mask (modules.py:173-175)
        set_in(a, ab[(NL - 1) & 1], per, Tp, T);
        a.Ncols = T;
        a.mask = mask; a.mask_bs = mask_bs;
        set_out(a.out[0], out, out_bs, out_ld, H, (NL == 1 ? 0u : (unsigned)F_ACC) | F_OUTMASK);
        return launch_conv(*rs_l[NL - 1], a, B, st);
      }
    }
    // The whole stack in ONE persistent launch (wn_stack.hip) while every 32-column tile has a CU of its own and nothing conditions the layers
    if (!g && stack_ws.p && wn_stack_applies(H, K, DR, NL, B, T)) {
      const PackedConv* il[16]; const PackedConv* rl[16]; const float* wf[16];
      bool all = true;
      for (int i = 0; i < NL; ++i) { il[i] = in_l[i].get(); rl[i] = rs_l[i].get(); wf[i] = in_f25[i]->f(); all = all && wf[i] != nullptr; }
      if (all) {
        const int r = launch_wn_stack_f25(il, rl, wf, NL, H, src, src_bs, src_ld, out, out_bs, out_ld, mask, mask_bs, stack_ws.f(), B, T, st);
        if (r < 0) return r;
        if (r == 0) return SVOC_OK;
      }
    }
    // ... and batches beyond its capacity as one such launch per group of whole utterances (32 x 512: two of sixteen; 8 x 4096: four of two) - round 6
    if (const int ng = (!g && stack_ws.p) ? wn_stack_groups(H, K, DR, NL, B, T) : 0) {
      const PackedConv* il[16]; const PackedConv* rl[16]; const float* wf[16];
      bool all = true;
      for (int i = 0; i < NL; ++i) { il[i] = in_l[i].get(); rl[i] = rs_l[i].get(); wf[i] = in_f25[i]->f(); all = all && wf[i] != nullptr; }
      if (all) {
        const int bg = B / ng;
        for (int gi = 0; gi < ng; ++gi) {
          const long long b0 = (long long)gi * bg;
          const int r = launch_wn_stack_f25(il, rl, wf, NL, H, src + b0 * src_bs, src_bs, src_ld, out + b0 * out_bs, out_bs, out_ld, mask + b0 * mask_bs, mask_bs,
                                            stack_ws.f(), bg, T, st, gi == 0);
          if (r < 0) return r;
          if (r != 0) SVOC_FAIL(SVOC_ERR_UNSUPPORTED, "WN: the stack launch refused a group it had accepted");
        }
        return SVOC_OK;
      }
    }
    for (int i = 0; i < NL; ++i) {
      const bool last = i == NL - 1;
      float* dst = (i & 1) ? xb : xa;
      const float* gl = g ? gc + (long long)i * 2 * H * gTp : nullptr;
      const int gts = g_T == 1 ? 0 : 1;
      {   // whole layer in one kernel (wn_fused.hip) when eligible
        const int r = launch_wn_layer_fused(*in_l[i], *rs_l[i], H, src, src_bs, src_ld, dst, per, Tp, out, out_bs, out_ld, mask, mask_bs,
                                            gl, gper, gTp, gts, i == 0 ? 1 : 0, last ? 1 : 0, B, T, st, in_f25[i]->f());
        if (r < 0) return r;
        if (r == 0) { src = dst; src_bs = per; src_ld = Tp; continue; }
      }
      {   // in_layer + fused_add_tanh_sigmoid_multiply (commons.py:100-107)
        ConvArgs a = mk_args();
        set_in(a, src, src_bs, src_ld, T);
        a.Ncols = T;
        a.mode = EPI_GATE;
        set_out(a.out[0], acts, per, Tp, H);
        if (g) { a.gadd = gl; a.gadd_bs = gper; a.gadd_ld = gTp; a.gadd_ts = gts; }
        SVOC_TRY(launch_conv(*in_l[i], a, B, st));
      }
      {   // res_skip 1x1 + residual/skip bookkeeping (modules.py:168-175)
        ConvArgs a = mk_args();
        set_in(a, acts, per, Tp, T);
        a.Ncols = T;
        a.mask = mask; a.mask_bs = mask_bs;
        if (!last) {
          a.split_row = rs_l[i]->split_row;
          set_out(a.out[0], dst, per, Tp, H, F_RES | F_OUTMASK);
          set_res(a.out[0], src, src_bs, src_ld);
          set_out(a.out[1], out, out_bs, out_ld, H, i == 0 ? 0u : (unsigned)F_ACC);
        } else {
          set_out(a.out[0], out, out_bs, out_ld, H, (i == 0 ? 0u : (unsigned)F_ACC) | F_OUTMASK);
        }
        SVOC_TRY(launch_conv(*rs_l[i], a, B, st));
      }
      src = dst; src_bs = per; src_ld = Tp;
    }
    return SVOC_OK;
  }
};

// =================================================================== ResBlock1 / ResBlock2 (modules.py:187-256)
struct ResSink { float* y; long long bs; int ld; unsigned flags; float div; };

struct ResBlock {
  int kind = 1, C = 0, K = 0, ND = 0;
  std::vector<std::unique_ptr<PackedConv>> c1, c2;
  std::vector<std::unique_ptr<PackedWino>> w1, w2;   // Winograd form of the undilated convolutions (null where not applicable)
  DevBuf ws;

  int create(int kind_, int channels, int k, const int* dil, int nd, const TensorTable& tab, const std::string& prefix, hipStream_t st) {
    if ((kind_ != 1 && kind_ != 2) || channels <= 0 || k <= 0 || (k % 2) == 0 || nd <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "ResBlock: bad hyper-parameters");
    kind = kind_; C = channels; K = k; ND = nd;
    for (int i = 0; i < ND; ++i) {
      PackSpec sp{}; sp.Cin = C; sp.Cout = C; sp.K = K; sp.dil = dil[i];
      c1.emplace_back(new PackedConv());
      const std::string n1 = prefix + (kind == 1 ? "convs1." : "convs.") + std::to_string(i);
      SVOC_TRY(pack_conv_named(*c1.back(), sp, tab, n1, st));
      w1.emplace_back(nullptr);
      if (wino_supported(C, C, K, dil[i])) {                   // C = 32: F(4,3) only (round 4), else the fused direct kernel
        w1.back().reset(new PackedWino());
        SVOC_TRY(pack_wino_named(*w1.back(), C, C, K, tab, n1, st));
      }
      if (kind == 1) {
        PackSpec s2{}; s2.Cin = C; s2.Cout = C; s2.K = K; s2.dil = 1;
        c2.emplace_back(new PackedConv());
        SVOC_TRY(pack_conv_named(*c2.back(), s2, tab, prefix + "convs2." + std::to_string(i), st));
        w2.emplace_back(nullptr);
        if (wino_supported(C, C, K, 1)) {
          w2.back().reset(new PackedWino());
          SVOC_TRY(pack_wino_named(*w2.back(), C, C, K, tab, prefix + "convs2." + std::to_string(i), st));
        }
      }
    }
    return SVOC_OK;
  }

  // x read-only [B][C][x_ld]; A, Bf scratch [B][C][ld]; the block's result goes to `sink`
  int run(hipStream_t st, const float* x, long long x_bs, int x_ld, const float* mask, long long mask_bs, float* A,
          float* Bf, long long bs, int ld, const ResSink& sink, int B, int L, hipEvent_t wait_before_last = nullptr,
          hipEvent_t record_after_last = nullptr) {
    const float* cur = x; long long cur_bs = x_bs; int cur_ld = x_ld;
    for (int i = 0; i < ND; ++i) {
      const bool last = i == ND - 1;
      if (kind == 1 && mask == nullptr) {
        // fused iteration (resblock_fused.hip): reads `cur` with halos, so it must not write in place -> ping-pong Bf/A
        float* dst = last ? sink.y : ((i & 1) ? A : Bf);
        const long long dbs = last ? sink.bs : bs;
        const int dld = last ? sink.ld : ld;
        if (last && wait_before_last) SVOC_HIP(hipStreamWaitEvent(st, wait_before_last, 0));
        const int r = launch_resblock_fused(*c1[i], *c2[i], cur, cur_bs, cur_ld, dst, dbs, dld, last ? sink.flags : 0u,
                                            last ? sink.div : 1.0f, B, L, st);
        if (r < 0) return r;
        if (r == 0) {
          if (last && record_after_last) SVOC_HIP(hipEventRecord(record_after_last, st));
          cur = dst; cur_bs = dbs; cur_ld = dld;
          continue;
        }
      }
      const float* cin = cur; long long cin_bs = cur_bs; int cin_ld = cur_ld;
      // unfused: c1 -> scratch, c2 (+ residual) -> next.  The scratch must differ from `cur` (which may be A or Bf
      // after fused iterations).
      float* scratch = (cur == A) ? Bf : A;
      float* nxt = (cur == A || cur == Bf) ? const_cast<float*>(cur) : Bf;   // in place is safe here (elementwise residual)
      if (kind == 1) {
        ConvArgs a = mk_args();
        set_in(a, cur, cur_bs, cur_ld, L);
        a.pre_slope = 0.1f; a.in_mask = mask; a.in_mask_bs = mask_bs;
        a.Ncols = L;
        set_out(a.out[0], scratch, bs, ld, C);
        int rw = w1[i] ? launch_conv_wino(*w1[i], a, B, c1[i]->dil, st) : 1;
        if (rw < 0) return rw;
        if (rw == 1) SVOC_TRY(launch_conv(*c1[i], a, B, st));
        cin = scratch; cin_bs = bs; cin_ld = ld;
      }
      ConvArgs a = mk_args();
      set_in(a, cin, cin_bs, cin_ld, L);
      a.pre_slope = 0.1f; a.in_mask = mask; a.in_mask_bs = mask_bs;
      a.Ncols = L;
      a.mask = mask; a.mask_bs = mask_bs;
      if (!last) {
        set_out(a.out[0], nxt, bs, ld, C, F_RES);
      } else {
        set_out(a.out[0], sink.y, sink.bs, sink.ld, C, F_RES | sink.flags | (mask ? (unsigned)F_OUTMASK : 0u));
        a.out[0].div = sink.div;
      }
      set_res(a.out[0], cur, cur_bs, cur_ld);
      if (last && wait_before_last) SVOC_HIP(hipStreamWaitEvent(st, wait_before_last, 0));
      {
        const PackedWino* pw = kind == 1 ? w2[i].get() : w1[i].get();
        int rw = pw ? launch_conv_wino(*pw, a, B, kind == 1 ? 1 : c1[i]->dil, st) : 1;
        if (rw < 0) return rw;
        if (rw == 1) SVOC_TRY(launch_conv(kind == 1 ? *c2[i] : *c1[i], a, B, st));
      }
      if (last && record_after_last) SVOC_HIP(hipEventRecord(record_after_last, st));
      cur = nxt; cur_bs = bs; cur_ld = ld;
    }
    return SVOC_OK;
  }

  int forward(hipStream_t st, const float* x, const float* mask, float* y, int B, int L) {
    const int ld = pad4(L);
    const long long bs = (long long)C * ld;
    SVOC_TRY(ws.ensure((size_t)(2 * bs * B) * sizeof(float)));
    ResSink sink{y, (long long)C * L, L, 0u, 1.0f};
    return run(st, x, (long long)C * L, L, mask, L, ws.f(), ws.f() + bs * B, bs, ld, sink, B, L);
  }
};

// =================================================================== ResidualCouplingLayer (modules.py:298-343)
struct Coupling {
  int C = 0, half = 0, H = 0, mean_only = 0, flipped = 0;
  PackedConv pre, post;
  WNStack enc;
  DevBuf ws;

  int create(int channels, int hidden, int k, int dr, int nl, int gin, int mean_only_, int flipped_, const TensorTable& tab,
             const std::string& prefix, hipStream_t st) {
    if (channels <= 0 || channels % 2) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "coupling: channels should be divisible by 2");
    C = channels; half = C / 2; H = hidden; mean_only = mean_only_; flipped = flipped_;
    // Flip folding (modules.py:272): in flipped orientation logical x0[c] lives at physical row C-1-c
    // (block [half,C), reversed) and logical x1[c] at physical row half-1-c (block [0,half), reversed).
    std::vector<int> rev(half), rev2(2 * half);
    for (int q = 0; q < half; ++q) rev[q] = half - 1 - q;
    for (int q = 0; q < half; ++q) { rev2[q] = half - 1 - q; rev2[half + q] = half + (half - 1 - q); }
    PackSpec ps{}; ps.Cin = half; ps.Cout = H; ps.K = 1; ps.in_perm = flipped ? rev.data() : nullptr;
    SVOC_TRY(pack_conv_named(pre, ps, tab, prefix + "pre", st));
    SVOC_TRY(enc.create(H, k, dr, nl, gin, tab, prefix + "enc.", st));
    PackSpec qs{}; qs.Cin = H; qs.K = 1;
    if (mean_only) { qs.Cout = half; qs.out_perm = flipped ? rev.data() : nullptr; }
    else { qs.Cout = 2 * half; qs.paired = true; qs.out_perm = flipped ? rev2.data() : nullptr; }
    SVOC_TRY(pack_conv_named(post, qs, tab, prefix + "post", st));
    return SVOC_OK;
  }

  int reserve(int B, int T, int g_T) {
    SVOC_TRY(ws.ensure((size_t)(2LL * H * pad4(T) * B) * sizeof(float)));
    return enc.reserve(B, T, g_T);
  }

  // src/dst: physical [B][C][ld] buffers (dst may equal src).  Only the x1 block of dst is written;
  // when dst != src the caller copies the x0 block.
  int run(hipStream_t st, const float* src, long long s_bs, int s_ld, float* dst, long long d_bs, int d_ld,
          const float* mask, long long mask_bs, const float* g, int g_T, int reverse, float* logdet, int B, int T) {
    const int Tp = pad4(T);
    const long long per = (long long)H * Tp;
    SVOC_TRY(ws.ensure((size_t)(2 * per * B) * sizeof(float)));
    float* h = ws.f();
    float* wo = h + per * B;
    const int x0_row = flipped ? half : 0, x1_row = flipped ? 0 : half;
    {
      ConvArgs a = mk_args();
      set_in(a, src + (long long)x0_row * s_ld, s_bs, s_ld, T);
      a.Ncols = T; a.mask = mask; a.mask_bs = mask_bs;
      set_out(a.out[0], h, per, Tp, H, F_OUTMASK);
      SVOC_TRY(launch_conv(pre, a, B, st));
    }
    SVOC_TRY(enc.forward(st, h, per, Tp, mask, mask_bs, g, g_T, wo, per, Tp, B, T));
    {
      ConvArgs a = mk_args();
      set_in(a, wo, per, Tp, T);
      a.Ncols = T; a.mask = mask; a.mask_bs = mask_bs;
      set_out(a.out[0], dst + (long long)x1_row * d_ld, d_bs, d_ld, half);
      set_res(a.out[0], src + (long long)x1_row * s_ld, s_bs, s_ld);
      if (mean_only) {
        a.out[0].flags = reverse ? F_CPL_REV : F_CPL_FWD;
      } else {
        a.mode = reverse ? EPI_CPL_FULL_REV : EPI_CPL_FULL_FWD;
        a.logdet = reverse ? nullptr : logdet;
      }
      SVOC_TRY(launch_conv(post, a, B, st));
    }
    return SVOC_OK;
  }
};

// =================================================================== ResidualCouplingBlock (models.py:50-80)
struct Flow {
  int C = 0, NF = 0;
  std::vector<std::unique_ptr<Coupling>> rev_l, fwd_l;   // per direction (shared when the flip parity agrees)
  std::vector<Coupling*> rev_p, fwd_p;
  DevBuf ws;

  int create(int channels, int hidden, int k, int dr, int nl, int nf, int gin, const TensorTable& tab, const std::string& prefix, hipStream_t st) {
    C = channels; NF = nf;
    if (nf <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "flow: n_flows must be positive");
    for (int i = 0; i < NF; ++i) {
      const int f_rev = (NF - i) % 2, f_fwd = i % 2;
      const std::string p = prefix + "flows." + std::to_string(2 * i) + ".";
      rev_l.emplace_back(new Coupling());
      SVOC_TRY(rev_l.back()->create(C, hidden, k, dr, nl, gin, 1, f_rev, tab, p, st));
      rev_p.push_back(rev_l.back().get());
      if (f_fwd == f_rev) { fwd_p.push_back(rev_l.back().get()); }
      else {
        fwd_l.emplace_back(new Coupling());
        SVOC_TRY(fwd_l.back()->create(C, hidden, k, dr, nl, gin, 1, f_fwd, tab, p, st));
        fwd_p.push_back(fwd_l.back().get());
      }
    }
    return SVOC_OK;
  }

  int reserve(int B, int T, int g_T, int reverse) {
    for (auto* c : (reverse ? rev_p : fwd_p)) SVOC_TRY(c->reserve(B, T, g_T));
    return SVOC_OK;
  }

  // in-place on P [B][C][ld]; the physical buffer is never flipped except for a final flip when NF is odd
  int run_inplace(hipStream_t st, float* P, long long bs, int ld, const float* mask, long long mask_bs, const float* g, int g_T,
                  int reverse, int B, int T) {
    if (reverse) { for (int i = NF - 1; i >= 0; --i) SVOC_TRY(rev_p[i]->run(st, P, bs, ld, P, bs, ld, mask, mask_bs, g, g_T, 1, nullptr, B, T)); }
    else { for (int i = 0; i < NF; ++i) SVOC_TRY(fwd_p[i]->run(st, P, bs, ld, P, bs, ld, mask, mask_bs, g, g_T, 0, nullptr, B, T)); }
    return SVOC_OK;
  }

  int forward(hipStream_t st, const float* x, const float* mask, const float* g, int g_T, int reverse, float* y, int B, int T) {
    const int Tp = pad4(T);
    const long long per = (long long)C * Tp;
    SVOC_TRY(ws.ensure((size_t)(per * B) * sizeof(float)));
    float* P = ws.f();
    SVOC_TRY(k_copy2d(st, x, (long long)C * T, T, P, per, Tp, B, C, T, nullptr, 0));
    SVOC_TRY(run_inplace(st, P, per, Tp, mask, T, g, g_T, reverse, B, T));
    if (NF % 2 == 0) return k_copy2d(st, P, per, Tp, y, (long long)C * T, T, B, C, T, nullptr, 0);
    // odd number of flips: one physical channel reversal on the way out
    return k_flip_copy(st, P, per, Tp, y, (long long)C * T, T, B, C, T);
  }
};

// =================================================================== Generator (models.py:115-167)
struct Generator {
  svoc_generator_config cfg{};
  PackedConv conv_pre;
  std::unique_ptr<PackedWino> conv_pre_w;                  // the same convolution in Winograd F(4,4) form (round 6): taken when nothing masks or conditions the input
  std::unique_ptr<PackedConv> cond;
  std::vector<std::unique_ptr<PackedConv>> ups;
  std::vector<std::unique_ptr<PackedCtWino>> ups_w;       // F(4,2) form of the same upsampler where convt_wino.hip serves the shape
  std::vector<std::unique_ptr<ResBlock>> rbs;
  DevBuf conv_post_w;
  int post_C = 0;
  DevBuf ws;
  int hop = 1;
  // The n_kernels ResBlock chains of one MRF stage are independent until their last conv; they run on separate
  // streams so that workgroups of different kernels (different durations) share the CUs: identical co-resident
  // workgroups of ONE kernel stay phase-locked (all stage, all MFMA, all store) and their memory time adds to
  // their MFMA time.  Also fills each kernel's tail.  Ordered with events (hipGraph-capturable fork/join).
  std::vector<hipStream_t> chain_st;
  std::vector<hipEvent_t> chain_done;
  hipEvent_t ev_fork = nullptr;
  bool use_streams = true;
  ~Generator() {
    for (auto s : chain_st) if (s) (void)hipStreamDestroy(s);
    for (auto e : chain_done) if (e) (void)hipEventDestroy(e);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
  }

  int create(const svoc_generator_config& c, const TensorTable& tab, const std::string& prefix, hipStream_t st) {
    cfg = c;
    use_streams = c.n_kernels > 1;                          // one ResBlock chain per stage: everything on the caller's stream
    if (use_streams) {
      chain_st.assign(c.n_kernels, nullptr);
      chain_done.assign(c.n_kernels, nullptr);
      for (int j = 0; j < c.n_kernels; ++j) {
        SVOC_HIP(hipStreamCreateWithFlags(&chain_st[j], hipStreamNonBlocking));
        SVOC_HIP(hipEventCreateWithFlags(&chain_done[j], hipEventDisableTiming));
      }
      SVOC_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    }
    if (c.n_upsamples <= 0 || c.n_upsamples > 8 || c.n_kernels <= 0 || c.n_kernels > 8) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "generator: bad configuration");
    PackSpec ps{}; ps.Cin = c.initial_channel; ps.Cout = c.upsample_initial_channel; ps.K = 7; ps.pad = 3;
    SVOC_TRY(pack_conv_named(conv_pre, ps, tab, prefix + "conv_pre", st));
    if (wino_supported(ps.Cin, ps.Cout, 7, 1)) {
      conv_pre_w.reset(new PackedWino());
      SVOC_TRY(pack_wino_named(*conv_pre_w, ps.Cin, ps.Cout, 7, tab, prefix + "conv_pre", st));
    }
    if (c.gin_channels > 0) {
      PackSpec cs{}; cs.Cin = c.gin_channels; cs.Cout = c.upsample_initial_channel; cs.K = 1;
      cond.reset(new PackedConv());
      SVOC_TRY(pack_conv_named(*cond, cs, tab, prefix + "cond", st));
    }
    int ch = c.upsample_initial_channel;
    hop = 1;
    for (int i = 0; i < c.n_upsamples; ++i) {
      const int u = c.upsample_rates[i], k = c.upsample_kernel_sizes[i];
      if (ch % 2) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "generator: channel count not divisible at stage %d", i);
      PackSpec us{}; us.Cin = ch; us.Cout = ch / 2; us.K = k; us.transposed = true; us.stride = u; us.tpad = (k - u) / 2;
      if (k < u || ((k - u) % 2) != 0) SVOC_FAIL(SVOC_ERR_UNSUPPORTED, "generator: upsample kernel %d / rate %d not supported", k, u);
      ups.emplace_back(new PackedConv());
      SVOC_TRY(pack_conv_named(*ups.back(), us, tab, prefix + "ups." + std::to_string(i), st));
      ups_w.emplace_back(nullptr);
      if (convt_wino_supported(us.Cin, us.Cout, k, u, us.tpad)) {
        ups_w.back().reset(new PackedCtWino());
        SVOC_TRY(pack_convt_wino_named(*ups_w.back(), us.Cin, us.Cout, k, u, us.tpad, tab, prefix + "ups." + std::to_string(i), st));
      }
      ch /= 2;
      hop *= u;
      for (int j = 0; j < c.n_kernels; ++j) {
        rbs.emplace_back(new ResBlock());
        SVOC_TRY(rbs.back()->create(c.resblock_kind, ch, c.resblock_kernel_sizes[j], c.resblock_dilation_sizes[j], c.n_dilations[j],
                                    tab, prefix + "resblocks." + std::to_string(i * c.n_kernels + j) + ".", st));
      }
    }
    post_C = ch;
    const svoc_tensor* pw = tab.find(prefix + "conv_post.weight");
    if (!pw) SVOC_FAIL(SVOC_ERR_MISSING_TENSOR, "missing tensor %sconv_post.weight", prefix.c_str());
    if (pw->ndim != 3 || pw->shape[0] != 1 || pw->shape[1] != ch || pw->shape[2] != 7) SVOC_FAIL(SVOC_ERR_SHAPE, "conv_post.weight has wrong shape");
    SVOC_TRY(conv_post_w.ensure((size_t)ch * 7 * sizeof(float)));
    SVOC_HIP(hipMemcpyAsync(conv_post_w.p, pw->data, (size_t)ch * 7 * sizeof(float), hipMemcpyDeviceToDevice, st));
    SVOC_HIP(hipStreamSynchronize(st));
    return SVOC_OK;
  }

  size_t stage_floats(int T) const {   // largest [C_i][stage_ld(L_i)] over conv_pre output and all stages
    size_t m = (size_t)cfg.upsample_initial_channel * stage_ld(T);
    int ch = cfg.upsample_initial_channel; long long L = T;
    for (int i = 0; i < cfg.n_upsamples; ++i) { ch /= 2; L *= cfg.upsample_rates[i]; m = std::max(m, (size_t)ch * stage_ld((int)L)); }
    return m;
  }
  int n_bufs() const { return use_streams ? 3 + 2 * cfg.n_kernels : 4; }
  size_t workspace_bytes(int B, int T) const { return ((size_t)n_bufs() * stage_floats(T) * B + (size_t)cfg.upsample_initial_channel * B) * sizeof(float); }

  // Grouped MRF (unfused ResBlock1 stages): the n_kernels chains advance in lock step, one launch per step carrying
  // all chains' convolutions, longest tiles (largest kernel size) first.  The stream plan's buffers are reused:
  // (A, Bf) per chain.  The last c2 of each chain accumulates into XS in chain order, so those run one by one.
  bool mrf_grouped(int stage, int C) const {
    static const bool on = !(getenv("SVOC_GROUP") && atoi(getenv("SVOC_GROUP")) == 0);
    static const bool fuse = !(getenv("SVOC_FUSE") && atoi(getenv("SVOC_FUSE")) == 0);
    const int nk = cfg.n_kernels;
    if (!on || nk < 2 || nk > 3) return false;
    // C = 64: measured faster conv by conv in Winograd form (grouped launches) than fused in direct form (8.4 against 9.9 ms per step, round 2)
    // C = 32 (round 4): conv by conv in F(4,3) form with one row tile per workgroup (SVOC_W4_C32=0: the fused kernel)
    bool c32_wino = C == 32 && nk == 3;
    for (int j = 0; j < nk && c32_wino; ++j) {
      const ResBlock& rb = *rbs[stage * nk + j];
      for (int it = 0; it < rb.ND && c32_wino; ++it)
        c32_wino = rb.kind == 1 && rb.w1[it] && rb.w1[it]->wp4.p && rb.w2[it] && rb.w2[it]->wp4.p;
    }
    if (fuse && C == 32 && !c32_wino) return false;        // C = 32 without the F(4,3) images: the fused direct-form ResBlock kernel on the stream plan
    const ResBlock& r0 = *rbs[stage * nk];
    for (int j = 0; j < nk; ++j) {
      const ResBlock& rb = *rbs[stage * nk + j];
      if (rb.kind != 1 || rb.ND != r0.ND) return false;
      // a grouped launch carries ONE dilation for its three members: the chains must agree on it at every step
      // (resblock_dilation_sizes like [[1,3,5],[3,5,1],[5,1,3]] run chain by chain on the stream plan instead)
      for (int it = 0; it < rb.ND; ++it) if (rb.c1[it]->dil != r0.c1[it]->dil) return false;
    }
    return true;
  }

  int run_mrf_grouped(hipStream_t st, int stage, const float* X, float* XS, const std::vector<float*>& bufs, long long bs, int ld,
                      int C, int B, int L) {
    const int nk = cfg.n_kernels;
    int order[3] = {0, 1, 2};
    std::sort(order, order + nk, [&](int a, int b) { return rbs[stage * nk + a]->K > rbs[stage * nk + b]->K; });
    const float* cur[3] = {X, X, X};
    const int ND = rbs[stage * nk]->ND;
    for (int it = 0; it < ND; ++it) {
      const bool last = it == ND - 1;
      const PackedConv* pcs[3];
      const PackedWino* pws[3];
      ConvArgs as[3];
      float* scratch[3];
      float* nxt[3];
      // one step of the three chains in one launch: the Winograd kernel where every member has that form (d = 1), else
      // the direct grouped kernel, else one by one
      auto launch3 = [&](bool wino, int dil) -> int {
        int r = 1;
        if (wino) r = launch_conv_wino_group(pws, as, nk, B, dil, st);
        // only the grouped F(4,3) kernels understand window-major rows: the query above promised them
        if (r == 1 && (as[0].wperm_in || as[0].wperm_out)) SVOC_FAIL(SVOC_ERR_UNSUPPORTED, "MRF: the grouped launch refused window-major rows it had accepted");
        if (r == 1) r = launch_conv_group(pcs, as, nk, B, st);
        if (r < 0) return r;
        if (r == 1) for (int q = 0; q < nk; ++q) SVOC_TRY(launch_conv(*pcs[q], as[q], B, st));
        return SVOC_OK;
      };
      // C = 32 / 64, every step but the last: c1 -> c2 of the three chains in ONE launch, the intermediate tile in LDS (conv_wino4_pair.hip)
      if ((C == 32 || C == 64) && nk == 3 && !last && wino4_pair_enabled() &&
          wino4_pair_tiles(C, L, variant_batch(B), rbs[stage * nk]->c1[it]->dil) >= mrf_min_tiles()) {
        const PackedWino* p1[3]; const PackedWino* p2[3];
        const float* xi[3]; float* yo[3];
        for (int q = 0; q < nk; ++q) {
          const int j = order[q];
          float* A = bufs[3 + 2 * j];
          float* Bf = bufs[4 + 2 * j];
          p1[q] = rbs[stage * nk + j]->w1[it].get();
          p2[q] = rbs[stage * nk + j]->w2[it].get();
          xi[q] = cur[j];
          yo[q] = (cur[j] == A) ? Bf : A;                   // never in place: neighbouring tiles read the input's halo
        }
        const int rp = launch_wino4_pair(p1, p2, xi, yo, bs, ld, B, L, rbs[stage * nk]->c1[it]->dil, 0.1f, st);
        if (rp < 0) return rp;
        if (rp == 0) {
          for (int q = 0; q < nk; ++q) cur[order[q]] = yo[q];
          continue;
        }
      }
      // C = 32, the LAST step (round 5): c1 -> c2 of the three chains AND the sum over the chains in ONE launch - the pair kernel's accumulate form
      // (xs = (rb_11 + rb_7 + rb_3) / 3 built in place by the workgroup that owns the tile; replaces the window-major c1 launch + the merged accumulate launch)
      if (C == 32 && nk == 3 && last && wino4_pair_enabled() &&
          wino4_pair_tiles(C, L, variant_batch(B), rbs[stage * nk]->c1[it]->dil) >= mrf_min_tiles()) {
        const PackedWino* p1[3]; const PackedWino* p2[3];
        const float* xi[3]; float* yo[3];
        for (int q = 0; q < nk; ++q) {
          const int j = order[q];
          p1[q] = rbs[stage * nk + j]->w1[it].get();
          p2[q] = rbs[stage * nk + j]->w2[it].get();
          xi[q] = cur[j];
          yo[q] = XS;
        }
        const int rp = launch_wino4_pair(p1, p2, xi, yo, bs, ld, B, L, rbs[stage * nk]->c1[it]->dil, 0.1f, st, 1, (float)nk);
        if (rp < 0) return rp;
        if (rp == 0) continue;
      }
      bool all_w = true, all_w2 = true;
      const PackedConv* pcs2[3];
      const PackedWino* pws2[3];
      ConvArgs as2[3];
      for (int q = 0; q < nk; ++q) {
        const int j = order[q];
        float* A = bufs[3 + 2 * j];
        float* Bf = bufs[4 + 2 * j];
        scratch[q] = (cur[j] == A) ? Bf : A;
        nxt[q] = (cur[j] == A || cur[j] == Bf) ? const_cast<float*>(cur[j]) : Bf;
        ConvArgs a = mk_args();
        set_in(a, cur[j], bs, ld, L);
        a.pre_slope = 0.1f;
        a.Ncols = L;
        set_out(a.out[0], scratch[q], bs, ld, C);
        as[q] = a;
        pcs[q] = rbs[stage * nk + j]->c1[it].get();
        pws[q] = rbs[stage * nk + j]->w1[it].get();
        all_w = all_w && pws[q] != nullptr;
        ConvArgs a2 = mk_args();
        set_in(a2, scratch[q], bs, ld, L);
        a2.pre_slope = 0.1f;
        a2.Ncols = L;
        set_out(a2.out[0], nxt[q], bs, ld, C, F_RES);
        set_res(a2.out[0], cur[j], bs, ld);
        as2[q] = a2;
        pcs2[q] = rbs[stage * nk + j]->c2[it].get();
        pws2[q] = rbs[stage * nk + j]->w2[it].get();
        all_w2 = all_w2 && pws2[q] != nullptr;
      }
      // xs = sum_j ResBlock_j(x) / n, accumulated in chain order (models.py:149-155): the last step's c2 members in chain order
      const PackedWino* apw[3];
      ConvArgs aas[3];
      if (last && nk == 3) {
        for (int j = 0; j < 3; ++j) {
          int q = 0;
          while (order[q] != j) ++q;
          ConvArgs a = as2[q];
          unsigned fl = F_RES;
          if (j > 0) fl |= F_ACC;
          if (j == nk - 1) fl |= F_DIV;
          set_out(a.out[0], XS, bs, ld, C, fl);
          a.out[0].div = (float)nk;
          set_res(a.out[0], cur[j], bs, ld);
          aas[j] = a; apw[j] = pws2[q];
        }
      }
      // A dilated c1 writes its rows window-major (one 16-byte store per lane and row instead of four scattered dwords) and the c2
      // behind it reads through the same map - when both launches are the grouped F(4,3) kernels
      {
        const int dil = pcs[0]->dil;
        if (dil > 1 && all_w && all_w2 && (!last || nk == 3)) {
          ConvArgs t1[3], t2[3];
          for (int q = 0; q < nk; ++q) { t1[q] = as[q]; t1[q].wperm_out = dil; t2[q] = (last && nk == 3) ? aas[q] : as2[q]; t2[q].wperm_in = dil; }
          const bool ok = launch_conv_wino_group(pws, t1, nk, B, dil, st, true) == 0 &&
                          ((last && nk == 3) ? launch_conv_wino4_accum(apw, t2, B, st, true) == 0 : launch_conv_wino_group(pws2, t2, nk, B, 1, st, true) == 0);
          if (ok) for (int q = 0; q < nk; ++q) { as[q].wperm_out = dil; as2[q].wperm_in = dil; if (last && nk == 3) aas[q].wperm_in = dil; }
        }
      }
      SVOC_TRY(launch3(all_w, pcs[0]->dil));
      for (int q = 0; q < nk; ++q) { as[q] = as2[q]; pcs[q] = pcs2[q]; pws[q] = pws2[q]; }
      all_w = all_w2;
      if (!last) {
        SVOC_TRY(launch3(all_w, 1));
        for (int q = 0; q < nk; ++q) cur[order[q]] = nxt[q];
      } else {
        // xs = sum_j ResBlock_j(x) / n, accumulated in chain order (models.py:149-155): one launch when the three members are the
        // F(4,3) kernel's (conv_wino4_accum_kernel), else one by one
        bool done = false;
        if (nk == 3) {
          const int ra = launch_conv_wino4_accum(apw, aas, B, st);
          if (ra < 0) return ra;
          done = ra == 0;
          if (!done && aas[0].wperm_in) SVOC_FAIL(SVOC_ERR_UNSUPPORTED, "MRF: the accumulate launch refused window-major inputs it had accepted");
        }
        for (int j = 0; j < nk && !done; ++j) {
          int q = 0;
          while (order[q] != j) ++q;
          ConvArgs a = as[q];
          unsigned fl = F_RES;
          if (j > 0) fl |= F_ACC;
          if (j == nk - 1) fl |= F_DIV;
          set_out(a.out[0], XS, bs, ld, C, fl);
          a.out[0].div = (float)nk;
          set_res(a.out[0], cur[j], bs, ld);
          int rw = pws[q] ? launch_conv_wino(*pws[q], a, B, 1, st) : 1;
          if (rw < 0) return rw;
          if (rw == 1) SVOC_TRY(launch_conv(*pcs[q], a, B, st));
        }
      }
    }
    return SVOC_OK;
  }

  int reserve(int B, int T) { return ws.ensure(workspace_bytes(B, T)); }

  int forward(hipStream_t st, const float* x, int x_ld, long long x_bs, const float* in_mask, long long in_mask_bs,
              const float* g, float* out, int B, int T) {
    const size_t sf = stage_floats(T);
    SVOC_TRY(ws.ensure(workspace_bytes(B, T)));
    const int nb = n_bufs();
    std::vector<float*> bufs(nb);
    for (int i = 0; i < nb; ++i) bufs[i] = ws.f() + (size_t)i * sf * B;
    float* gbias = ws.f() + (size_t)nb * sf * B;
    int ch = cfg.upsample_initial_channel;
    int L = T;
    int ld = stage_ld(L);
    // single-stream plan: 4 rotating buffers (R, X, XS, A; Bf reuses R).  multi-stream plan: P0/P1 ping-pong
    // for the stage results, one X, and (A, Bf) per chain.
    int r = use_streams ? 1 : 0;
    if (g) {
      if (!cond) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "generator: g given but gin_channels == 0");
      ConvArgs a = mk_args();
      set_in(a, g, cfg.gin_channels, 1, 1);
      a.Ncols = 1;
      set_out(a.out[0], gbias, ch, 1, ch);
      SVOC_TRY(launch_conv(*cond, a, B, st));
    }
    {   // conv_pre (+ cond(g)) (models.py:142-144)
      ConvArgs a = mk_args();
      set_in(a, x, x_bs, x_ld, T);
      a.in_mask = in_mask; a.in_mask_bs = in_mask_bs;
      a.Ncols = T;
      set_out(a.out[0], bufs[r], (long long)ch * ld, ld, ch);
      if (g) { a.gadd = gbias; a.gadd_bs = ch; a.gadd_ld = 1; a.gadd_ts = 0; }
      // k = 7, 192 -> 512 channels: the F(4,4) kernel issues half the direct form's MFMAs (114 -> ~55 us at 16 x 512); from half a tile per CU on
      int done = 1;
      if (conv_pre_w && !in_mask && !g) {
        done = launch_conv_wino(*conv_pre_w, a, B, 1, st, mrf_min_tiles());
        if (done < 0) return done;
      }
      if (done == 1) SVOC_TRY(launch_conv(conv_pre, a, B, st));
    }
    for (int i = 0; i < cfg.n_upsamples; ++i) {
      const int u = cfg.upsample_rates[i];
      float *R, *X, *XS;
      if (use_streams) { R = bufs[r]; XS = bufs[r ^ 1]; X = bufs[2]; }
      else { R = bufs[r]; X = bufs[(r + 1) & 3]; XS = bufs[(r + 2) & 3]; }
      const int Lo = L * u, ldo = stage_ld(Lo), cho = ch / 2;
      int ups_done = 1;
      if (ups_w[i]) {   // lrelu(0.1) -> ConvTranspose1d, Winograd F(4,2) over the polyphase filters (models.py:147-148)
        ups_done = launch_convt_wino(*ups_w[i], R, (long long)ch * ld, ld, 0.1f, X, (long long)cho * ldo, ldo, B, L, st);
        if (ups_done < 0) return ups_done;
      }
      if (ups_done == 1) {   // ... as the direct polyphase GEMM
        ConvArgs a = mk_args();
        set_in(a, R, (long long)ch * ld, ld, L);
        a.pre_slope = 0.1f;
        a.mode = EPI_UPS;
        a.Lout = Lo;
        a.Ncols = (Lo - 1 + ups[i]->ups_pad) / u + 1;
        set_out(a.out[0], X, (long long)cho * ldo, ldo, cho * u);
        SVOC_TRY(launch_conv(*ups[i], a, B, st));
      }
      const long long bs = (long long)cho * ldo;
      // Short inputs: when the three chains' convolutions are too small for the grouped Winograd launches (fewer than half a
      // workgroup per CU in total: mrf_min_tiles()) they would run one by one as K-split launches; the chains are independent until the
      // final accumulate, so they go to the three chain streams instead and their latencies overlap.
      bool small_stage = false;
      {
        static const bool on = !(getenv("SVOC_MRF_SMALL") && atoi(getenv("SVOC_MRF_SMALL")) == 0);
        const int mtl = cho / 32, wm = (mtl >= 4 && mtl % 4 == 0) ? 4 : 2;
        long long tiles = (long long)cfg.n_kernels * variant_batch(B) * ((Lo + (wm == 4 ? 63 : 127)) / (wm == 4 ? 64 : 128)) * ((mtl + wm - 1) / wm);
        if (mtl == 1) tiles = (long long)cfg.n_kernels * variant_batch(B) * ((Lo + 511) / 512);    // F(4,3), one row tile: 512 outputs per workgroup tile
        small_stage = on && (tiles < mrf_min_tiles() || (mtl == 1 && (Lo & 3)));
      }
      if (use_streams && !small_stage && mrf_grouped(i, cho)) {
        // MRF with the chains' step-i convolutions grouped into single launches (conv_group_kernel)
        SVOC_TRY(run_mrf_grouped(st, i, X, XS, bufs, bs, ldo, cho, B, Lo));
        r ^= 1;
        ch = cho; L = Lo; ld = ldo;
        continue;
      }
      if (use_streams) SVOC_HIP(hipEventRecord(ev_fork, st));
      for (int j = 0; j < cfg.n_kernels; ++j) {   // MRF: xs = sum_j ResBlock_j(x); x = xs / n (models.py:149-155)
        ResSink sink{XS, bs, ldo, 0u, 1.0f};
        if (j > 0) sink.flags |= F_ACC;
        if (j == cfg.n_kernels - 1) { sink.flags |= F_DIV; sink.div = (float)cfg.n_kernels; }
        if (use_streams) {
          SVOC_HIP(hipStreamWaitEvent(chain_st[j], ev_fork, 0));
          SVOC_TRY(rbs[i * cfg.n_kernels + j]->run(chain_st[j], X, bs, ldo, nullptr, 0, bufs[3 + 2 * j], bufs[4 + 2 * j], bs, ldo, sink, B, Lo,
                                                   j > 0 ? chain_done[j - 1] : nullptr, chain_done[j]));
        } else {
          SVOC_TRY(rbs[i * cfg.n_kernels + j]->run(st, X, bs, ldo, nullptr, 0, bufs[(r + 3) & 3], R, bs, ldo, sink, B, Lo));
        }
      }
      if (use_streams) { SVOC_HIP(hipStreamWaitEvent(st, chain_done[cfg.n_kernels - 1], 0)); r ^= 1; }
      else r = (r + 2) & 3;
      ch = cho; L = Lo; ld = ldo;
    }
    last = LastStage{bufs[r], ch, L, ld};
    return out ? post(st, last, out, B) : SVOC_OK;
  }
  // lrelu(0.01) -> conv_post -> tanh (models.py:156-158) on the last MRF stage left by a forward(); a captured plan keeps
  // its own copy of `last` (a replay does not run forward())
  struct LastStage { float* p = nullptr; int ch = 0, L = 0, ld = 0; };
  LastStage last;
  int post(hipStream_t st, const LastStage& ls, float* out, int B) {
    if (!ls.p) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "generator: post() without a forward()");
    return k_conv_post_tanh(st, ls.p, (long long)ls.ch * ls.ld, ls.ld, conv_post_w.f(), ls.ch, 7, 0.01f, out, B, ls.L);
  }
};

// =================================================================== PosteriorEncoder (models.py:83-112)
// Not on the infer path (training / voice conversion); built from the same kernels: pre 1x1 -> mask -> WN -> proj 1x1
// with the (m, logs, z = (m + eps*exp(logs))*mask) epilogue.
struct Posterior {
  int Cin = 0, Cout = 0, H = 0;
  PackedConv pre, proj;
  WNStack enc;
  DevBuf ws;

  // pre_name / enc_name: "pre" / "enc." for PosteriorEncoder (models.py:99-101), "pre_enc" / "encoder." for MelEncoder
  // (models.py:31-33)
  int create(int in_channels, int out_channels, int hidden, int k, int dr, int nl, int gin, const TensorTable& tab,
             const std::string& prefix, hipStream_t st, const char* pre_name = "pre", const char* enc_name = "enc.") {
    if (in_channels <= 0 || out_channels <= 0 || hidden <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "posterior encoder: bad configuration");
    Cin = in_channels; Cout = out_channels; H = hidden;
    PackSpec ps{}; ps.Cin = Cin; ps.Cout = H; ps.K = 1;
    SVOC_TRY(pack_conv_named(pre, ps, tab, prefix + pre_name, st));
    SVOC_TRY(enc.create(H, k, dr, nl, gin, tab, prefix + enc_name, st));
    PackSpec pj{}; pj.Cin = H; pj.Cout = 2 * Cout; pj.K = 1; pj.paired = true;
    SVOC_TRY(pack_conv_named(proj, pj, tab, prefix + "proj", st));
    return SVOC_OK;
  }

  size_t need(int B, int T) const { const int Tp = pad4(T); return (size_t)((2LL * H * Tp + Tp) * B) * sizeof(float); }

  // eps / z may be NULL (MelEncoder has no draw); x_out (nullable) receives the WN output [B][H][T] that
  // MelEncoder.forward returns first (models.py:42,47)
  int forward(hipStream_t st, const float* x, const int64_t* lengths, const float* g, int g_T, const float* eps, float* z, float* m,
              float* logs, float* x_mask, int B, int T, float* x_out = nullptr) {
    const int Tp = pad4(T);
    const long long hper = (long long)H * Tp;
    SVOC_TRY(ws.ensure(need(B, T)));
    float* xe = ws.f();
    float* eo = xe + hper * B;
    float* mask = eo + hper * B;
    SVOC_TRY(k_sequence_mask(st, lengths, mask, B, Tp));
    {   // x = pre(x) * x_mask (models.py:107)
      ConvArgs a = mk_args();
      set_in(a, x, (long long)Cin * T, T, T);
      a.Ncols = T; a.mask = mask; a.mask_bs = Tp;
      set_out(a.out[0], xe, hper, Tp, H, F_OUTMASK);
      SVOC_TRY(launch_conv(pre, a, B, st));
    }
    SVOC_TRY(enc.forward(st, xe, hper, Tp, mask, Tp, g, g_T, eo, hper, Tp, B, T));
    {   // stats = proj(x) * x_mask; z = (m + eps * exp(logs)) * x_mask (models.py:109-111)
      ConvArgs a = mk_args();
      set_in(a, eo, hper, Tp, T);
      a.Ncols = T; a.mask = mask; a.mask_bs = Tp;
      a.mode = EPI_PROJ;
      const long long ubs = (long long)Cout * T;
      set_out(a.out[0], m, ubs, T, Cout, F_OUTMASK);
      a.y2 = logs; a.y3 = z;
      a.eps = eps; a.eps_bs = ubs; a.eps_ld = T; a.noise_scale = 1.0f;
      SVOC_TRY(launch_conv(proj, a, B, st));
    }
    if (x_out) SVOC_TRY(k_copy2d(st, eo, hper, Tp, x_out, (long long)H * T, T, B, H, T, nullptr, 0));
    if (x_mask) SVOC_TRY(k_copy2d(st, mask, Tp, Tp, x_mask, T, T, B, 1, T, nullptr, 0));
    return SVOC_OK;
  }
};

// =================================================================== SynthesizerTrn.infer (models.py:331-339)
struct Synth {
  svoc_synth_config cfg{};
  PackedConv pre_enc, proj;
  WNStack enc;
  Flow flow;
  Generator dec;
  DevBuf ws;
  WeightPrefetch wn_prefetch[3];                            // the five WN stacks' weight images, per launch form (WNStack::weight_path)

  int create(const svoc_synth_config& c, const TensorTable& tab, hipStream_t st) {
    cfg = c;
    if (c.inter_channels <= 0 || c.hidden_channels <= 0 || c.n_mel <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "synth: bad configuration");
    PackSpec ps{}; ps.Cin = c.n_mel; ps.Cout = c.hidden_channels; ps.K = 1;
    SVOC_TRY(pack_conv_named(pre_enc, ps, tab, "enc_p.pre_enc", st));
    SVOC_TRY(enc.create(c.hidden_channels, c.enc_kernel_size, c.enc_dilation_rate, c.enc_n_layers, c.gin_channels, tab, "enc_p.encoder.", st));
    PackSpec pj{}; pj.Cin = c.hidden_channels; pj.Cout = 2 * c.inter_channels; pj.K = 1; pj.paired = true;
    SVOC_TRY(pack_conv_named(proj, pj, tab, "enc_p.proj", st));
    SVOC_TRY(flow.create(c.inter_channels, c.hidden_channels, c.flow_kernel_size, c.flow_dilation_rate, c.flow_n_layers, c.flow_n_flows,
                         c.gin_channels, tab, "flow.", st));
    SVOC_TRY(dec.create(c.dec, tab, "dec.", st));
    for (int path = 0; path < 3; ++path) {
      std::vector<std::pair<const void*, size_t>> bufs;
      enc.weight_buffers(path, bufs);
      for (auto* cp : flow.rev_p) cp->enc.weight_buffers(path, bufs);
      SVOC_TRY(wn_prefetch[path].build(bufs, st));
    }
    return SVOC_OK;
  }

  size_t own_bytes(int B, int T) const {
    const int Tp = pad4(T);
    return ((size_t)(2 * cfg.hidden_channels + 4 * cfg.inter_channels) * Tp + Tp) * B * sizeof(float);
  }
  int64_t workspace_bytes(int B, int T) const {
    return (int64_t)(own_bytes(B, T) + enc.need(B, T, 0) + (flow.rev_p.empty() ? 0 : flow.rev_p[0]->enc.need(B, T, 0)) * flow.NF +
                     (size_t)2 * cfg.hidden_channels * pad4(T) * B * sizeof(float) * flow.NF + dec.workspace_bytes(B, T));
  }
  // Sizes every workspace of the path for (B, T) up front, so that no later infer at that or a smaller shape allocates
  // (workspaces are grow-only; growing one frees and re-allocates it after a device synchronisation).
  int reserve(int B, int T) {
    SVOC_TRY(ws.ensure(own_bytes(B, T)));
    SVOC_TRY(enc.reserve(B, T, 0));
    SVOC_TRY(flow.reserve(B, T, 0, 1));
    return dec.reserve(B, T);
  }

  // ---- the launch plan proper.  `mel`, `lengths`, `eps` are read and nothing user-owned is written: the five small
  // user-visible tensors and the waveform are produced by tail() from library-owned buffers, so the same plan can be
  // replayed from a captured hipGraph with stable pointers.
  struct Bufs { float *xe, *eo, *mp, *lp, *P, *zp, *mask; };
  Bufs bufs(int B, int T) const {
    const int H = cfg.hidden_channels, IC = cfg.inter_channels, Tp = pad4(T);
    const long long hper = (long long)H * Tp, iper = (long long)IC * Tp;
    Bufs b;
    b.xe = ws.f();                    // masked pre_enc output
    b.eo = b.xe + hper * B;           // encoder (WN) output
    b.mp = b.eo + hper * B;           // m_p
    b.lp = b.mp + iper * B;           // logs_p
    b.P = b.lp + iper * B;            // z_p, then z (flows run in place)
    b.zp = b.P + iper * B;            // copy of z_p for the caller (flows overwrite P)
    b.mask = b.zp + iper * B;         // [B][Tp]
    return b;
  }

  int body(hipStream_t st, const float* mel, const int64_t* lengths, const float* eps, float noise_scale, int Td, bool want_zp,
           int B, int T) {
    const int H = cfg.hidden_channels, IC = cfg.inter_channels;
    const int Tp = pad4(T);
    const long long hper = (long long)H * Tp, iper = (long long)IC * Tp;
    SVOC_TRY(ws.ensure(own_bytes(B, T)));
    const Bufs w = bufs(B, T);
    SVOC_TRY(k_sequence_mask(st, lengths, w.mask, B, Tp));   // row stride Tp; entries >= T are never read
    // the WN stacks' weight images back into the memory-side cache (the decoder of the call before moved ~5 MB per frame through it): misc_kernels.hip
    SVOC_TRY(wn_prefetch[enc.weight_path(B, T)].run(st));
    {   // pre_enc 1x1, stored already multiplied by x_mask (models.py:38-42)
      ConvArgs a = mk_args();
      set_in(a, mel, (long long)cfg.n_mel * T, T, T);
      a.Ncols = T; a.mask = w.mask; a.mask_bs = Tp;
      set_out(a.out[0], w.xe, hper, Tp, H, F_OUTMASK);
      SVOC_TRY(launch_conv(pre_enc, a, B, st));
    }
    SVOC_TRY(enc.forward(st, w.xe, hper, Tp, w.mask, Tp, nullptr, 0, w.eo, hper, Tp, B, T));
    {   // proj -> (m_p, logs_p) * mask, reparameterisation z_p = m_p + eps*exp(logs_p)*noise_scale (models.py:44-46, 336)
      ConvArgs a = mk_args();
      set_in(a, w.eo, hper, Tp, T);
      a.Ncols = T; a.mask = w.mask; a.mask_bs = Tp;
      a.mode = EPI_PROJ;
      set_out(a.out[0], w.mp, iper, Tp, IC);
      a.y2 = w.lp; a.y3 = w.P;
      a.eps = eps; a.eps_bs = (long long)IC * T; a.eps_ld = T; a.noise_scale = noise_scale;
      SVOC_TRY(launch_conv(proj, a, B, st));
    }
    if (want_zp) SVOC_TRY(k_copy2d(st, w.P, iper, Tp, w.zp, iper, Tp, B, IC, T, nullptr, 0));
    if (flow.NF % 2) SVOC_FAIL(SVOC_ERR_UNSUPPORTED, "synth: odd n_flows is not supported on the fused path");
    SVOC_TRY(flow.run_inplace(st, w.P, iper, Tp, w.mask, Tp, nullptr, 0, 1, B, T));
    // dec((z * x_mask)[:, :, :max_len]) (models.py:338), up to the last MRF stage; conv_post runs in tail().  With two or more coupling layers BOTH
    // halves of z have been the x1 of a layer - x1 = (x1 - m) * x_mask, modules.py:341 - so z is masked already and z * x_mask is z bit for bit
    // (signed zeros included): the decoder is called without the mask, which lets conv_pre take the Winograd kernel.
    const bool z_is_masked = flow.NF >= 2;
    return dec.forward(st, w.P, Tp, iper, z_is_masked ? nullptr : w.mask, Tp, nullptr, nullptr, B, Td);
  }

  int tail(hipStream_t st, const Generator::LastStage& ls, float* o, float* x_mask, float* z, float* z_p, float* m_p, float* logs_p,
           int B, int T) {
    const int IC = cfg.inter_channels, Tp = pad4(T);
    const long long iper = (long long)IC * Tp, ubs = (long long)IC * T;
    const Bufs w = bufs(B, T);
    {   // the five small user-visible outputs in ONE launch (round 6; one launch each until then)
      CopyMany cm{};
      cm.cols = T; cm.B = B;
      auto add = [&](const float* src, long long s_bs, float* dst, long long d_bs, int rows) {
        if (!dst) return;
        const int k = cm.n++;
        cm.src[k] = src; cm.dst[k] = dst; cm.s_bs[k] = s_bs; cm.d_bs[k] = d_bs; cm.s_ld[k] = Tp; cm.d_ld[k] = T; cm.rows[k] = rows;
      };
      add(w.mp, iper, m_p, ubs, IC); add(w.lp, iper, logs_p, ubs, IC); add(w.zp, iper, z_p, ubs, IC); add(w.P, iper, z, ubs, IC); add(w.mask, Tp, x_mask, T, 1);
      const int rc = k_copy2d_many(st, cm);
      if (rc == SVOC_ERR_UNSUPPORTED) {                     // (a batch too large for the grid's z dimension: one launch per output)
        for (int k = 0; k < cm.n; ++k) SVOC_TRY(k_copy2d(st, cm.src[k], cm.s_bs[k], cm.s_ld[k], cm.dst[k], cm.d_bs[k], cm.d_ld[k], B, cm.rows[k], T, nullptr, 0));
      } else if (rc != SVOC_OK) return rc;
    }
    return dec.post(st, ls, o, B);
  }

  // ---- captured plans for short inputs.  At 1 x 200 frames the ~100 dependent launches of the path are a few tens of
  // microseconds each, so host launch cost and inter-kernel gaps are a sizeable part of the latency.  The body above is
  // captured once per (B, T, max_len, noise_scale, eps?, z_p?) into a hipGraph that reads its inputs from library-owned
  // staging buffers; a call then costs three small device-to-device copies, one graph launch and the tail.
  // identity of every workspace allocation the captured kernels point into: a plan is re-captured when one of them moved
  // (a larger shape grew it)
  unsigned long long ws_fingerprint() const {
    unsigned long long h = 1469598103934665603ull;
    auto mix = [&](const void* q) { h = (h ^ (unsigned long long)reinterpret_cast<uintptr_t>(q)) * 1099511628211ull; };
    mix(ws.p); mix(enc.ws.p); mix(dec.ws.p);
    for (auto* c : flow.rev_p) { mix(c->ws.p); mix(c->enc.ws.p); }
    mix(reinterpret_cast<const void*>((uintptr_t)persist_epoch()));      // the persistent launches' controls changed (a wait was given up: per-layer launches from now on)
    return h;
  }
  struct PlanKey {
    int B = 0, T = 0, Td = 0; float noise = 0; bool has_eps = false, want_zp = false;
    int vb = 0;                         // variant_batch(B) at capture: the kernel variants are frozen into the graph
    bool operator==(const PlanKey& o) const {
      return B == o.B && T == o.T && Td == o.Td && noise == o.noise && has_eps == o.has_eps && want_zp == o.want_zp && vb == o.vb;
    }
  };
  struct Plan {
    unsigned long long fp = 0;
    PlanKey key;
    hipGraphExec_t exec = nullptr;
    hipEvent_t done = nullptr;          // recorded behind every launch of `exec`: the exec is destroyed only once it has completed
    Generator::LastStage last;          // where the captured body leaves the last MRF stage
    long long conv_launches = 0, other_launches = 0, convs = 0; double conv_flops = 0, exec_flops = 0;
    unsigned long long last_use = 0;
    bool idle() const { return !done || hipEventQuery(done) == hipSuccess; }
    ~Plan() { if (exec) (void)hipGraphExecDestroy(exec); if (done) (void)hipEventDestroy(done); }
  };
  // A serving process sees hundreds of distinct utterance lengths below SVOC_GRAPH_MAX_FRAMES.  A shape earns a plan only
  // when it comes back: first sights go into a fixed table of counters (no allocation, nothing to evict), and a plan is
  // captured on the SVOC_GRAPH_MIN_SEEN-th call (default 2).  At most MAX_PLANS plans exist; the least recently used one
  // makes room, and it is destroyed without any device-wide synchronisation: a plan whose last launch has not completed
  // waits in `retired` (polled with hipEventQuery on later calls; only if more than MAX_RETIRED pile up does the oldest
  // get a hipEventSynchronize on ITS event).  All plans read their inputs from one staging area sized for
  // SVOC_GRAPH_MAX_FRAMES, allocated once, so that evicting a plan frees no device memory (hipFree synchronises).
  static constexpr int MAX_PLANS = 32, MAX_RETIRED = 8, N_COUNTERS = 64;
  struct Counter { PlanKey key; int seen = 0; unsigned long long last_use = 0; };
  Counter counters[N_COUNTERS];
  std::vector<std::unique_ptr<Plan>> plans, retired;
  DevBuf plan_stage;                    // mel | eps | lengths at fixed offsets
  unsigned long long use_clock = 0;
  long long plan_evictions = 0, plan_captures = 0, plan_sync_waits = 0;
  hipStream_t cap_st = nullptr;
  ~Synth() { plans.clear(); retired.clear(); if (cap_st) (void)hipStreamDestroy(cap_st); }

  static long long graph_max_frames() {
    static const long long v = getenv("SVOC_GRAPH_MAX_FRAMES") ? std::min(std::max(atoll(getenv("SVOC_GRAPH_MAX_FRAMES")), 0LL), 1LL << 22) : 32768;
    static const bool on = !(getenv("SVOC_GRAPH") && atoi(getenv("SVOC_GRAPH")) == 0);
    return on ? v : 0;
  }

  // staging area shared by all plans: lengths | mel | eps at offsets that depend only on SVOC_GRAPH_MAX_FRAMES
  struct Stage { int64_t* len; float* mel; float* eps; };
  int stage_ptrs(Stage& sp) {
    const size_t F = (size_t)graph_max_frames();
    const size_t len_b = (F * sizeof(int64_t) + 255) / 256 * 256;     // B <= B * T <= F entries
    SVOC_TRY(plan_stage.ensure(len_b + F * (size_t)(cfg.n_mel + cfg.inter_channels) * sizeof(float)));
    sp.len = reinterpret_cast<int64_t*>(plan_stage.p);
    sp.mel = reinterpret_cast<float*>(static_cast<char*>(plan_stage.p) + len_b);
    sp.eps = sp.mel + F * (size_t)cfg.n_mel;
    return SVOC_OK;
  }

  int capture(Plan& pl, hipStream_t st) {
    const int B = pl.key.B, T = pl.key.T;
    Stage sp;
    SVOC_TRY(stage_ptrs(sp));
    SVOC_TRY(reserve(B, T));
    if (!cap_st) SVOC_HIP(hipStreamCreateWithFlags(&cap_st, hipStreamNonBlocking));
    if (!pl.done) SVOC_HIP(hipEventCreateWithFlags(&pl.done, hipEventDisableTiming));
    long long cl0, ol0, cl1, ol1; double cf0, cf1;
    const long long nc0 = stats_convs();
    const double ef0 = stats_exec_flops();
    stats_get(&cl0, &cf0, &ol0);
    SVOC_HIP(hipStreamBeginCapture(cap_st, hipStreamCaptureModeThreadLocal));
    const int rc = body(cap_st, sp.mel, sp.len, pl.key.has_eps ? sp.eps : nullptr, pl.key.noise, pl.key.Td, pl.key.want_zp, B, T);
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(cap_st, &graph);
    stats_get(&cl1, &cf1, &ol1);
    const long long nc1 = stats_convs();
    const double ef1 = stats_exec_flops();
    stats_add_bulk(cl0 - cl1, cf0 - cf1, ol0 - ol1, nc0 - nc1, ef0 - ef1);          // the capture pass itself executed nothing
    if (rc != SVOC_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess || !graph) SVOC_FAIL(SVOC_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
    const hipError_t ei = hipGraphInstantiate(&pl.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess) { pl.exec = nullptr; SVOC_FAIL(SVOC_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ei)); }
    pl.conv_launches = cl1 - cl0; pl.conv_flops = cf1 - cf0; pl.other_launches = ol1 - ol0; pl.convs = nc1 - nc0; pl.exec_flops = ef1 - ef0;
    pl.fp = ws_fingerprint();
    pl.last = dec.last;
    ++plan_captures;
    (void)st;
    return SVOC_OK;
  }

  // Takes plan i out of service.  Its graph may still be executing: it is destroyed when its completion event has fired.
  void retire(size_t i) {
    std::unique_ptr<Plan> p = std::move(plans[i]);
    plans.erase(plans.begin() + i);
    ++plan_evictions;
    // an evicted shape has to earn its plan again (two more sights): with more hot shapes than MAX_PLANS every call would
    // otherwise evict + capture + instantiate, which costs more than direct launches
    for (auto& q : counters) if (q.seen > 0 && q.key == p->key) q.seen = 0;
    if (!p->idle()) retired.push_back(std::move(p));
  }
  void reap_retired() {
    for (size_t i = 0; i < retired.size();) {
      if (retired[i]->idle()) retired.erase(retired.begin() + i); else ++i;
    }
    while (retired.size() > (size_t)MAX_RETIRED) {        // bounded: wait for the OLDEST retired plan only (never the whole device)
      (void)hipEventSynchronize(retired.front()->done);
      ++plan_sync_waits;
      retired.erase(retired.begin());
    }
  }

  // The plan to replay for this call, or nullptr for direct launches.
  Plan* plan_for(const PlanKey& key, hipStream_t st) {
    static const int min_seen = getenv("SVOC_GRAPH_MIN_SEEN") ? std::max(2, atoi(getenv("SVOC_GRAPH_MIN_SEEN"))) : 2;
    if ((long long)key.B * key.T > graph_max_frames() || prof_enabled()) return nullptr;
    {   // a caller that is itself capturing (torch.cuda.graph around infer) gets plain launches: neither a nested capture
        // nor a graph launch is legal on a capturing stream
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
    }
    if (!retired.empty()) reap_retired();
    for (size_t i = 0; i < plans.size(); ++i)
      if (plans[i]->key == key) {
        Plan* pl = plans[i].get();
        pl->last_use = ++use_clock;
        if (pl->exec && pl->fp != ws_fingerprint()) {      // a workspace moved (grown by a larger shape): capture again
          retire(i);
          break;
        }
        return pl;
      }
    Counter* c = nullptr;
    Counter* lru = &counters[0];
    for (auto& q : counters) {
      if (q.seen != 0 && q.key == key) { c = &q; break; }
      if (q.last_use < lru->last_use) lru = &q;
    }
    if (!c) { c = lru; c->key = key; c->seen = 0; }
    c->last_use = ++use_clock;
    if (c->seen < 0) return nullptr;                       // capture failed for this shape before: direct launches
    if (++c->seen < min_seen) return nullptr;
    if (plans.size() >= (size_t)MAX_PLANS) {
      size_t v = 0;
      for (size_t i = 1; i < plans.size(); ++i) if (plans[i]->last_use < plans[v]->last_use) v = i;
      retire(v);
    }
    std::unique_ptr<Plan> np(new Plan());
    np->key = key;
    np->last_use = use_clock;
    if (capture(*np, st) != SVOC_OK) { c->seen = -1; return nullptr; }
    plans.push_back(std::move(np));
    return plans.back().get();
  }

  int infer(hipStream_t st, const float* mel, const int64_t* lengths, const float* eps, float noise_scale, int max_len, float* o,
            float* x_mask, float* z, float* z_p, float* m_p, float* logs_p, int B, int T) {
    SVOC_TRY(async_error_check());                          // (plan replays do not pass through WNStack::forward)
    const int Td = (max_len > 0 && max_len < T) ? max_len : T;
    const bool want_zp = z_p != nullptr;
    PlanKey key; key.B = B; key.T = T; key.Td = Td; key.noise = noise_scale; key.has_eps = eps != nullptr; key.want_zp = want_zp;
    key.vb = variant_batch(B);
    if (Plan* pl = plan_for(key, st)) {
      Stage sp;
      SVOC_TRY(stage_ptrs(sp));
      const size_t mel_n = (size_t)B * cfg.n_mel * T, eps_n = (size_t)B * cfg.inter_channels * T;
      SVOC_HIP(hipMemcpyAsync(sp.mel, mel, mel_n * sizeof(float), hipMemcpyDeviceToDevice, st));
      if (eps) SVOC_HIP(hipMemcpyAsync(sp.eps, eps, eps_n * sizeof(float), hipMemcpyDeviceToDevice, st));
      SVOC_HIP(hipMemcpyAsync(sp.len, lengths, (size_t)B * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
      SVOC_HIP(hipGraphLaunch(pl->exec, st));
      SVOC_HIP(hipEventRecord(pl->done, st));
      stats_add_bulk(pl->conv_launches, pl->conv_flops, pl->other_launches, pl->convs, pl->exec_flops);
      return tail(st, pl->last, o, x_mask, z, z_p, m_p, logs_p, B, T);
    }
    SVOC_TRY(body(st, mel, lengths, eps, noise_scale, Td, want_zp, B, T));
    return tail(st, dec.last, o, x_mask, z, z_p, m_p, logs_p, B, T);
  }
};

}  // namespace svoc

// ======================================================================= C ABI
using namespace svoc;

struct svoc_wn : svoc::HandleDevice { WNStack m; };
struct svoc_resblock : svoc::HandleDevice { ResBlock m; };
struct svoc_coupling : svoc::HandleDevice { Coupling m; };
struct svoc_flow : svoc::HandleDevice { Flow m; };
struct svoc_generator : svoc::HandleDevice { Generator m; };
struct svoc_synth : svoc::HandleDevice { Synth m; };
struct svoc_posterior : svoc::HandleDevice { Posterior m; };
struct svoc_mel_encoder : svoc::HandleDevice { Posterior m; };

#define SVOC_GUARD_BEGIN try {
#define SVOC_GUARD_END } catch (const std::exception& e) { ::svoc::set_error("exception: %s", e.what()); return SVOC_ERR_NOMEM; }

extern "C" {

int svoc_wn_create(svoc_wn** out, int hidden_channels, int kernel_size, int dilation_rate, int n_layers, int gin_channels,
                   const svoc_tensor* tensors, int n_tensors, const char* prefix) {
  if (!out || !tensors) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_wn_create: null argument");
  *out = nullptr;
  SVOC_GUARD_BEGIN
  std::unique_ptr<svoc_wn> h(new svoc_wn());
  TensorTable tab(tensors, n_tensors);
  SVOC_TRY(h->m.create(hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels, tab, prefix ? prefix : "", nullptr));
  *out = h.release();
  return SVOC_OK;
  SVOC_GUARD_END
}
int svoc_wn_forward(svoc_wn* h, void* stream, const float* x, const float* x_mask, const float* g, int g_T, float* out, int B, int T) {
  if (!h || !x || !x_mask || !out || B <= 0 || T <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_wn_forward: bad arguments");
  SVOC_GUARD_BEGIN
  const long long bs = (long long)h->m.H * T;
  return h->m.forward(as_stream(stream), x, bs, T, x_mask, T, g, g_T, out, bs, T, B, T);
  SVOC_GUARD_END
}
void svoc_wn_destroy(svoc_wn* h) { svoc::destroy_handle(h); }

int svoc_resblock_create(svoc_resblock** out, int kind, int channels, int kernel_size, const int* dilations, int n_dilations,
                         const svoc_tensor* tensors, int n_tensors, const char* prefix) {
  if (!out || !tensors || !dilations) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_resblock_create: null argument");
  *out = nullptr;
  SVOC_GUARD_BEGIN
  std::unique_ptr<svoc_resblock> h(new svoc_resblock());
  TensorTable tab(tensors, n_tensors);
  SVOC_TRY(h->m.create(kind, channels, kernel_size, dilations, n_dilations, tab, prefix ? prefix : "", nullptr));
  *out = h.release();
  return SVOC_OK;
  SVOC_GUARD_END
}
int svoc_resblock_forward(svoc_resblock* h, void* stream, const float* x, const float* x_mask, float* y, int B, int L) {
  if (!h || !x || !y || B <= 0 || L <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_resblock_forward: bad arguments");
  SVOC_GUARD_BEGIN
  return h->m.forward(as_stream(stream), x, x_mask, y, B, L);
  SVOC_GUARD_END
}
void svoc_resblock_destroy(svoc_resblock* h) { svoc::destroy_handle(h); }

int svoc_coupling_create(svoc_coupling** out, int channels, int hidden_channels, int kernel_size, int dilation_rate, int n_layers,
                         int gin_channels, int mean_only, const svoc_tensor* tensors, int n_tensors, const char* prefix) {
  if (!out || !tensors) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_coupling_create: null argument");
  *out = nullptr;
  SVOC_GUARD_BEGIN
  std::unique_ptr<svoc_coupling> h(new svoc_coupling());
  TensorTable tab(tensors, n_tensors);
  SVOC_TRY(h->m.create(channels, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels, mean_only, 0, tab, prefix ? prefix : "", nullptr));
  *out = h.release();
  return SVOC_OK;
  SVOC_GUARD_END
}
int svoc_coupling_forward(svoc_coupling* h, void* stream, const float* x, const float* x_mask, const float* g, int g_T, int reverse,
                          float* y, float* logdet, int B, int T) {
  if (!h || !x || !x_mask || !y || B <= 0 || T <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_coupling_forward: bad arguments");
  SVOC_GUARD_BEGIN
  hipStream_t st = as_stream(stream);
  Coupling& m = h->m;
  const long long bs = (long long)m.C * T;
  if (y != x) SVOC_TRY(k_copy2d(st, x, bs, T, y, bs, T, B, m.half, T, nullptr, 0));   // x0 passes through
  if (!reverse && logdet) SVOC_TRY(k_fill(st, logdet, (size_t)B, 0.0f));             // mean_only: logs == 0
  return m.run(st, x, bs, T, y, bs, T, x_mask, T, g, g_T, reverse, logdet, B, T);
  SVOC_GUARD_END
}
void svoc_coupling_destroy(svoc_coupling* h) { svoc::destroy_handle(h); }

int svoc_flow_create(svoc_flow** out, int channels, int hidden_channels, int kernel_size, int dilation_rate, int n_layers, int n_flows,
                     int gin_channels, const svoc_tensor* tensors, int n_tensors, const char* prefix) {
  if (!out || !tensors) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_flow_create: null argument");
  *out = nullptr;
  SVOC_GUARD_BEGIN
  std::unique_ptr<svoc_flow> h(new svoc_flow());
  TensorTable tab(tensors, n_tensors);
  SVOC_TRY(h->m.create(channels, hidden_channels, kernel_size, dilation_rate, n_layers, n_flows, gin_channels, tab, prefix ? prefix : "", nullptr));
  *out = h.release();
  return SVOC_OK;
  SVOC_GUARD_END
}
int svoc_flow_forward(svoc_flow* h, void* stream, const float* x, const float* x_mask, const float* g, int g_T, int reverse, float* y,
                      int B, int T) {
  if (!h || !x || !x_mask || !y || B <= 0 || T <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_flow_forward: bad arguments");
  SVOC_GUARD_BEGIN
  return h->m.forward(as_stream(stream), x, x_mask, g, g_T, reverse, y, B, T);
  SVOC_GUARD_END
}
void svoc_flow_destroy(svoc_flow* h) { svoc::destroy_handle(h); }

int svoc_generator_create(svoc_generator** out, const svoc_generator_config* cfg, const svoc_tensor* tensors, int n_tensors,
                          const char* prefix) {
  if (!out || !cfg || !tensors) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_generator_create: null argument");
  *out = nullptr;
  SVOC_GUARD_BEGIN
  std::unique_ptr<svoc_generator> h(new svoc_generator());
  TensorTable tab(tensors, n_tensors);
  SVOC_TRY(h->m.create(*cfg, tab, prefix ? prefix : "", nullptr));
  *out = h.release();
  return SVOC_OK;
  SVOC_GUARD_END
}
int svoc_generator_forward(svoc_generator* h, void* stream, const float* x, int x_ld, int64_t x_bs, const float* in_mask,
                           int64_t in_mask_bs, const float* g, float* out, int B, int T) {
  if (!h || !x || !out || B <= 0 || T <= 0 || x_ld < T) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_generator_forward: bad arguments");
  SVOC_GUARD_BEGIN
  return h->m.forward(as_stream(stream), x, x_ld, x_bs, in_mask, in_mask_bs, g, out, B, T);
  SVOC_GUARD_END
}
void svoc_generator_destroy(svoc_generator* h) { svoc::destroy_handle(h); }

int svoc_synth_create(svoc_synth** out, const svoc_synth_config* cfg, const svoc_tensor* tensors, int n_tensors) {
  if (!out || !cfg || !tensors) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_synth_create: null argument");
  *out = nullptr;
  SVOC_GUARD_BEGIN
  std::unique_ptr<svoc_synth> h(new svoc_synth());
  TensorTable tab(tensors, n_tensors);
  SVOC_TRY(h->m.create(*cfg, tab, nullptr));
  *out = h.release();
  return SVOC_OK;
  SVOC_GUARD_END
}
int svoc_synth_infer(svoc_synth* h, void* stream, const float* mel, const int64_t* lengths, const float* eps, float noise_scale,
                     int max_len, float* o, float* x_mask, float* z, float* z_p, float* m_p, float* logs_p, int B, int T) {
  if (!h || !mel || !lengths || !o || B <= 0 || T <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_synth_infer: bad arguments");
  if (!eps && noise_scale != 0.0f) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_synth_infer: eps is NULL but noise_scale != 0");
  SVOC_GUARD_BEGIN
  return h->m.infer(as_stream(stream), mel, lengths, eps, noise_scale, max_len, o, x_mask, z, z_p, m_p, logs_p, B, T);
  SVOC_GUARD_END
}
int64_t svoc_synth_workspace_bytes(svoc_synth* h, int B, int T) { return h ? h->m.workspace_bytes(B, T) : 0; }
int svoc_synth_reserve(svoc_synth* h, int B, int T) {
  if (!h || B <= 0 || T <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_synth_reserve: bad arguments");
  SVOC_GUARD_BEGIN
  return h->m.reserve(B, T);
  SVOC_GUARD_END
}
int svoc_synth_hop(svoc_synth* h) { return h ? h->m.dec.hop : 0; }
int svoc_synth_plan_stats(svoc_synth* h, int64_t* out5) {
  if (!h || !out5) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_synth_plan_stats: null argument");
  out5[0] = (int64_t)h->m.plans.size(); out5[1] = h->m.plan_captures; out5[2] = h->m.plan_evictions;
  out5[3] = h->m.plan_sync_waits; out5[4] = (int64_t)h->m.retired.size();
  return SVOC_OK;
}
void svoc_synth_destroy(svoc_synth* h) { svoc::destroy_handle(h); }

int svoc_posterior_create(svoc_posterior** out, int in_channels, int out_channels, int hidden_channels, int kernel_size,
                          int dilation_rate, int n_layers, int gin_channels, const svoc_tensor* tensors, int n_tensors, const char* prefix) {
  if (!out || !tensors) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_posterior_create: null argument");
  *out = nullptr;
  SVOC_GUARD_BEGIN
  std::unique_ptr<svoc_posterior> h(new svoc_posterior());
  TensorTable tab(tensors, n_tensors);
  SVOC_TRY(h->m.create(in_channels, out_channels, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels, tab,
                       prefix ? prefix : "", nullptr));
  *out = h.release();
  return SVOC_OK;
  SVOC_GUARD_END
}
int svoc_posterior_forward(svoc_posterior* h, void* stream, const float* x, const int64_t* lengths, const float* g, int g_T,
                           const float* eps, float* z, float* m, float* logs, float* x_mask, int B, int T) {
  if (!h || !x || !lengths || !eps || !z || !m || !logs || B <= 0 || T <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_posterior_forward: bad arguments");
  SVOC_GUARD_BEGIN
  return h->m.forward(as_stream(stream), x, lengths, g, g_T, eps, z, m, logs, x_mask, B, T);
  SVOC_GUARD_END
}
void svoc_posterior_destroy(svoc_posterior* h) { svoc::destroy_handle(h); }

int svoc_mel_encoder_create(svoc_mel_encoder** out, int n_mel, int out_channels, int hidden_channels, int kernel_size,
                            int dilation_rate, int n_layers, int gin_channels, const svoc_tensor* tensors, int n_tensors,
                            const char* prefix) {
  if (!out || !tensors) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_mel_encoder_create: null argument");
  *out = nullptr;
  SVOC_GUARD_BEGIN
  std::unique_ptr<svoc_mel_encoder> h(new svoc_mel_encoder());
  TensorTable tab(tensors, n_tensors);
  SVOC_TRY(h->m.create(n_mel, out_channels, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels, tab,
                       prefix ? prefix : "", nullptr, "pre_enc", "encoder."));
  *out = h.release();
  return SVOC_OK;
  SVOC_GUARD_END
}
int svoc_mel_encoder_forward(svoc_mel_encoder* h, void* stream, const float* x, const int64_t* lengths, float* x_out, float* m,
                             float* logs, float* x_mask, int B, int T) {
  if (!h || !x || !lengths || !m || !logs || B <= 0 || T <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_mel_encoder_forward: bad arguments");
  SVOC_GUARD_BEGIN
  // g is overwritten with None by the reference (models.py:36)
  return h->m.forward(as_stream(stream), x, lengths, nullptr, 0, nullptr, nullptr, m, logs, x_mask, B, T, x_out);
  SVOC_GUARD_END
}
void svoc_mel_encoder_destroy(svoc_mel_encoder* h) { svoc::destroy_handle(h); }

// ---- diagnostics: device buffer ([workgroup][8] int64, zeroed by the caller) that resblock_fused_kernel fills with its
// phase cycle stamps; NULL switches the stamps off
int svoc_debug_set_stamp_buffer(void* buf) { set_debug_stamp_buffer(static_cast<long long*>(buf)); return SVOC_OK; }

// ---- diagnostics: phase timing of one convolution launch (cycle stamps per workgroup)
int svoc_debug_conv_timing(void* stream, const float* x, const float* weight, const float* bias, const float* residual, float* y,
                           int B, int C, int L, int kernel_size, int dilation, double* out4) {
  if (!x || !weight || !y || !out4) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_debug_conv_timing: bad arguments");
  SVOC_GUARD_BEGIN
  hipStream_t st = as_stream(stream);
  PackedConv pc;
  PackSpec sp{}; sp.Cin = C; sp.Cout = C; sp.K = kernel_size; sp.dil = dilation;
  SVOC_TRY(pack_conv(pc, sp, weight, nullptr, bias, st));
  const size_t maxblocks = 1 << 20;
  DevBuf dbg;
  SVOC_TRY(dbg.ensure(maxblocks * 4 * sizeof(long long)));
  SVOC_HIP(hipMemsetAsync(dbg.p, 0, maxblocks * 4 * sizeof(long long), st));
  ConvArgs a = mk_args();
  set_in(a, x, (long long)C * L, L, L);
  a.pre_slope = 0.1f;
  a.Ncols = L;
  a.dbg = (long long*)dbg.p;
  a.dbg_wall = getenv("SVOC_DBG_WALL") && atoi(getenv("SVOC_DBG_WALL")) != 0;   // phases in 10 ns units
  set_out(a.out[0], y, (long long)C * L, L, C, residual ? (unsigned)F_RES : 0u);
  if (residual) set_res(a.out[0], residual, (long long)C * L, L);
  for (int it = 0; it < 2; ++it) SVOC_TRY(launch_conv(pc, a, B, st));
  SVOC_HIP(hipStreamSynchronize(st));
  std::vector<long long> h(maxblocks * 4);
  SVOC_HIP(hipMemcpy(h.data(), dbg.p, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
  double s0 = 0, s1 = 0, s2 = 0; long long n = 0, tmin = 0, tmax = 0;
  if (getenv("SVOC_DBG_DUMP")) {   // two records from each of SVOC_DBG_DUMP sections (roles of the persistent kernels)
    const int nsec = std::max(1, atoi(getenv("SVOC_DBG_DUMP")));
    std::vector<size_t> idx;
    for (size_t i = 0; i < maxblocks; ++i) if (h[4 * i + 3] != 0) idx.push_back(i);
    for (int sct = 0; sct < nsec; ++sct)
      for (size_t j = 0; j < 2; ++j) {
        const size_t q = idx.size() * sct / nsec + j;
        if (q >= idx.size()) continue;
        const long long* d = &h[4 * idx[q]];
        fprintf(stderr, "dbg[%zu] %lld %lld %lld\n", idx[q], d[1] - d[0], d[2] - d[1], d[3] - d[2]);
      }
  }
  for (size_t i = 0; i < maxblocks; ++i) {
    const long long* d = &h[4 * i];
    if (d[3] == 0) continue;
    s0 += (double)(d[1] - d[0]); s1 += (double)(d[2] - d[1]); s2 += (double)(d[3] - d[2]);
    if (n == 0 || d[0] < tmin) tmin = d[0];
    if (n == 0 || d[3] > tmax) tmax = d[3];
    ++n;
  }
  out4[0] = n ? s0 / n : 0; out4[1] = n ? s1 / n : 0; out4[2] = n ? s2 / n : 0; out4[3] = (double)(tmax - tmin);
  return SVOC_OK;
  SVOC_GUARD_END
}

// ---- single ops
int svoc_flip_channels(void* stream, const float* x, float* y, int B, int C, int T) {
  if (!x || !y || B <= 0 || C <= 0 || T <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_flip_channels: bad arguments");
  return k_flip_copy(as_stream(stream), x, (long long)C * T, T, y, (long long)C * T, T, B, C, T);
}
int svoc_fold_weight_norm(void* stream, const float* weight_v, const float* weight_g, float* weight, int64_t d0, int64_t inner) {
  if (!weight_v || !weight_g || !weight || d0 <= 0 || inner <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_fold_weight_norm: bad arguments");
  SVOC_GUARD_BEGIN
  return fold_weight_norm(as_stream(stream), weight_v, weight_g, weight, d0, inner);
  SVOC_GUARD_END
}
int svoc_conv1d(void* stream, const float* x, const float* weight_v, const float* weight_g, const float* bias, const float* residual,
                float* y, int B, int Cin, int Cout, int L, int kernel_size, int dilation, float pre_slope) {
  if (!x || !weight_v || !y || B <= 0 || L <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_conv1d: bad arguments");
  SVOC_GUARD_BEGIN
  hipStream_t st = as_stream(stream);
  PackedConv pc;
  PackSpec sp{}; sp.Cin = Cin; sp.Cout = Cout; sp.K = kernel_size; sp.dil = dilation;
  SVOC_TRY(pack_conv(pc, sp, weight_v, weight_g, bias, st));
  ConvArgs a = mk_args();
  set_in(a, x, (long long)Cin * L, L, L);
  a.pre_slope = pre_slope;
  a.Ncols = L;
  set_out(a.out[0], y, (long long)Cout * L, L, Cout, residual ? (unsigned)F_RES : 0u);
  if (residual) set_res(a.out[0], residual, (long long)Cout * L, L);
  SVOC_TRY(launch_conv(pc, a, B, st));
  SVOC_HIP(hipStreamSynchronize(st));   // packed weights are freed on return
  return SVOC_OK;
  SVOC_GUARD_END
}
int svoc_conv1d_winograd(void* stream, const float* x, const float* weight_v, const float* weight_g, const float* bias,
                         const float* residual, float* y, int B, int Cin, int Cout, int L, int kernel_size, int dilation,
                         float pre_slope) {
  if (!x || !weight_v || !y || B <= 0 || L <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_conv1d_winograd: bad arguments");
  if (!wino_supported(Cin, Cout, kernel_size, dilation))
    SVOC_FAIL(SVOC_ERR_UNSUPPORTED, "svoc_conv1d_winograd: kernel 3/7/11, dilation 1/3/5, channels multiples of 32, Cin >= 64");
  SVOC_GUARD_BEGIN
  hipStream_t st = as_stream(stream);
  PackedWino pw;
  SVOC_TRY(pack_wino(pw, Cin, Cout, kernel_size, weight_v, weight_g, bias, st));
  ConvArgs a = mk_args();
  set_in(a, x, (long long)Cin * L, L, L);
  a.pre_slope = pre_slope;
  a.Ncols = L;
  set_out(a.out[0], y, (long long)Cout * L, L, Cout, residual ? (unsigned)F_RES : 0u);
  if (residual) set_res(a.out[0], residual, (long long)Cout * L, L);
  const int r = launch_conv_wino(pw, a, B, dilation, st, 0);
  if (r == 1) SVOC_FAIL(SVOC_ERR_UNSUPPORTED, "svoc_conv1d_winograd: needs 16-byte aligned rows (L %% 4 == 0)");
  if (r < 0) return r;
  SVOC_HIP(hipStreamSynchronize(st));   // transformed weights are freed on return
  return SVOC_OK;
  SVOC_GUARD_END
}
int svoc_conv_transpose1d(void* stream, const float* x, const float* weight_v, const float* weight_g, const float* bias, float* y, int B,
                          int Cin, int Cout, int L, int kernel_size, int stride, float pre_slope) {
  if (!x || !weight_v || !y || B <= 0 || L <= 0 || stride <= 0 || kernel_size < stride || ((kernel_size - stride) % 2))
    SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_conv_transpose1d: bad arguments");
  SVOC_GUARD_BEGIN
  hipStream_t st = as_stream(stream);
  const int Lo = L * stride;
  if (convt_wino_supported(Cin, Cout, kernel_size, stride, (kernel_size - stride) / 2)) {
    PackedCtWino pw;
    SVOC_TRY(pack_convt_wino(pw, Cin, Cout, kernel_size, stride, (kernel_size - stride) / 2, weight_v, weight_g, bias, st));
    const int r = launch_convt_wino(pw, x, (long long)Cin * L, L, pre_slope, y, (long long)Cout * Lo, Lo, B, L, st);
    if (r < 0) return r;
    if (r == 0) { SVOC_HIP(hipStreamSynchronize(st)); return SVOC_OK; }
  }
  PackedConv pc;
  PackSpec sp{}; sp.Cin = Cin; sp.Cout = Cout; sp.K = kernel_size; sp.transposed = true; sp.stride = stride; sp.tpad = (kernel_size - stride) / 2;
  SVOC_TRY(pack_conv(pc, sp, weight_v, weight_g, bias, st));
  ConvArgs a = mk_args();
  set_in(a, x, (long long)Cin * L, L, L);
  a.pre_slope = pre_slope;
  a.mode = EPI_UPS;
  a.Lout = Lo;
  a.Ncols = (Lo - 1 + pc.ups_pad) / stride + 1;
  set_out(a.out[0], y, (long long)Cout * Lo, Lo, Cout * stride);
  SVOC_TRY(launch_conv(pc, a, B, st));
  SVOC_HIP(hipStreamSynchronize(st));
  return SVOC_OK;
  SVOC_GUARD_END
}

}  // extern "C"
