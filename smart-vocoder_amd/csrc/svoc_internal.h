// Internal declarations shared by the HIP translation units of libsvoc_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <map>
#include <memory>

#include "../../include/svoc.h"

namespace svoc {

// ------------------------------------------------------------------ errors
void set_error(const char* fmt, ...);
#define SVOC_FAIL(code, ...) do { ::svoc::set_error(__VA_ARGS__); return (code); } while (0)
#define SVOC_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    ::svoc::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); return SVOC_ERR_HIP; } } while (0)
#define SVOC_TRY(expr) do { int r_ = (expr); if (r_ != SVOC_OK) return r_; } while (0)

// ------------------------------------------------------------------ device buffers
struct DevBuf {
  void* p = nullptr; size_t bytes = 0;
  DevBuf() = default; DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int ensure(size_t n) {           // grow-only
    if (n <= bytes) return SVOC_OK;
    if (p) { (void)hipDeviceSynchronize(); (void)hipFree(p); p = nullptr; bytes = 0; }
    hipError_t e = hipMalloc(&p, n);
    if (e != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", n, hipGetErrorString(e)); p = nullptr; return SVOC_ERR_NOMEM; }
    bytes = n; return SVOC_OK;
  }
  float* f() const { return (float*)p; }
};

// ------------------------------------------------------------------ state-dict table
struct TensorTable {
  std::map<std::string, const svoc_tensor*> m;
  TensorTable(const svoc_tensor* t, int n) { for (int i = 0; i < n; ++i) if (t[i].name) m[t[i].name] = &t[i]; }
  const svoc_tensor* find(const std::string& name) const { auto it = m.find(name); return it == m.end() ? nullptr : it->second; }
};

// ------------------------------------------------------------------ implicit-GEMM convolution
constexpr int KC = 32;           // input channels staged in LDS per chunk

enum EpiFlags : unsigned {
  F_RES      = 1u << 0,   // val += res[row][col]
  F_ACC      = 1u << 1,   // val = y_old + val
  F_DIV      = 1u << 2,   // val = val / div
  F_OUTMASK  = 1u << 3,   // val *= mask[col]
  F_CPL_REV  = 1u << 4,   // val = (res - val*mask) * mask              (mean-only coupling, reverse)
  F_CPL_FWD  = 1u << 5,   // val = val*mask + res*mask                   (mean-only coupling, forward)
  F_LOGCLAMP = 1u << 6,   // val = log(max(val, log_clamp))               (mel_processing.py:19-25)
};
enum EpiMode : int {
  EPI_PLAIN = 0,          // per-element flags above, optional row split
  EPI_UPS = 1,            // polyphase ConvTranspose1d scatter: row=(o*s+r), col=m -> y[o][m*s+r-pad]
  EPI_GATE = 2,           // tile pairs: y = tanh(v0+g0) * sigmoid(v1+g1)
  EPI_PROJ = 3,           // tile pairs: m=v0*mask, logs=v1*mask, z_p = m + eps*exp(logs)*noise
  EPI_CPL_FULL_REV = 4,   // tile pairs: x1 = (x1 - v0*mask) * exp(-(v1*mask)) * mask
  EPI_MAG = 6,            // tile pairs (re, im): y = sqrt(v0^2 + v1^2 + mag_eps)  (mel_processing.py:69)
  EPI_CPL_FULL_FWD = 5,   // tile pairs: x1 = v0*mask + x1*exp(v1*mask)*mask ; logdet += sum(v1*mask)
};

struct EpiOut {
  float* y; long long y_bs; int y_ld; int nrows;     // nrows: valid rows of this output set
  const float* res; long long res_bs; int res_ld;
  unsigned flags; float div;
};

struct ConvArgs {
  // input activations [B][Cin][x_ld]; positions outside [0,Lin) read as zero
  const float* x; long long x_bs; int x_ld; int Cin; int Lin;
  const float* in_mask; long long in_mask_bs;      // optional multiplier [B][>=Lin], applied after the activation
  float pre_slope;                                   // leaky-relu slope applied while staging (1 = none)
  int vec4;                                          // 1: rows are 16-byte aligned -> float4 staging loads
  int xcd;                                           // 1: XCD-aware tile order (xcd_linear)
  // packed weights / bias
  const float* wp; const float* bias;
  int nchunks; int kcs; int ktaps; int dil; int pad;   // kcs: channels staged per round trip (multiple of 32);          // tap j reads x[n + j*dil - pad]
  int mtiles;                                        // valid 32-row tiles
  int ksg_total;                                     // groups of 4 k-steps per m-tile
  int xoff0; int row_len;                            // LDS tile: starts at n0+xoff0 (multiple of 4), row_len floats
  int Ncols;                                         // output columns
  int ntn; int B;                                    // time tiles per batch element, batch size (persistent and grouped kernels)
  int gy;                                            // row blocks per time tile (grouped kernel)
  long long* dbg;                                    // optional [nblocks][4] cycle stamps (diagnostics)
  int dbg_wall;                                      // stamps from the 100 MHz wall clock instead of the shader clock
  // epilogue
  int mode;
  const float* mask; long long mask_bs;              // [B][>=Ncols] output-side mask
  int split_row;                                     // rows >= split_row use out[1] (row index rebased)
  EpiOut out[2];
  const float* gadd; long long gadd_bs; int gadd_ld; int gadd_ts;   // per-(batch,row[,col]) additive term (g conditioning)
  int half_rows;                                     // paired modes: channels per half (H)
  int ups_s; int ups_pad; int Lout;                  // EPI_UPS
  const float* eps; long long eps_bs; int eps_ld; float noise_scale;  // EPI_PROJ
  float* y2; float* y3;                              // EPI_PROJ: logs_p, z_p (same strides as out[0])
  float* logdet;                                     // EPI_CPL_FULL_FWD
  float mag_eps; float log_clamp;                    // EPI_MAG / F_LOGCLAMP
  // F(4,3) grouped launches only (conv_wino4.hip, round 4): a dilated convolution writes its rows window-major (wperm_out = its
  // dilation) with 16-byte stores and the undilated convolution behind it reads them through the same map (wperm_in)
  int wperm_out; int wperm_in;
};

// One convolution layer repacked for the MFMA kernel.
struct PackedConv {
  DevBuf wp, bias;
  int Cin = 0, CinP = 0, Cout = 0, rows = 0, mtiles = 0, ktaps = 0, dil = 1, pad = 0;
  int ksg_total = 0;
  bool paired = false; int half_rows = 0;
  int split_row = 1 << 30;        // first packed row of the second row group (PackSpec::split_at)
  bool transposed = false; int ups_s = 1, ups_pad = 0;
  double flops_per_col = 0;       // algorithmic 2*MAC per output column (per batch element)
};

struct PackSpec {
  int Cin, Cout, K;               // source tensor dims (Conv1d: [Cout][Cin][K]; ConvTranspose1d: [Cin][Cout][K])
  int dil = 1;                    // Conv1d dilation
  int pad = -1;                   // -1: "same" padding (K*dil-dil)/2
  bool transposed = false; int stride = 1; int tpad = 0;    // ConvTranspose1d(stride, padding)
  bool paired = false;            // rows [0,Cout/2) and [Cout/2,Cout) interleaved tile-wise
  int split_at = 0;               // >0: rows [0,split_at) and [split_at,Cout) each padded to a multiple of 32
  const int* out_perm = nullptr;  // packed logical out-channel o reads source channel out_perm[o]
  const int* in_perm = nullptr;   // packed in-channel c reads source channel in_perm[c]
};

int pack_conv(PackedConv& pc, const PackSpec& spec, const float* w_or_v, const float* g, const float* bias, hipStream_t st);
int pack_conv_named(PackedConv& pc, PackSpec spec, const TensorTable& tab, const std::string& prefix, hipStream_t st,
                    bool bias_required = true);

int fold_weight_norm(hipStream_t st, const float* v, const float* g, float* w, long long d0, long long inner);

// Fills geometry fields (wp, bias, taps, tiles, LDS tile) of `a` from `pc`; caller sets x/epilogue fields first.
int launch_conv(const PackedConv& pc, ConvArgs a, int B, hipStream_t st);

// XCD-aware tile order.  Workgroups are handed to the 8 XCDs round-robin by linear id, and each XCD has its own L2:
// in the natural order the time tiles t and t+1 (which share a (k-1)*dilation halo of up to 45 % of a tile) and the
// row blocks of one tile (which share the whole input tile) always land on different XCDs and are fetched from HBM
// once per XCD.  xcd_linear() turns workgroup id `lin` of `total` into a tile id such that XCD c walks the
// contiguous range [start(c), start(c+1)): neighbours are issued back to back on the same L2.  Bijective for any
// total.  on = 0 keeps the natural order (SVOC_XCD=0, for A/B runs).
#if defined(__HIPCC__)
// Gate of a WN layer (reference commons.py:100-107): tanh(a) * sigmoid(b) from the hardware exp2 / rcp (1 ulp each) - 9
// instructions where tanhf + expf + an IEEE division take ~60, and the gate runs on eight accumulator registers per lane
// and layer between two MFMA phases.  tanh(a) = 1 - 2 / (1 + e^{2a}): absolute error <= ~2e-7, saturates correctly.
__device__ __forceinline__ float gate_tanh_sigmoid(float a, float b) {
  const float ea = __builtin_amdgcn_exp2f(a * 2.885390082f);      // e^{2a}
  const float eb = __builtin_amdgcn_exp2f(b * -1.442695041f);     // e^{-b}
  const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + ea);
  return t * __builtin_amdgcn_rcpf(1.0f + eb);
}
__device__ __forceinline__ int xcd_linear(int lin, int total, int on) {
  if (!on) return lin;
  const int c = lin & 7, i = lin >> 3;
  const int q = total >> 3, r = total & 7;
  return c * q + min(c, r) + i;
}
#endif
int xcd_mapping_enabled();   // misc_kernels.hip
// Batch size that kernel-VARIANT choices are made for (misc_kernels.hip).  Tile shapes, K split, Winograd-vs-direct, fused-
// vs-unfused WN layers and the MRF launch plan are picked from the launch size, and different variants sum in different
// orders, so an utterance alone and inside a batch agree to fp32 rounding but not bit for bit.  svoc_set_variant_batch(n)
// makes every such choice as if the batch were n (0 = the real batch): ranks that each run a shard of an n-utterance job
// then produce exactly the bits a single process produces for the whole job.  Grid sizes still follow B.
int variant_batch(int B);
// Per-device launch prerequisites (misc_kernels.hip).  A process may drive several GPUs (one handle per device), so
// neither the compute-unit count nor "this kernel may use 160 KiB of dynamic LDS" can live in a function-local static.
int device_cu_count();                       // compute units of the CURRENT device (cached per device)
long long mrf_min_tiles();                   // fewest tiles for the grouped / paired / accumulate Winograd launches of the MRF
int* async_error_word();                     // pinned host word the persistent launches raise when a bounded wait gives up (null: allocation failed)
int async_error_check();                     // SVOC_OK, or fails (once) when that word was raised since the last check
// the persistent launches' run-time controls (misc_kernels.hip): bound of their waits in ticks of the 100 MHz wall counter (SVOC_PERSIST_TIMEOUT_MS,
// default 2000), the tile whose workgroup withholds its flags (diagnostics: -1), "a wait was given up: per-layer launches from now on", a counter
// that moves when any of them changes (captured plans are captured again), and what the occupancy calculator says a kernel's residency is
unsigned long long persist_timeout_ticks();
int persist_timeout_ms();
int persist_fault_tile();
bool persist_disabled();
unsigned persist_epoch();
int persist_control(int fault_tile, int timeout_ms, int reenable);
int persist_capacity(const void* kernel, int threads, size_t lds_bytes);
int ensure_max_dyn_lds(const void* kernel);  // hipFuncSetAttribute(MaxDynamicSharedMemorySize, 160 KiB) once per (kernel, device)

struct ConvGroup { ConvArgs a[3]; int end[3]; };     // conv_group_kernel: end[i] = first workgroup id after problem i
int launch_conv_group(const PackedConv* const* pcs, const ConvArgs* as, int n, int B, hipStream_t st);   // 1 = not eligible

// conv_wino.hip: Winograd F(2,3) form of the big convolutions (k = 3 / 7 / 11, dilation 1 / 3 / 5, C >= 64): 1/3 fewer MFMAs
struct PackedWino {
  DevBuf wp, bias;
  DevBuf wp4;                     // F(4,3) image (conv_wino4.hip), present when that form is enabled and the shape is eligible
  bool f44 = false;               // wp4 is in F(4,4) form (conv_wino4.h: k = 7 / 11 of the 128-row layout)
  DevBuf wp44;                    // k = 3 of the 128-row layout: second image in F(4,4) form (the merged accumulate launch, whose members share accumulators)
  DevBuf wraw;                    // plain weights [Cout][Cin * K], weight norm applied: the dilated launches' row tails (conv_wino4.hip: conv_wino4_tail_kernel)
  int Cin = 0, Cout = 0, K = 0, nchunks = 0, mtiles = 0, slots = 0;
  double flops_per_col = 0;       // algorithmic 2*MAC of the convolution per output column
};
bool wino_supported(int Cin, int Cout, int K, int dil);
int pack_wino(PackedWino& pw, int Cin, int Cout, int K, const float* w_or_v, const float* g, const float* bias, hipStream_t st);
int pack_wino_named(PackedWino& pw, int Cin, int Cout, int K, const TensorTable& tab, const std::string& prefix, hipStream_t st);
// `a`: x / pre_slope / Ncols / out[0] as for launch_conv (plain epilogue, flags within F_RES | F_ACC | F_DIV); 1 = not
// eligible (alignment, odd length, masks, or fewer than min_tiles workgroups; min_tiles < 0: twice the CU count)
int launch_conv_wino(const PackedWino& pw, const ConvArgs& a, int B, int dil, hipStream_t st, long long min_tiles = -1);
// query = true: launches nothing and returns 0 iff the F(4,3) kernel would run (the only form that honours wperm_in / wperm_out)
int launch_conv_wino_group(const PackedWino* const* pws, const ConvArgs* as, int n, int B, int dil, hipStream_t st, bool query = false);
int launch_conv_wino4_accum(const PackedWino* const* pws, const ConvArgs* as, int B, hipStream_t st, bool query = false);   // chain order k = 3, 7, 11

// conv_wino4_pair.hip: c1 (dilation D1) -> c2 of one ResBlock1 iteration at C = 32 / 64 in one launch for the three chains (members
// k = 11, 7, 3; y != x); 1 = not eligible
bool wino4_c32_enabled();
bool wino4_pair_enabled();
long long wino4_pair_tiles(int C, int L, int B, int D1);
int launch_wino4_pair(const PackedWino* const* pw1, const PackedWino* const* pw2, const float* const* x, float* const* y, long long bs, int ld,
                      int B, int L, int D1, float slope, hipStream_t st, int accum = 0, float div = 1.0f);

// convt_wino.hip: Winograd F(4,2) form of the polyphase transposed convolution (upsamplers k = 2 s: s = 8 or 2)
struct PackedCtWino {
  DevBuf wp, bias;
  int Cin = 0, Cout = 0, S = 0, tpad = 0, nchunks = 0, mtiles = 0;
  double flops_per_col = 0;       // algorithmic 2*MAC per INPUT column
};
bool convt_wino_supported(int Cin, int Cout, int K, int stride, int tpad);
int pack_convt_wino(PackedCtWino& pw, int Cin, int Cout, int K, int stride, int tpad, const float* w_or_v, const float* g,
                    const float* bias, hipStream_t st);
int pack_convt_wino_named(PackedCtWino& pw, int Cin, int Cout, int K, int stride, int tpad, const TensorTable& tab,
                          const std::string& prefix, hipStream_t st);
// 1 = not eligible (alignment, or fewer workgroups than half the CUs)
int launch_convt_wino(const PackedCtWino& pw, const float* x, long long x_bs, int x_ld, float pre_slope, float* y, long long y_bs,
                      int y_ld, int B, int Lin, hipStream_t st);

// resblock_fused.hip: one ResBlock1 iteration (c1 -> lrelu -> c2 -> + x) in one kernel; returns 1 when not eligible
void set_debug_stamp_buffer(long long* p);
long long* debug_stamp_buffer();   // misc_kernels.hip: device buffer for per-workgroup cycle stamps, or nullptr
int launch_resblock_fused(const PackedConv& c1, const PackedConv& c2, const float* x, long long x_bs, int x_ld, float* y,
                          long long y_bs, int y_ld, unsigned flags, float div, int B, int L, hipStream_t st);
bool resblock_fused_ct_supported(int C, int k, int dil);   // shapes served by the compile-time-specialised kernel

// wn_fused.hip: one WN layer (in_layer -> gate -> res_skip -> residual/skip) in one kernel; returns 1 when not eligible
int launch_wn_layer_fused(const PackedConv& in_l, const PackedConv& rs_l, int H, const float* x, long long x_bs, int x_ld,
                          float* xo, long long xo_bs, int xo_ld, float* out, long long out_bs, int out_ld, const float* mask,
                          long long mask_bs, const float* gadd, long long gadd_bs, int gadd_ld, int gadd_ts, int first, int last,
                          int B, int T, hipStream_t st, const float* wpf = nullptr);
// F(2,5) image of an in_layer (H = 192, k = 5, dilation 1; wn_layer_f25_kernel), left empty when the form does not apply
bool wn_f25_enabled();
// wn_stack.hip: a whole WN stack in one persistent launch (neighbour-to-neighbour edge exchange between the layers)
size_t wn_stack_scratch_bytes();
bool wn_stack_applies(int H, int K, int dil_rate, int NL, int B, int T);
int wn_stack_groups(int H, int K, int dil_rate, int NL, int B, int T);      // > 1: that many launches over equal groups of utterances (batches beyond the launch's capacity)
int launch_wn_stack_f25(const PackedConv* const* in_l, const PackedConv* const* rs_l, const float* const* wpf, int NL, int H, const float* x, long long x_bs,
                        int x_ld, float* out, long long out_bs, int out_ld, const float* mask, long long mask_bs, float* scratch, int B, int T, hipStream_t st, bool first_group = true);
int wn_stack_prepare(float* scratch, const PackedConv* const* in_l, const PackedConv* const* rs_l, const float* const* wpf, int NL, hipStream_t st);
bool wn_layer_prefers_unfused(int B, int T);              // short inputs: fewer 32-column tiles than half the CUs (wn_fused.hip's size gate)
// wn_mesh.hip: a whole WN stack for SHORT inputs in one persistent launch (twelve workgroups per 32-column tile, two hand-overs per layer)
bool wn_mesh_enabled();
size_t wn_mesh_scratch_bytes();
bool wn_mesh_supported(int H, int K, int dil_rate, int NL);
bool wn_mesh_applies(int H, int K, int dil_rate, int NL, int B, int T);
int pack_wn_mesh(DevBuf& img, const float* f25, hipStream_t st);
int wn_mesh_prepare(float* scratch, const PackedConv* const* in_l, const float* const* wm, const float* const* wrs, int NL, hipStream_t st);
int launch_wn_mesh_f25(const PackedConv* const* in_l, const PackedConv* const* rs_l, int NL, int H, const float* x, long long x_bs, int x_ld, float* out,
                       long long out_bs, int out_ld, const float* mask, long long mask_bs, float* scratch, int B, int T, hipStream_t st);
// wn_small.hip: short inputs, one launch per layer: res_skip of the previous layer + F(2,5) in_layer + gate; 1 = does not apply
bool wn_small_enabled();
int pack_wn_rs16_named(DevBuf& img, int H, int Cout, const TensorTable& tab, const std::string& prefix, hipStream_t st);
int launch_wn_small_layer(const PackedConv& in_l, const float* wpf, const float* wrs, double rs_flops_per_col, const float* x, long long x_bs,
                          int x_ld, const float* ap, long long ap_bs, int ap_ld, float* xo, long long xo_bs, int xo_ld, float* out, long long out_bs,
                          int out_ld, float* ao, long long ao_bs, int ao_ld, const float* mask, long long mask_bs, const float* gadd,
                          long long gadd_bs, int gadd_ld, int gadd_ts, int skip_first, int B, int T, hipStream_t st);
int pack_wn_f25_named(DevBuf& img, int H, int K, int dil, const TensorTable& tab, const std::string& prefix, hipStream_t st);

// One streaming read over a set of weight images at the head of a call: puts them back into the 256 MB memory-side cache that the decoder's traffic
// of the call before emptied (misc_kernels.hip prefetch_kernel)
struct PrefetchSeg { const void* p; unsigned n16; unsigned pad_; };      // at most 64 KB: n16 sixteen-byte groups from p
struct WeightPrefetch {
  DevBuf tab; int n = 0; size_t bytes = 0;
  int build(const std::vector<std::pair<const void*, size_t>>& bufs, hipStream_t st);
  int run(hipStream_t st) const;
};
bool weight_prefetch_enabled();
// ------------------------------------------------------------------ small kernels (misc_kernels.hip)
int k_sequence_mask(hipStream_t st, const int64_t* lengths, float* mask, int B, int T);
int k_gate(hipStream_t st, const float* a, const float* b, float* y, int B, int H, int T);
int k_conv_post_tanh(hipStream_t st, const float* x, long long x_bs, int x_ld, const float* w, int C, int K, float slope,
                     float* y, int B, int L);
struct CopyMany { const float* src[5]; float* dst[5]; long long s_bs[5], d_bs[5]; int s_ld[5], d_ld[5], rows[5]; int n, cols, B; };
int k_copy2d_many(hipStream_t st, const CopyMany& m);      // up to five strided copies with one column count in one launch (misc_kernels.hip)
int k_copy2d(hipStream_t st, const float* src, long long s_bs, int s_ld, float* dst, long long d_bs, int d_ld, int B,
             int rows, int cols, const float* mask, long long mask_bs);
int k_flip_copy(hipStream_t st, const float* src, long long s_bs, int s_ld, float* dst, long long d_bs, int d_ld, int B,
                int C, int T);
int k_fill(hipStream_t st, float* p, size_t n, float v);
int k_frame_blocks(hipStream_t st, const float* y, int B, int Lw, int pad, int pad2, int hop, float* xt, long long xt_bs, int xt_ld, int nblocks);

// stats / profiler
bool prof_enabled();
int prof_begin(hipStream_t st, const std::string& desc, double flops);
void prof_end(hipStream_t st, int idx);
// one GEMM-family kernel launch computing nconv convolutions; exec_flops = 2 x the multiply-adds the matrix pipe really issues
// for them (shares of the direct form for k = 3, 7, 11: F(4,3) 1/2, 4/7, 6.5/11 (default); F(4,4) -, 3.5/7, 5.25/11 (128-row layout); F(2,3) 2/3, 5/7, 8/11; F(4,2) upsamplers 5/8;
// F(2,5) WN in_layers 3/5; < 0: same as flops)
void stats_add_conv(double flops, int nconv = 1, double exec_flops = -1.0);
double stats_exec_flops();
void stats_add_other();
void stats_get(long long* conv_launches, double* conv_flops, long long* other_launches);
void stats_add_bulk(long long conv_launches, double conv_flops, long long other_launches, long long convs, double exec_flops);   // replay of a captured plan
long long stats_convs();

// Every ABI handle remembers the device it was created on: its workspaces, packed weights and captured graphs live there,
// so destroy must synchronise and free on THAT device even when the caller's current device has moved on (a process that
// drives several GPUs; Python GC of a module that lives on cuda:1 while cuda:0 is current).
struct HandleDevice {
  int dev = -1;
  HandleDevice() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
};
template <class H> inline void destroy_handle(H* h) {
  if (!h) return;
  int cur = -1;
  const int dev = h->dev;
  if (hipGetDevice(&cur) != hipSuccess) cur = -1;
  if (dev >= 0 && cur != dev) (void)hipSetDevice(dev);
  (void)hipDeviceSynchronize();
  delete h;
  if (dev >= 0 && cur >= 0 && cur != dev) (void)hipSetDevice(cur);
}

inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }

}  // namespace svoc
