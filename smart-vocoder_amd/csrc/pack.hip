// Load-time weight preparation: weight-norm fold + repack into MFMA fragment order.
//
// The reference keeps (weight_g, weight_v) and re-normalises on every forward
// (torch weight_norm: modules.py:128,135,145,191-206; models.py:125 — 172
// launches per infer call).  Here w = v * (g / ||v||) is folded once, on the
// device, while the weights are rewritten into the layout the MFMA kernel
// streams: wp[m-tile][k-step group][lane][4], where lane l of k-step ks holds
// W[row = 32*mt + (l & 31)][channel = chunk*32 + 2*cp + (l >> 5)][tap j] with
// ks = (chunk*ktaps + j)*16 + cp.  Row/column maps fold in:
//   - the tile-pair interleave of two-half convolutions (WN gate, proj, coupling stats),
//   - modules.Flip (modules.py:270-277) as channel permutations,
//   - the polyphase split of ConvTranspose1d: virtual row o*s + r uses taps r, r+s, ...
#include "svoc_internal.h"

namespace svoc {

__global__ void wn_scale_kernel(const float* __restrict__ v, const float* __restrict__ g, float* __restrict__ scale,
                                long long inner) {
  __shared__ float red[256];
  const int i = blockIdx.x;
  const float* p = v + (long long)i * inner;
  float s = 0.f;
  for (long long k = threadIdx.x; k < inner; k += blockDim.x) s += p[k] * p[k];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) scale[i] = g[i] / sqrtf(red[0]);
}

__global__ void fold_apply_kernel(const float* __restrict__ v, const float* __restrict__ scale, float* __restrict__ w,
                                  long long inner, long long total) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < total) w[e] = v[e] * scale[e / inner];
}

__global__ void pack_weights_kernel(const float* __restrict__ src, const float* __restrict__ scale, int scale_by_out,
                                    const int* __restrict__ row_o, const int* __restrict__ row_tap0,
                                    const int* __restrict__ col_src, float* __restrict__ wp, int ksg_total, int ktaps,
                                    int tap_stride, int Ktot, long long so, long long sc, long long stp,
                                    long long total) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int s = (int)(e & 3);
  const int lane = (int)((e >> 2) & 63);
  const long long rest = e >> 8;
  const int ksg = (int)(rest % ksg_total);
  const int mt = (int)(rest / ksg_total);
  const int ks = ksg * 4 + s;
  const int per_chunk = ktaps * (KC / 2);
  const int ch = ks / per_chunk;
  const int rem = ks - ch * per_chunk;
  const int j = rem / (KC / 2);
  const int cp = rem - j * (KC / 2);
  const int chan = ch * KC + 2 * cp + (lane >> 5);
  const int row = mt * 32 + (lane & 31);
  const int o = row_o[row];
  const int cs = col_src[chan];
  const int tap = row_tap0[row] + j * tap_stride;
  float val = 0.f;
  if (o >= 0 && cs >= 0 && tap < Ktot) {
    val = src[(long long)o * so + (long long)cs * sc + (long long)tap * stp];
    if (scale) val = val * scale[scale_by_out ? o : cs];
  }
  wp[e] = val;
}

__global__ void pack_bias_kernel(const float* __restrict__ bias, const int* __restrict__ row_o, float* __restrict__ bp,
                                 int rowsP) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rowsP) return;
  const int o = row_o[r];
  bp[r] = (bias && o >= 0) ? bias[o] : 0.f;
}

namespace {
struct TmpDev {
  void* p = nullptr;
  ~TmpDev() { if (p) (void)hipFree(p); }
  int upload(const void* h, size_t n) {
    if (hipMalloc(&p, n) != hipSuccess) { set_error("hipMalloc(%zu) failed while packing", n); return SVOC_ERR_NOMEM; }
    SVOC_HIP(hipMemcpy(p, h, n, hipMemcpyHostToDevice));
    return SVOC_OK;
  }
  int alloc(size_t n) {
    if (hipMalloc(&p, n) != hipSuccess) { set_error("hipMalloc(%zu) failed while packing", n); return SVOC_ERR_NOMEM; }
    return SVOC_OK;
  }
};
}  // namespace

int pack_conv(PackedConv& pc, const PackSpec& sp, const float* w, const float* g, const float* bias, hipStream_t st) {
  if (sp.Cin <= 0 || sp.Cout <= 0 || sp.K <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "pack_conv: bad dims %d %d %d", sp.Cin, sp.Cout, sp.K);
  pc.Cin = sp.Cin;
  pc.Cout = sp.Cout;
  pc.CinP = round_up(sp.Cin, KC);
  pc.paired = sp.paired;
  pc.transposed = sp.transposed;
  std::vector<int> row_o, row_tap0, col_src(pc.CinP, -1);
  int tap_stride = 1;
  long long so, sc, stp = 1;
  int scale_by_out = 1;
  long long n_slices, inner;
  if (sp.transposed) {
    const int s = sp.stride;
    if (s <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "pack_conv: stride %d", s);
    pc.ktaps = (sp.K + s - 1) / s;
    pc.dil = -1;
    pc.pad = 0;
    pc.ups_s = s;
    pc.ups_pad = sp.tpad;
    pc.rows = sp.Cout * s;
    const int rowsP = round_up(pc.rows, 32);
    row_o.assign(rowsP, -1);
    row_tap0.assign(rowsP, 0);
    for (int v = 0; v < pc.rows; ++v) {
      const int o = v / s;
      row_o[v] = sp.out_perm ? sp.out_perm[o] : o;
      row_tap0[v] = v % s;
    }
    tap_stride = s;
    so = sp.K;
    sc = (long long)sp.Cout * sp.K;
    scale_by_out = 0;
    n_slices = sp.Cin;
    inner = (long long)sp.Cout * sp.K;
    pc.half_rows = 0;
  } else {
    pc.ktaps = sp.K;
    pc.dil = sp.dil;
    pc.pad = sp.pad >= 0 ? sp.pad : (sp.K * sp.dil - sp.dil) / 2;
    pc.ups_s = 1;
    pc.ups_pad = 0;
    if (sp.paired) {
      if (sp.Cout % 2) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "paired conv needs an even Cout");
      const int H = sp.Cout / 2, Hp = round_up(H, 32);
      pc.half_rows = H;
      pc.rows = 2 * Hp;
      row_o.assign(pc.rows, -1);
      row_tap0.assign(pc.rows, 0);
      for (int t = 0; t < Hp / 32; ++t)
        for (int rr = 0; rr < 32; ++rr) {
          const int chn = t * 32 + rr;
          if (chn >= H) continue;
          const int oa = chn, ob = H + chn;
          row_o[(2 * t) * 32 + rr] = sp.out_perm ? sp.out_perm[oa] : oa;
          row_o[(2 * t + 1) * 32 + rr] = sp.out_perm ? sp.out_perm[ob] : ob;
        }
    } else if (sp.split_at > 0 && sp.split_at < sp.Cout) {
      const int n0 = sp.split_at, n1 = sp.Cout - sp.split_at;
      const int p0 = round_up(n0, 32), p1 = round_up(n1, 32);
      pc.half_rows = 0;
      pc.rows = p0 + p1;
      pc.split_row = p0;
      row_o.assign(pc.rows, -1);
      row_tap0.assign(pc.rows, 0);
      for (int o = 0; o < n0; ++o) row_o[o] = sp.out_perm ? sp.out_perm[o] : o;
      for (int o = 0; o < n1; ++o) row_o[p0 + o] = sp.out_perm ? sp.out_perm[n0 + o] : n0 + o;
    } else {
      pc.half_rows = 0;
      pc.rows = sp.Cout;
      const int rowsP = round_up(pc.rows, 32);
      row_o.assign(rowsP, -1);
      row_tap0.assign(rowsP, 0);
      for (int o = 0; o < sp.Cout; ++o) row_o[o] = sp.out_perm ? sp.out_perm[o] : o;
    }
    so = (long long)sp.Cin * sp.K;
    sc = sp.K;
    scale_by_out = 1;
    n_slices = sp.Cout;
    inner = (long long)sp.Cin * sp.K;
  }
  for (int c = 0; c < sp.Cin; ++c) col_src[c] = sp.in_perm ? sp.in_perm[c] : c;
  const int rowsP = (int)row_o.size();
  pc.mtiles = rowsP / 32;
  pc.ksg_total = (pc.CinP / KC) * pc.ktaps * (KC / 8);
  pc.flops_per_col = 2.0 * sp.Cin * sp.Cout * sp.K;

  const long long total = (long long)pc.mtiles * pc.ksg_total * 256;
  // one extra (zeroed) group: the kernels prefetch the weight stream one group past the last one they use
  SVOC_TRY(pc.wp.ensure((size_t)(total + 256) * sizeof(float)));
  SVOC_HIP(hipMemsetAsync(pc.wp.f() + total, 0, 256 * sizeof(float), st));
  SVOC_TRY(pc.bias.ensure((size_t)rowsP * sizeof(float)));

  TmpDev d_row_o, d_row_tap0, d_col_src, d_scale;
  SVOC_TRY(d_row_o.upload(row_o.data(), row_o.size() * sizeof(int)));
  SVOC_TRY(d_row_tap0.upload(row_tap0.data(), row_tap0.size() * sizeof(int)));
  SVOC_TRY(d_col_src.upload(col_src.data(), col_src.size() * sizeof(int)));
  if (g) {
    SVOC_TRY(d_scale.alloc((size_t)n_slices * sizeof(float)));
    hipLaunchKernelGGL(wn_scale_kernel, dim3((unsigned)n_slices), dim3(256), 0, st, w, g, (float*)d_scale.p, inner);
  }
  const int thr = 256;
  hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((total + thr - 1) / thr)), dim3(thr), 0, st, w,
                     (const float*)d_scale.p, scale_by_out, (const int*)d_row_o.p, (const int*)d_row_tap0.p,
                     (const int*)d_col_src.p, pc.wp.f(), pc.ksg_total, pc.ktaps, tap_stride, sp.K, so, sc, stp, total);
  hipLaunchKernelGGL(pack_bias_kernel, dim3((rowsP + 255) / 256), dim3(256), 0, st, bias, (const int*)d_row_o.p,
                     pc.bias.f(), rowsP);
  SVOC_HIP(hipGetLastError());
  SVOC_HIP(hipStreamSynchronize(st));   // temporaries are freed on return
  return SVOC_OK;
}

int fold_weight_norm(hipStream_t st, const float* v, const float* g, float* w, long long d0, long long inner) {
  TmpDev d_scale;
  SVOC_TRY(d_scale.alloc((size_t)d0 * sizeof(float)));
  hipLaunchKernelGGL(wn_scale_kernel, dim3((unsigned)d0), dim3(256), 0, st, v, g, (float*)d_scale.p, inner);
  const long long total = d0 * inner;
  hipLaunchKernelGGL(fold_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, v, (const float*)d_scale.p, w,
                     inner, total);
  SVOC_HIP(hipGetLastError());
  SVOC_HIP(hipStreamSynchronize(st));
  return SVOC_OK;
}

int pack_conv_named(PackedConv& pc, PackSpec sp, const TensorTable& tab, const std::string& prefix, hipStream_t st,
                    bool bias_required) {
  const svoc_tensor* w = tab.find(prefix + ".weight");
  const svoc_tensor* v = tab.find(prefix + ".weight_v");
  const svoc_tensor* g = tab.find(prefix + ".weight_g");
  const svoc_tensor* b = tab.find(prefix + ".bias");
  const svoc_tensor* src = w ? w : v;
  if (!src) SVOC_FAIL(SVOC_ERR_MISSING_TENSOR, "missing tensor %s.weight / .weight_v", prefix.c_str());
  if (!w && !g) SVOC_FAIL(SVOC_ERR_MISSING_TENSOR, "missing tensor %s.weight_g", prefix.c_str());
  if (bias_required && !b) SVOC_FAIL(SVOC_ERR_MISSING_TENSOR, "missing tensor %s.bias", prefix.c_str());
  const int64_t d0 = sp.transposed ? sp.Cin : sp.Cout, d1 = sp.transposed ? sp.Cout : sp.Cin;
  if (src->ndim != 3 || src->shape[0] != d0 || src->shape[1] != d1 || src->shape[2] != sp.K)
    SVOC_FAIL(SVOC_ERR_SHAPE, "tensor %s has shape [%lld,%lld,%lld], expected [%lld,%lld,%d]", src->name,
              (long long)src->shape[0], (long long)src->shape[1], (long long)src->shape[2], (long long)d0, (long long)d1,
              sp.K);
  if (b && (b->ndim != 1 || b->shape[0] != sp.Cout)) SVOC_FAIL(SVOC_ERR_SHAPE, "tensor %s has wrong shape", b->name);
  if (!w && g && g->shape[0] != d0) SVOC_FAIL(SVOC_ERR_SHAPE, "tensor %s has wrong shape", g->name);
  return pack_conv(pc, sp, src->data, w ? nullptr : g->data, b ? b->data : nullptr, st);
}

}  // namespace svoc
