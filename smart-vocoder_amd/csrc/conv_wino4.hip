// Winograd F(4,3) / F(4,4) convolutions (kernels: conv_wino4_kernels.h): weight transform + packing, and the dispatch to the
// instantiations (conv_wino4_r4.hip / _r2.hip / _r1.hip: one per row-tile layout in F(4,3) form; conv_wino44_r*.hip: k = 7 / 11 in F(4,4) form).
#include "conv_wino4.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace svoc {

// ------------------------------------------------------------------ weight transform + packing
// wp[m-tile][chunk][slot][k-group][lane][4] as pack_wino (conv_wino.hip) with slots = 6 G + ND: slot 6 g + p holds U_p of the
// three-tap group g, slot 6 G + t the plain tap 4 t + 3.  f44: slots = 7 G, slot 7 g + p holds U_p of the four-tap group g (taps
// beyond the kernel are zero).
__global__ void pack_wino4_kernel(const float* __restrict__ src, const float* __restrict__ scale, float* __restrict__ wp, int Cin,
                                  int Cout, int K, int nchunks, int slots, long long total, int f44) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int s = (int)(e & 3);
  const int lane = (int)((e >> 2) & 63);
  long long rest = e >> 8;
  const int kg = (int)(rest & 3); rest >>= 2;
  const int slot = (int)(rest % slots); rest /= slots;
  const int ch = (int)(rest % nchunks);
  const int mt = (int)(rest / nchunks);
  const int row = mt * 32 + (lane & 31);
  const int chan = ch * KC + 2 * (4 * kg + s) + (lane >> 5);
  float val = 0.f;
  if (row < Cout && chan < Cin) {
    const float* w = src + ((long long)row * Cin + chan) * K;
    const float sc = scale ? scale[row] : 1.0f;
    const int G = (K + 1) / 4;
    if (f44) {
      const int g = slot / 7, pp = slot % 7;
      const float w0 = w[4 * g] * sc, w1 = w[4 * g + 1] * sc, w2 = w[4 * g + 2] * sc, w3 = 4 * g + 3 < K ? w[4 * g + 3] * sc : 0.f;
      switch (pp) {
        case 0: val = w0; break;
        case 1: val = -((w0 + w2) + (w1 + w3)) * (2.0f / 9.0f); break;
        case 2: val = -((w0 + w2) - (w1 + w3)) * (2.0f / 9.0f); break;
        case 3: val = ((w0 + 4.f * w2) + (2.f * w1 + 8.f * w3)) * (1.0f / 90.0f); break;
        case 4: val = ((w0 + 4.f * w2) - (2.f * w1 + 8.f * w3)) * (1.0f / 90.0f); break;
        case 5: val = ((w0 + 0.25f * w2) + (0.5f * w1 + 0.125f * w3)) * (32.0f / 45.0f); break;
        default: val = ((w0 + 0.25f * w2) - (0.5f * w1 + 0.125f * w3)) * (32.0f / 45.0f); break;
      }
    } else if (slot < 6 * G) {
      const int g = slot / 6, pp = slot % 6;
      const float w0 = w[4 * g] * sc, w1 = w[4 * g + 1] * sc, w2 = w[4 * g + 2] * sc;
      switch (pp) {
        case 0: val = 0.25f * w0; break;
        case 1: val = -((w0 + w2) + w1) * (1.0f / 6.0f); break;
        case 2: val = -((w0 + w2) - w1) * (1.0f / 6.0f); break;
        case 3: val = ((w0 + 4.f * w2) + 2.f * w1) * (1.0f / 24.0f); break;
        case 4: val = ((w0 + 4.f * w2) - 2.f * w1) * (1.0f / 24.0f); break;
        default: val = w2; break;
      }
    } else {
      val = w[4 * (slot - 6 * G) + 3] * sc;               // taps 3, 7
    }
  }
  wp[e] = val;
}

int wino4_slots(int K, bool f44) { const int G = (K + 1) / 4; return f44 ? 7 * G : 6 * G + (G - 1); }

int pack_wino4_image(float* wp4, int Cin, int Cout, int K, bool f44, const float* w_or_v, const float* scale, hipStream_t st) {
  const int nchunks = (Cin + KC - 1) / KC, mtiles = (Cout + 31) / 32, slots = wino4_slots(K, f44);
  const long long total = (long long)mtiles * nchunks * slots * 4 * 256;
  hipLaunchKernelGGL(pack_wino4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w_or_v, scale, wp4, Cin, Cout, K, nchunks,
                     slots, total, f44 ? 1 : 0);
  SVOC_HIP(hipGetLastError());
  return SVOC_OK;
}

// ------------------------------------------------------------------ launches
bool wino4_enabled() {
  static const bool on = !(getenv("SVOC_WINO_F4") && atoi(getenv("SVOC_WINO_F4")) == 0);      // SVOC_WINO_F4=0: the F(2,3) kernels
  return on;
}
// k = 7 / 11 run in F(4,4) form wherever the F(4,3) family applies (the F(4,3) form of those kernel sizes was an A/B arm until round 5)
bool wino44_enabled() { return wino4_enabled(); }
// SVOC_W4_C32=0: the C = 32 stage keeps the fused direct-form ResBlock kernel (resblock_fused.hip)
bool wino4_c32_enabled() {
  static const bool on = wino4_enabled() && !(getenv("SVOC_W4_C32") && atoi(getenv("SVOC_W4_C32")) == 0);
  return on;
}
// column tiles per row: a tile is 32 * (4 / NRT) consecutive windows; a row of L outputs has D * ceil(L / 4D) windows
int wino4_ntn(int L, int D, int NRT) {
  const long long nw = (long long)D * ((L + 4 * D - 1) / (4 * D));
  const int nwt = 32 * (4 / NRT);
  return (int)((nw + nwt - 1) / nwt);
}
// one persistent workgroup per CU (eight waves of up to 256 registers)
unsigned wino4_grid(long long total) { return (unsigned)std::min<long long>(total, (long long)device_cu_count()); }

// the instantiations, one translation unit per row-tile layout and form (conv_wino4_launch.h)
template <int NRT, bool F44> int wino4_launch_nrt(const WinoArgs& w, int K, int D, long long total, hipStream_t st);
template <int NRT, bool F44> int wino4_launch_group_nrt(const WinoGroup& g, int D, int in_perm, int out_perm, long long total, hipStream_t st);
#define SVOC_W4_EXTERN(NRT)                                                                                               \
  extern template int wino4_launch_nrt<NRT, false>(const WinoArgs&, int, int, long long, hipStream_t);                    \
  extern template int wino4_launch_nrt<NRT, true>(const WinoArgs&, int, int, long long, hipStream_t);                     \
  extern template int wino4_launch_group_nrt<NRT, true>(const WinoGroup&, int, int, int, long long, hipStream_t);
SVOC_W4_EXTERN(4) SVOC_W4_EXTERN(2) SVOC_W4_EXTERN(1)
#undef SVOC_W4_EXTERN

// f44: the weight image is in F(4,4) form (k = 7 / 11); k = 3: F(4,3)
int wino4_launch(const WinoArgs& w, int K, int D, int NRT, bool f44, long long total, hipStream_t st) {
  if (f44 != (K >= 7)) return 1;
  if (f44) return NRT == 4 ? wino4_launch_nrt<4, true>(w, K, D, total, st) : (NRT == 2 ? wino4_launch_nrt<2, true>(w, K, D, total, st) : wino4_launch_nrt<1, true>(w, K, D, total, st));
  return NRT == 4 ? wino4_launch_nrt<4, false>(w, K, D, total, st) : (NRT == 2 ? wino4_launch_nrt<2, false>(w, K, D, total, st) : wino4_launch_nrt<1, false>(w, K, D, total, st));
}
// in_perm (D = 1): 0, or the dilation of the convolutions that wrote the members' inputs window-major; out_perm (D > 1): nonzero =
// the members write window-major.  Members k = 11 / 7 in F(4,4) form, k = 3 in F(4,3)
int wino4_launch_group(const WinoGroup& g, int D, int NRT, int in_perm, int out_perm, bool f44, long long total, hipStream_t st) {
  if (!f44) return 1;
  return NRT == 4 ? wino4_launch_group_nrt<4, true>(g, D, in_perm, out_perm, total, st)
                  : (NRT == 2 ? wino4_launch_group_nrt<2, true>(g, D, in_perm, out_perm, total, st) : wino4_launch_group_nrt<1, true>(g, D, in_perm, out_perm, total, st));
}

}  // namespace svoc
