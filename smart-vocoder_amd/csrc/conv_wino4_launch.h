// Launch templates of the F(4,3) / F(4,4) kernels for ONE row-tile layout NRT and ONE form (conv_wino4_kernels.h).  Each (layout, form)
// is instantiated in a translation unit of its own (conv_wino4_r4.hip / _r2.hip / _r1.hip: 128- / 64- / 32-row blocks in F(4,3) form;
// conv_wino44_r*.hip: k = 7 / 11 in F(4,4) form) so that the build compiles them side by side - as one file the family took five
// minutes of a six-minute build.
#pragma once
#include "conv_wino4_kernels.h"

namespace svoc {

unsigned wino4_grid(long long total);                      // conv_wino4.hip: one persistent workgroup per CU

// F44: the form of the k = 7 / 11 convolutions; k = 3 stays F(4,3) in these launches (six products against seven)
template <int K, int D, int NRT, int PERM = 0, bool F44 = false>
static size_t wino4_lds() { return (size_t)W4Geo<K, D, NRT, D == 1 ? PERM : 0, 1, (F44 && K >= 7)>::LDS_BYTES; }
template <int K, int D, int NRT, bool F44 = false>
static int wino4_launch_one(const WinoArgs& w, long long total, hipStream_t st) {
  static_assert(W4Geo<K, D, NRT, 0, 1, F44>::LDS_BYTES <= 160 * 1024, "tile does not fit");
  const unsigned grid = wino4_grid(total);
  const size_t lds = wino4_lds<K, D, NRT, 0, F44>();
  if (w.dbg) {                                             // stamped build (tools/wino4_timeline.py)
    auto kern = conv_wino4_kernel<K, D, NRT, true, F44>;
    SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, w, (int)total);
  } else {
    auto kern = conv_wino4_kernel<K, D, NRT, false, F44>;
    SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, w, (int)total);
  }
  return SVOC_OK;
}
template <int NRT, bool F44>
int wino4_launch_nrt(const WinoArgs& w, int K, int D, long long total, hipStream_t st) {
  int rc = 1;
#define SVOC_W4(KK, DD) if (K == KK && D == DD) rc = wino4_launch_one<KK, DD, NRT, F44>(w, total, st);
  // k = 3 exists in F(4,3) form only, k = 7 / 11 in F(4,4) form only (round 5: F(4,3) for k = 7 / 11 was reachable through SVOC_W4_F44=0 alone and went
  // with that switch; its measured cost is in DESIGN.md)
  if constexpr (!F44) { SVOC_W4(3, 1) SVOC_W4(3, 3) SVOC_W4(3, 5) }
  else { SVOC_W4(7, 1) SVOC_W4(11, 1) SVOC_W4(7, 3) SVOC_W4(11, 3) SVOC_W4(7, 5) SVOC_W4(11, 5) }
#undef SVOC_W4
  return rc;
}
template <int D, int NRT, int PERM, bool F44>
static int wino4_launch_group_d(const WinoGroup& g, long long total, hipStream_t st) {
  auto kern = conv_wino4_group_kernel<D, NRT, PERM, F44>;
  SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
  const size_t l11 = wino4_lds<11, D, NRT, PERM, F44>(), l7 = wino4_lds<7, D, NRT, PERM, F44>(), l3 = wino4_lds<3, D, NRT, PERM>();
  const size_t lds = std::max(l11, std::max(l7, l3));
  hipLaunchKernelGGL(kern, dim3(wino4_grid(total)), dim3(512), lds, st, g);
  return SVOC_OK;
}
// in_perm (D = 1): 0, or the dilation of the convolutions that wrote the members' inputs window-major; out_perm (D > 1): nonzero =
// the members write window-major
template <int NRT, bool F44>
int wino4_launch_group_nrt(const WinoGroup& g, int D, int in_perm, int out_perm, long long total, hipStream_t st) {
  if (D == 1) return in_perm == 5 ? wino4_launch_group_d<1, NRT, 5, F44>(g, total, st) : (in_perm == 3 ? wino4_launch_group_d<1, NRT, 3, F44>(g, total, st) : wino4_launch_group_d<1, NRT, 0, F44>(g, total, st));
  if (D == 3) return out_perm ? wino4_launch_group_d<3, NRT, 3, F44>(g, total, st) : wino4_launch_group_d<3, NRT, 0, F44>(g, total, st);
  return out_perm ? wino4_launch_group_d<5, NRT, 5, F44>(g, total, st) : wino4_launch_group_d<5, NRT, 0, F44>(g, total, st);
}
#define SVOC_W4_INSTANTIATE_K3(NRT) template int wino4_launch_nrt<NRT, false>(const WinoArgs&, int, int, long long, hipStream_t);
#define SVOC_W4_INSTANTIATE_F44(NRT)                                                                                      \
  template int wino4_launch_nrt<NRT, true>(const WinoArgs&, int, int, long long, hipStream_t);                            \
  template int wino4_launch_group_nrt<NRT, true>(const WinoGroup&, int, int, int, long long, hipStream_t);

}  // namespace svoc
