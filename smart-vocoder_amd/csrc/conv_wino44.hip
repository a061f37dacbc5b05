// The F(4,4) instantiations of the Winograd convolution kernels (conv_wino4_kernels.h; form: conv_wino4.h): the 128-row layout
// (NRT = 4, C % 128 == 0), k = 7 / 11 (reference modules.py:190-207 at models.py:129-133, the C = 256 / 128 MRF stages).  Round 4: the
// F(4,3) streams of these stages run at 67 cycles per MFMA against a pipe limit of 64 and their producers have slack, so the lever is
// once more FEWER products: four-tap groups in seven products and no left-over taps - k = 11: 21 products per window instead of 26,
// k = 7: 14 instead of 16 (k = 3 stays F(4,3): 6).  A translation unit of its own so that the build compiles it beside conv_wino4.hip.
#include "conv_wino4_kernels.h"

namespace svoc {

unsigned wino4_grid(long long total);

template <int K, int D, int NRT = 4>
static int wino44_launch_one(const WinoArgs& w, long long total, hipStream_t st) {
  using Geo = W4Geo<K, D, NRT, 0, 1, true>;
  static_assert(Geo::LDS_BYTES <= 160 * 1024, "tile does not fit");
  const unsigned grid = wino4_grid(total);
  if (w.dbg) {                                             // stamped build (tools/wino4_timeline.py)
    auto kern = conv_wino4_kernel<K, D, NRT, true, true>;
    SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), (size_t)Geo::LDS_BYTES, st, w, (int)total);
  } else {
    auto kern = conv_wino4_kernel<K, D, NRT, false, true>;
    SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), (size_t)Geo::LDS_BYTES, st, w, (int)total);
  }
  return SVOC_OK;
}
int wino44_launch(const WinoArgs& w, int K, int D, int NRT, long long total, hipStream_t st) {
  int rc = 1;
  if (NRT != 4) {                                          // SVOC_W4_F44=2: the undilated convolutions of the 64- / 32-row layouts (measurements)
    if (D != 1 || !(K == 7 || K == 11)) return 1;
    if (NRT == 2) return K == 7 ? wino44_launch_one<7, 1, 2>(w, total, st) : wino44_launch_one<11, 1, 2>(w, total, st);
    return K == 7 ? wino44_launch_one<7, 1, 1>(w, total, st) : wino44_launch_one<11, 1, 1>(w, total, st);
  }
#define SVOC_W44(KK, DD) if (K == KK && D == DD) rc = wino44_launch_one<KK, DD>(w, total, st);
  SVOC_W44(7, 1) SVOC_W44(11, 1) SVOC_W44(7, 3) SVOC_W44(11, 3) SVOC_W44(7, 5) SVOC_W44(11, 5)
#undef SVOC_W44
  return rc;
}

template <int D, int PERM>
static int wino44_launch_group_d(const WinoGroup& g, long long total, hipStream_t st) {
  auto kern = conv_wino4_group_kernel<D, 4, PERM, true>;
  SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
  constexpr int P = D == 1 ? PERM : 0;
  const size_t lds = std::max((size_t)W4Geo<11, D, 4, P, 1, true>::LDS_BYTES, std::max((size_t)W4Geo<7, D, 4, P, 1, true>::LDS_BYTES, (size_t)W4Geo<3, D, 4, P>::LDS_BYTES));
  hipLaunchKernelGGL(kern, dim3(wino4_grid(total)), dim3(512), lds, st, g);
  return SVOC_OK;
}
int wino44_launch_group(const WinoGroup& g, int D, int in_perm, int out_perm, long long total, hipStream_t st) {
  if (D == 1) return in_perm == 5 ? wino44_launch_group_d<1, 5>(g, total, st) : (in_perm == 3 ? wino44_launch_group_d<1, 3>(g, total, st) : wino44_launch_group_d<1, 0>(g, total, st));
  if (D == 3) return out_perm ? wino44_launch_group_d<3, 3>(g, total, st) : wino44_launch_group_d<3, 0>(g, total, st);
  return out_perm ? wino44_launch_group_d<5, 5>(g, total, st) : wino44_launch_group_d<5, 0>(g, total, st);
}

template <int PERM>
static int wino44_launch_accum_p(const WinoGroup& g, long long total, hipStream_t st) {
  auto kern = conv_wino4_accum_kernel<4, PERM, true>;
  SVOC_TRY(ensure_max_dyn_lds((const void*)kern));
  const size_t lds = std::max((size_t)W4Geo<11, 1, 4, PERM, 1, true>::LDS_BYTES, std::max((size_t)W4Geo<7, 1, 4, PERM, 1, true>::LDS_BYTES, (size_t)W4Geo<3, 1, 4, PERM>::LDS_BYTES));
  hipLaunchKernelGGL(kern, dim3(wino4_grid(total)), dim3(512), lds, st, g);
  return SVOC_OK;
}
int wino44_launch_accum(const WinoGroup& g, int in_perm, long long total, hipStream_t st) {
  if (in_perm == 5) return wino44_launch_accum_p<5>(g, total, st);
  if (in_perm == 3) return wino44_launch_accum_p<3>(g, total, st);
  return wino44_launch_accum_p<0>(g, total, st);
}

}  // namespace svoc
