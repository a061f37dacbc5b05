// The 32-row layout (NRT = 1) with k = 7 / 11 in Winograd F(4,4) form (conv_wino4.h: four-tap groups in seven products, no
// left-over taps - k = 11: 21 products per window instead of 26, k = 7: 14 instead of 16; reference modules.py:190-207 at
// models.py:129-133), instantiated here (conv_wino4_launch.h).  Round 4: the F(4,3) streams run at 67-71 cycles per MFMA against
// a pipe limit of 64, so the lever is once more FEWER products.
#include "conv_wino4_launch.h"

namespace svoc {
SVOC_W4_INSTANTIATE_F44(1)
}  // namespace svoc
