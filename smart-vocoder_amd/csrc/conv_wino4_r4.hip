// F(4,3) kernels of the 128-row layout (NRT = 4: conv_wino4.h), instantiated here (conv_wino4_launch.h).
#include "conv_wino4_launch.h"

namespace svoc {
SVOC_W4_INSTANTIATE_K3(4)
}  // namespace svoc
