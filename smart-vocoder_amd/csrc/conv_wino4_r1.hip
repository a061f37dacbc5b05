// F(4,3) kernels of the 32-row layout (NRT = 1: conv_wino4.h), instantiated here (conv_wino4_launch.h).
#include "conv_wino4_launch.h"

namespace svoc {
SVOC_W4_INSTANTIATE_K3(1)
}  // namespace svoc
