// A-panel-resident 1x1 convolution of the trunk (conv1x1_panel.hip): launcher shared with gemm.hip's convolution dispatch.
#pragma once
#include "gemm.h"

namespace gic {

// Launches the panel-resident kernel if the convolution qualifies (1x1 / stride 1, K = 256, N >= 512 and a multiple of 64, bf16,
// BatchNorm-sum epilogue, optional BatchNorm + ReLU of the input on load) and returns true; false: nothing launched.
bool try_conv1x1_panel(const GemmDesc& d, hipStream_t stream);

}  // namespace gic
