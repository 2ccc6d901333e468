// Pixel-resident 1x1 convolution of the trunk's small maps (conv1x1_pix.hip): launcher shared with gemm.hip's convolution dispatch.
#pragma once
#include "gemm.h"

namespace gic {

// Launches the pixel-resident kernel if the convolution qualifies (1x1 / stride 1, K = 256 | 512, N >= 512 and a multiple of 64, at least 128 rows, bf16,
// BatchNorm-sum epilogue, BatchNorm + ReLU of the input on load) and returns true; false: nothing launched.
bool try_conv1x1_pix(const GemmDesc& d, hipStream_t stream);

}  // namespace gic
