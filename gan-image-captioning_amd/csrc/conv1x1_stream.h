// Streaming shallow-K 1x1 convolution of the trunk (conv1x1_stream.hip): launcher shared with gemm.hip's convolution dispatch.
#pragma once
#include "gemm.h"

namespace gic {

// Launches the persistent streaming kernel if the convolution qualifies (1x1 / stride 1, K = 64 or 128, bf16, BatchNorm-sum epilogue,
// optional BatchNorm + ReLU of the input on load, >= 1024 output tiles) and returns true; false: nothing launched.
bool try_conv1x1_stream(const GemmDesc& d, hipStream_t stream);

}  // namespace gic
