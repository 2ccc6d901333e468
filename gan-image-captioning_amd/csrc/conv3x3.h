// Patch-resident 3x3 convolution of the trunk (conv3x3.hip): launcher shared with gemm.hip's convolution dispatch.
#pragma once
#include "gemm.h"

namespace gic {

// Launches the patch-resident kernel if the convolution qualifies (3x3 / stride 1 / pad 1, bf16 NHWC, Cin % 64 == 0, BatchNorm-sum
// epilogue, optional BatchNorm + ReLU of the input on load) and returns true; false: nothing launched, the caller falls back.
bool try_conv3x3_patch(const GemmDesc& d, hipStream_t stream);

}  // namespace gic
