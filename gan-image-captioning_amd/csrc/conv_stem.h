// Streaming stem convolution of the trunk (conv_stem.hip): launcher shared with gemm.hip's convolution dispatch.
#pragma once
#include "gemm.h"

namespace gic {

// Launches the streaming stem kernel if the convolution qualifies (window 7 x 8 over a pre-padded NHWC4 bf16 image, stride 2, 64 output
// channels, output rows of <= 128 pixels, BatchNorm-sum epilogue) and returns true; false: nothing launched.
bool try_conv_stem(const GemmDesc& d, hipStream_t stream);

}  // namespace gic
