// conv3 -> bn3 + shortcut + ReLU -> next conv1 in one launch (conv_b2b.hip); see there.
#pragma once
#include "common.h"

namespace gic {

struct B2bDesc {
  const void* y2; const void* w3; const void* res; const void* w1n;     // bf16: [M, C2], [4 C2, C2], [M, 4 C2], [C1N, 4 C2]
  void* out; void* y1n;                                                  // bf16: [M, 4 C2] block output, [M, C1N] raw conv1_next output
  const float* stats2; const float* gamma2; const float* beta2;          // BatchNorm of y2 (bn2): column sums [nrep2][2][C2]
  const float* stats3; const float* gamma3; const float* beta3;          // bn3: sums of conv3's output from the statistics-only pass
  const float* res_stats; const float* res_gamma; const float* res_beta; // the projection shortcut's BatchNorm, or null (identity)
  float* stats1;                                                         // [nrep1][2][C1N]: conv1_next's column sums are ADDED
  int nrep2, nrep3, res_nrep, nrep1;
  float inv_count;                                                       // 1 / M (all four BatchNorms normalise over the same rows)
  int M;
  unsigned y2_bytes, res_bytes, y1n_bytes = 0;
};

// false: shapes it has no instantiation for (the caller runs the separate launches)
bool try_conv_b2b(const B2bDesc& d, int C2, int C1N, hipStream_t stream);

}  // namespace gic
