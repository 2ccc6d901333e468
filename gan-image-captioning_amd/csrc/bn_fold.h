// BatchNorm batch statistics arrive as per-channel sums in a few REPLICAS ([nrep][2][C] f32: the producing convolution's workgroups add
// into replica b % nrep so that few atomics share an address); every consumer folds them per channel.  Shared by the kernels that
// normalise on load (gemm.hip, conv3x3.hip, conv1x1_stream.hip, conv1x1_panel.hip) and by encoder.hip's bn_act family.
#pragma once
#include "common.h"

namespace gic {

// Sum / sum of squares of channel c over the replicas: eight independent pairs of loads in flight per round trip (a load / add loop
// waits for each replica in turn: measured as one L2 round trip per replica in every workgroup's prologue); replicas past nrep
// re-read the last one with weight 0.
__device__ __forceinline__ void fold_replicas(const float* stats, int nrep, int C, int c, float& s1, float& s2) {
  s1 = s2 = 0.f;
  for (int r0 = 0; r0 < nrep; r0 += 8) {
    float a[8], q[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const long rr = min(r0 + r, nrep - 1);
      a[r] = stats[rr * 2 * C + c];
      q[r] = stats[rr * 2 * C + C + c];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float wgt = r0 + r < nrep ? 1.f : 0.f;
      s1 += wgt * a[r]; s2 += wgt * q[r];
    }
  }
}

}  // namespace gic
