// zz_level2.h -- placeholder until the level >= 2 kernels land: flags an error instead of encoding.
#pragma once
#include "zz_common.h"
#define ZZ_L2_SCRATCH_BYTES 16
namespace zz {
__global__ void k_l2_unimplemented(zz_packet_params P) { if (threadIdx.x == 0 && blockIdx.x == 0) atomicOr(P.err, 4u); }
static inline void launch_level2(const zz_packet_params& pp, uint8_t*, hipStream_t st)
{
    hipLaunchKernelGGL(k_l2_unimplemented, dim3(1), dim3(64), 0, st, pp);
}
}
