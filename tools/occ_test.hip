// How many workgroups with L bytes of LDS does the runtime place on one CU? (manual: hipcc --offload-arch=gfx950 tools/occ_test.hip -o /tmp/occ && /tmp/occ)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int L> __global__ void k(unsigned* o) { __shared__ unsigned s[L / 4]; s[threadIdx.x] = threadIdx.x; __syncthreads(); o[blockIdx.x] = s[(threadIdx.x * 7) % (L / 4)]; }
template <int L> void probe(int threads) { int nb = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k<L>, threads, 0); printf("LDS %6d B, %3d threads: %d workgroups per CU\n", L, threads, nb); }
int main() {
    probe<16384>(64); probe<16384>(128); probe<16896>(128); probe<17408>(128); probe<17920>(128); probe<18432>(128);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu\n", p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor);
    return 0;
}
