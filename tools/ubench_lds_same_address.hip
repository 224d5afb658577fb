// Probe: what does a store cost when all 64 lanes of the instruction hit ONE LDS address (level 2's hand-over words, written
// "by every lane, no lane mask to set up"), against lane 0 alone under a lane mask and against 64 distinct addresses?
// One wavefront per workgroup, one workgroup; cycles per store from s_memtime around 32 stores.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ub tools/ubench_lds_same_address.hip && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int KIND>
__global__ __launch_bounds__(64) void k(unsigned long long* out, uint32_t seed)
{
    __shared__ uint32_t buf[4096];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) buf[i] = 0;
    __syncthreads();
    unsigned long long t0, t1, acc = 0;
    uint32_t x = seed;
    for (int r = 0; r < 256; ++r) {
        x = x * 1664525u + 1013904223u;
        const uint32_t base = (x >> 20) & 1023u;          // uniform
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            if (KIND == 0) buf[base + s] = x + s;                                 // all lanes, one address
            else if (KIND == 1) { if (lane == 0) buf[base + s] = x + s; }          // lane 0 under a mask
            else if (KIND == 2) buf[((base + s * 64) & 4095u & ~63u) + lane] = x + s;   // 64 consecutive dwords
            else if (KIND == 3) ((uint16_t*)buf)[base + s] = (uint16_t)(x + s);   // all lanes, one 16-bit address
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        acc += t1 - t0;
    }
    if (lane == 0) out[KIND] = acc;
    if (buf[lane] == 0xdeadbeef) out[8] = 1;
}

int main()
{
    unsigned long long* d; hipMalloc(&d, 16 * 8); hipMemset(d, 0, 16 * 8);
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, 1u);
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, 1u);
    hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d, 1u);
    hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, d, 1u);
    unsigned long long h[16]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[4] = { "64 lanes, one dword address", "lane 0 alone (exec mask)", "64 lanes, 64 consecutive dwords", "64 lanes, one 16-bit address" };
    for (int i = 0; i < 4; ++i) printf("%-36s %7.1f cycles per store (s_memtime ticks x 1: 100 MHz counter? see note)  raw %llu\n", names[i], (double)h[i] / (256.0 * 32.0), h[i]);
    printf("(s_memtime counts at a fixed rate; compare the rows with each other)\n");
    return 0;
}
