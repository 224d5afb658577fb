// Probe: what does ONE wavefront's stream of LDS operations cost inside a 16-wavefront workgroup that owns 160 KiB of LDS
// (k_l6_matches' placer)? Cycles per LDS instruction (s_memtime around wavefront 0's work; 15 other wavefronts wait at the
// workgroup barrier, or run VALU work) for: returning adds, non-returning adds, 16-bit scattered stores, 32-bit stores, loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ROUNDS 64
template <int KIND, bool BUSY>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, uint32_t seed)
{
    __shared__ __attribute__((aligned(16))) uint16_t sorted[65536 + 64];
    __shared__ __attribute__((aligned(16))) uint32_t Tw[4096 + 4];
    __shared__ __attribute__((aligned(16))) uint32_t ring[2][30][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 1024) Tw[i] = 0;
    for (int i = threadIdx.x; i < 2 * 30 * 64; i += 1024) (&ring[0][0][0])[i] = (i * 2654435761u) >> 19;
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0, acc = 0;
    uint32_t x = seed + threadIdx.x * 977u, sink = 0;
    if (wave == 0) __builtin_amdgcn_s_setprio(3);
    for (int r = 0; r < ROUNDS; ++r) {
        if (wave == 0) {
            uint32_t a[30], o[30];
#pragma unroll
            for (int s = 0; s < 30; ++s) { x = x * 1664525u + 1013904223u; a[s] = (x >> 19) & 8191u; }
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
            if (KIND == 0) {
#pragma unroll
                for (int s = 0; s < 30; ++s) o[s] = atomicAdd(&Tw[a[s] >> 1], 1u << ((a[s] & 1) << 4));
#pragma unroll
                for (int s = 0; s < 30; ++s) sink += o[s];
            } else if (KIND == 1) {
#pragma unroll
                for (int s = 0; s < 30; ++s) atomicAdd(&Tw[a[s] >> 1], 1u << ((a[s] & 1) << 4));
            } else if (KIND == 2) {
#pragma unroll
                for (int s = 0; s < 30; ++s) sorted[8 + a[s] * 8 + (lane & 7)] = (uint16_t)(a[s] + s);
            } else if (KIND == 3) {
#pragma unroll
                for (int s = 0; s < 30; ++s) ring[r & 1][s][lane] = a[s];
            } else if (KIND == 4) {
#pragma unroll
                for (int s = 0; s < 30; ++s) o[s] = ring[r & 1][s][lane];
#pragma unroll
                for (int s = 0; s < 30; ++s) sink += o[s];
            } else if (KIND == 5) {
#pragma unroll
                for (int s = 0; s < 30; ++s) o[s] = ((uint16_t*)Tw)[a[s]];
#pragma unroll
                for (int s = 0; s < 30; ++s) sink += o[s];
            } else if (KIND == 6) {            // two adjacent 8-byte words at a random 8-aligned offset of a 64 KiB image (ds_read2_b64)
#pragma unroll
                for (int s = 0; s < 30; ++s) { const uint64_t* p = (const uint64_t*)((const uint8_t*)sorted + a[s] * 8); const uint64_t u = p[0], v = p[1]; o[s] = (uint32_t)(u ^ (v >> 7)); }
#pragma unroll
                for (int s = 0; s < 30; ++s) sink += o[s];
            } else if (KIND == 7) {            // one 8-byte word at a random 8-aligned offset
#pragma unroll
                for (int s = 0; s < 30; ++s) { const uint64_t* p = (const uint64_t*)((const uint8_t*)sorted + a[s] * 8); const uint64_t u = p[0]; o[s] = (uint32_t)(u ^ (u >> 37)); }
#pragma unroll
                for (int s = 0; s < 30; ++s) sink += o[s];
            } else if (KIND == 9) {            // sixteen bytes at a random 16-aligned offset (ds_read_b128)
#pragma unroll
                for (int s = 0; s < 30; ++s) { const uint4 u = *(const uint4*)((const uint8_t*)sorted + a[s] * 16); o[s] = u.x ^ u.y ^ u.z ^ u.w; }
#pragma unroll
                for (int s = 0; s < 30; ++s) sink += o[s];
            } else {                           // one 4-byte word at a random 4-aligned offset
#pragma unroll
                for (int s = 0; s < 30; ++s) o[s] = ((const uint32_t*)sorted)[a[s] * 2 + (lane & 1)];
#pragma unroll
                for (int s = 0; s < 30; ++s) sink += o[s];
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
            acc += t1 - t0;
        } else if (BUSY) {
            for (int i = 0; i < 200; ++i) { x = x * 1664525u + 1013904223u; sink ^= x >> 7; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
    if (sink == 0x12345u) out[0] = sink;
}

template <int KIND, bool BUSY> static void run(const char* name)
{
    unsigned long long* d; hipMalloc(&d, 256 * 8);
    hipLaunchKernelGGL((k<KIND, BUSY>), dim3(256), dim3(1024), 0, 0, d, 12345u);
    hipDeviceSynchronize();
    unsigned long long h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
    printf("%-44s others %s: %7.1f cycles per instruction\n", name, BUSY ? "busy  " : "at the barrier", s / 256 / ROUNDS / 30);
    hipFree(d);
}
int main()
{
    run<0, false>("ds_add_rtn_u32, 64 scattered counters"); run<0, true>("ds_add_rtn_u32, 64 scattered counters");
    run<1, false>("ds_add_u32 (no return)"); run<1, true>("ds_add_u32 (no return)");
    run<2, false>("ds_write_b16 scattered"); run<2, true>("ds_write_b16 scattered");
    run<3, false>("ds_write_b32 lane-contiguous"); run<3, true>("ds_write_b32 lane-contiguous");
    run<4, false>("ds_read_b32 lane-contiguous"); run<4, true>("ds_read_b32 lane-contiguous");
    run<5, false>("ds_read_u16 scattered"); run<5, true>("ds_read_u16 scattered");
    run<6, false>("ds_read2_b64 scattered, 8-aligned"); run<7, false>("ds_read_b64 scattered, 8-aligned"); run<8, false>("ds_read_b32 scattered"); run<9, false>("ds_read_b128 scattered, 16-aligned");
    return 0;
}
