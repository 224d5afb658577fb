// Probe: what does a byte-addressed (misaligned) LDS read cost on gfx950? The compiler emits ds_read_b64 / ds_read_b32 for align-1
// accesses, and the results are right (k_l6_matches with -DZZ_L6_WORDS4=2 is bit-exact) -- but level 6 lost 22 % with them
// (profiles/r05_ab_l6_unaligned_reads_and_packed_key.txt). Here: 64 lanes gather at pseudo-random offsets of a 64 KiB image, as
// the candidates of k_l6_matches do, with the offset's low bits forced to a given misalignment; throughput form (32 independent
// reads in flight per measurement), one wavefront and sixteen wavefronts per workgroup; cycles per read instruction from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ub tools/ubench_lds_misaligned.hip && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint64_t __attribute__((aligned(1))) u64u;
typedef uint32_t __attribute__((aligned(1))) u32u;
typedef __attribute__((address_space(3))) const uint8_t* lds_bytes;

// KIND 0: aligned ds_read_b64 (offset & ~7)        1: ds_read_b64 at offset & ~7 | 4 (four-byte aligned)
//      2: ds_read_b64 at any byte offset           3: ds_read2_b32 + ds_read_b32 at offset & ~3 (k_l6_matches' default: three dwords)
//      4: aligned ds_read_b32                      5: ds_read_b32 at any byte offset
template <int KIND>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, uint32_t seed, int slot)
{
    __shared__ __attribute__((aligned(16))) uint8_t img[65536 + 64];
    const int tid = threadIdx.x;
    for (int i = tid; i < (65536 + 64) / 4; i += blockDim.x) ((uint32_t*)img)[i] = i * 2654435761u;
    __syncthreads();
    const lds_bytes L = (lds_bytes)(const uint8_t*)img;
    unsigned long long t0, t1, acc = 0;
    uint32_t x = seed + tid * 747796405u;
    uint64_t sink = 0;
    for (int r = 0; r < 128; ++r) {
        uint32_t off[32];
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            x = x * 1664525u + 1013904223u;
            uint32_t o = (x >> 16) & 65535u;
            if (KIND == 0) o &= ~7u;
            else if (KIND == 1) o = (o & ~7u) | 4u;
            else if (KIND == 3 || KIND == 4) o &= ~3u;
            off[s] = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            if (KIND <= 2) sink ^= *(__attribute__((address_space(3))) const u64u*)(L + off[s]);
            else if (KIND == 3) {
                const __attribute__((address_space(3))) uint32_t* p = (const __attribute__((address_space(3))) uint32_t*)(L + off[s]);
                sink ^= ((uint64_t)p[1] << 32 | p[0]) + p[2];
            }
            else sink ^= *(__attribute__((address_space(3))) const u32u*)(L + off[s]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        acc += t1 - t0;
    }
    if (tid == 0) out[slot] = acc;
    if ((uint32_t)sink == 0x89abcdefu) out[63] = 1;
}

template <int KIND>
static void run(unsigned long long* d, const char* name, int& slot)
{
    for (int threads : { 64, 1024 }) {
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, d, 1u, slot);
        unsigned long long h = 0; hipMemcpy(&h, d + slot, 8, hipMemcpyDeviceToHost);
        printf("%-58s %2d wavefront(s): %7.1f ticks per read instruction of one wavefront, %6.2f per instruction of the workgroup\n", name, threads / 64,
               (double)h / (128.0 * 32.0), (double)h / (128.0 * 32.0) / (threads / 64));
        ++slot;
    }
}

int main()
{
    unsigned long long* d; hipMalloc(&d, 64 * 8); hipMemset(d, 0, 64 * 8);
    int slot = 0;
    run<0>(d, "ds_read_b64, eight-byte aligned", slot);
    run<1>(d, "eight bytes, four-byte aligned (compiles to ds_read2_b32)", slot);
    run<2>(d, "ds_read_b64, any byte offset", slot);
    run<3>(d, "ds_read2_b32 + ds_read_b32, four-byte aligned (3 dwords)", slot);
    run<4>(d, "ds_read_b32, aligned", slot);
    run<5>(d, "ds_read_b32, any byte offset", slot);
    printf("(s_memtime counts at a fixed rate; compare the rows with each other. Offsets are pseudo-random over 64 KiB: bank conflicts as in a gather)\n");
    return 0;
}
