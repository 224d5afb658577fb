// Micro-benchmark: how fast can ONE wavefront issue VALU / SALU instructions on gfx950, as a function of the number of
// independent dependency chains in its instruction stream (ILP) and of the number of wavefronts per SIMD?
// Prints cycles per instruction per wave (wall clock x 2.4 GHz / instructions executed by one wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int CHAINS, int KIND>
__global__ __launch_bounds__(64) void k_chain(uint32_t* out, int iters)
{
    extern __shared__ uint32_t pad[];
    uint32_t a = threadIdx.x, b = threadIdx.x + 1, c = threadIdx.x + 2, d = threadIdx.x + 3;
    uint32_t sa = blockIdx.x, sb = blockIdx.x + 1;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {           // VALU, 32 instructions per trip
            if (CHAINS == 1)
                asm volatile(
                    ".rept 32\n\tv_add_u32 %0, %0, %0\n\t.endr" : "+v"(a));
            else if (CHAINS == 2)
                asm volatile(
                    ".rept 16\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %1, %1, %1\n\t.endr" : "+v"(a), "+v"(b));
            else
                asm volatile(
                    ".rept 8\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %1, %1, %1\n\tv_add_u32 %2, %2, %2\n\tv_add_u32 %3, %3, %3\n\t.endr"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 1) {    // SALU
            if (CHAINS == 1)
                asm volatile(".rept 32\n\ts_add_u32 %0, %0, %0\n\t.endr" : "+s"(sa) :: "scc");
            else
                asm volatile(".rept 16\n\ts_add_u32 %0, %0, %0\n\ts_add_u32 %1, %1, %1\n\t.endr" : "+s"(sa), "+s"(sb) :: "scc");
        } else {                   // VALU chain interleaved with an independent SALU chain (16 + 16)
            asm volatile(".rept 16\n\tv_add_u32 %0, %0, %0\n\ts_add_u32 %1, %1, %1\n\t.endr" : "+v"(a), "+s"(sa) :: "scc");
        }
    }
    if (a + b + c + d + sa + sb == 0x12345) out[0] = 1;
    if (iters < 0) out[1] = pad[threadIdx.x];
}

template <int CHAINS, int KIND> void run(const char* name, int waves_per_cu, uint32_t* dout)
{
    const int iters = 20000;
    const int grid = 256 * waves_per_cu;
    // dynamic LDS sized so that exactly waves_per_cu single-wave workgroups fit a CU (160 KiB)
    const unsigned lds = waves_per_cu >= 32 ? 0 : (163840 / waves_per_cu - 256) & ~255u;
    hipFuncSetAttribute((const void*)k_chain<CHAINS, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const unsigned l = lds > 65536 ? 65536 : lds;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_chain<CHAINS, KIND>), dim3(grid), dim3(64), l, 0, dout, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_chain<CHAINS, KIND>), dim3(grid), dim3(64), l, 0, dout, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)iters * 32;
    printf("%-28s waves/CU %2d (lds %6u): %.2f cycles per instruction per wave\n", name, waves_per_cu, l, ms * 1e-3 * 2.4e9 / instr);
}

int main()
{
    uint32_t* dout; hipMalloc(&dout, 64);
    for (int w : { 1, 4, 8, 16, 32 }) {
        run<1, 0>("VALU 1 chain", w, dout);
        run<2, 0>("VALU 2 chains", w, dout);
        run<4, 0>("VALU 4 chains", w, dout);
        run<1, 1>("SALU 1 chain", w, dout);
        run<2, 1>("SALU 2 chains", w, dout);
        run<1, 2>("VALU+SALU interleaved", w, dout);
    }
    return 0;
}
