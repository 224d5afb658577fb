// Micro-benchmark: cost of a scalar "walk" iteration on gfx950 (cycles per iteration, s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ __launch_bounds__(64) void k_walk(const uint64_t* masks, const uint32_t* lens, uint64_t* out, int iters, int mode)
{
    __shared__ uint16_t T[8192 + 512];   // same LDS footprint as the encode kernel => same residency
    T[threadIdx.x] = 0;
    const int lane = threadIdx.x;
    uint64_t E = masks[blockIdx.x & 255];
    uint32_t la = lens[lane];
    uint64_t acc = 0;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    uint32_t events = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t pos = 0;
        uint64_t lits = 0, mst = 0;
        if (mode == 0) {
            for (;;) {
                const uint64_t Er = E & (~0ull << pos);
                if (!Er) break;
                const int e = __builtin_ctzll(Er);
                lits |= ((1ull << e) - 1) & (~0ull << pos);
                mst |= 1ull << e;
                pos = e + __builtin_amdgcn_readlane(la, e);
                events++;
                if (pos >= 64) break;
            }
        } else {
            // pure SALU dependent chain, no readlane, no inner branch besides the loop
            for (int j = 0; j < 6; ++j) {
                const uint64_t Er = E & (~0ull << pos);
                const int e = Er ? __builtin_ctzll(Er) : 63;
                lits |= ((1ull << e) - 1) & (~0ull << pos);
                mst |= 1ull << e;
                pos = (e + 3) & 63;
                events++;
            }
        }
        acc += lits ^ mst;
        E = (E << 1) | (E >> 63);
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[blockIdx.x * 3] = t1 - t0; out[blockIdx.x * 3 + 1] = events; out[blockIdx.x * 3 + 2] = acc; }
}

int main()
{
    const int nblk = 256 * 9;
    std::vector<uint64_t> masks(256);
    std::vector<uint32_t> lens(64);
    uint64_t s = 12345;
    for (auto& m : masks) { m = 0; for (int i = 0; i < 64; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; if ((s >> 60) < 3) m |= 1ull << i; } }
    for (auto& l : lens) { s = s * 6364136223846793005ull + 1442695040888963407ull; l = 4 + (s >> 62); }
    uint64_t *dm, *dout; uint32_t* dl;
    hipMalloc(&dm, 256 * 8); hipMalloc(&dl, 64 * 4); hipMalloc(&dout, nblk * 24);
    hipMemcpy(dm, masks.data(), 256 * 8, hipMemcpyHostToDevice);
    hipMemcpy(dl, lens.data(), 64 * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode)
        for (int grid : { 256, 256 * 9 }) {
            hipLaunchKernelGGL(k_walk, dim3(grid), dim3(64), 0, 0, dm, dl, dout, 2000, mode);
            hipDeviceSynchronize();
            std::vector<uint64_t> o(grid * 3);
            hipMemcpy(o.data(), dout, grid * 24, hipMemcpyDeviceToHost);
            double cyc = 0, ev = 0;
            for (int i = 0; i < grid; ++i) { cyc += o[i * 3]; ev += o[i * 3 + 1]; }
            printf("mode %d grid %5d: %.1f cycles/event, %.2f events/iter\n", mode, grid, cyc / ev, ev / grid / 2000);
        }
    return 0;
}
