// Probe: in which order does the LDS serve the lanes of ONE wavefront instruction that add to the SAME address?
// The counting sort of the extended levels (zz_level6.h) wants "ascending lane" -- then ds_add_rtn on a bucket's counter gives
// every position its place, and LDS instructions of one wavefront executing in issue order give the blocks theirs.
// For each trial a wavefront draws 64 keys with many duplicates (several distributions, among them keys that share a dword and
// keys that share a bank), adds 1 << 16*(key&1) to word[key>>1] with return, twice in a row (two "blocks" back to back without
// a wait in between), and the host compares every returned count with the lane-ascending, block-ascending one.
// Prints the number of lanes checked and of those that differ.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define NKEY 8192
__global__ __launch_bounds__(1024) void k_probe(const uint16_t* keys, uint16_t* ranks, int trials_per_wave, int busy_waves)
{
    __shared__ uint32_t T[16][NKEY / 2 / 16];      // a small table per wavefront: 512 keys each (keys are taken mod 512)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* mine = T[wave];
    for (int i = lane; i < NKEY / 2 / 16; i += 64) mine[i] = 0;
    __syncthreads();
    if (wave >= busy_waves) return;
    const size_t base = ((size_t)blockIdx.x * 16 + wave) * trials_per_wave;
    for (int t = 0; t < trials_per_wave; ++t) {
        const uint16_t* kk = keys + (base + t) * 128;
        const uint32_t ka = kk[lane] & 511u, kb = kk[64 + lane] & 511u;
        // two blocks back to back, no wait between them
        const uint32_t oa = atomicAdd(&mine[ka >> 1], 1u << ((ka & 1u) << 4));
        const uint32_t ob = atomicAdd(&mine[kb >> 1], 1u << ((kb & 1u) << 4));
        ranks[(base + t) * 128 + lane] = (uint16_t)(oa >> ((ka & 1u) << 4));
        ranks[(base + t) * 128 + 64 + lane] = (uint16_t)(ob >> ((kb & 1u) << 4));
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        // undo (so that counts stay small), in any order
        atomicSub(&mine[ka >> 1], 1u << ((ka & 1u) << 4));
        atomicSub(&mine[kb >> 1], 1u << ((kb & 1u) << 4));
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
}

int main()
{
    const int blocks = 256, tpw = 512;
    const size_t trials = (size_t)blocks * 16 * tpw;
    std::vector<uint16_t> keys(trials * 128), ranks(trials * 128);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); };
    for (size_t t = 0; t < trials; ++t) {
        const int kind = t % 8;
        for (int i = 0; i < 128; ++i) {
            uint32_t k;
            switch (kind) {
            case 0: k = rnd() % 512; break;                       // few duplicates
            case 1: k = rnd() % 16; break;                        // many
            case 2: k = rnd() % 3; break;                         // three values, two of them in one dword
            case 3: k = 7; break;                                 // all the same
            case 4: k = (rnd() % 8) * 64; break;                  // same bank, different addresses
            case 5: k = (rnd() % 4) * 64 + (rnd() & 1); break;    // same bank, halves of a dword
            case 6: k = (i & 1) ? 5 : rnd() % 512; break;         // one heavy key among light ones
            default: k = rnd() % 40; break;
            }
            keys[t * 128 + i] = (uint16_t)k;
        }
    }
    uint16_t *dk, *dr;
    hipMalloc(&dk, keys.size() * 2); hipMalloc(&dr, ranks.size() * 2);
    hipMemcpy(dk, keys.data(), keys.size() * 2, hipMemcpyHostToDevice);
    for (int busy : { 16, 1 }) {
        hipMemset(dr, 0xFF, ranks.size() * 2);
        hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(1024), 0, 0, dk, dr, tpw, busy);
        hipDeviceSynchronize();
        hipMemcpy(ranks.data(), dr, ranks.size() * 2, hipMemcpyDeviceToHost);
        size_t checked = 0, bad = 0;
        for (size_t t = 0; t < trials; ++t) {
            if ((int)((t / tpw) % 16) >= busy) continue;
            uint16_t cnt[512] = { 0 };
            for (int i = 0; i < 128; ++i) {
                const uint32_t k = keys[t * 128 + i] & 511u;
                if (ranks[t * 128 + i] != cnt[k]) {
                    if (bad < 5) printf("  trial %zu kind %zu lane %d: got %u, lane order gives %u\n", t, t % 8, i, ranks[t * 128 + i], cnt[k]);
                    ++bad;
                }
                ++cnt[k]; ++checked;
            }
        }
        printf("%d busy wavefronts per workgroup: %zu lanes checked, %zu out of lane order\n", busy, checked, bad);
    }
    return 0;
}
