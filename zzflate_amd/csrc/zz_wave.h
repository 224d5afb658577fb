// zz_wave.h -- wavefront (64-lane) primitives for gfx950: ballots, DPP prefix scans, uniform reads,
// bounds-safe unaligned loads, and a wave-cooperative byte copy.
#pragma once
#include "zz_common.h"

namespace zz {

// Inside ONE wavefront, LDS instructions execute in issue order, so lane-to-lane hand-offs through LDS need no
// s_barrier and no s_waitcnt -- only the compiler must keep the program order of the LDS accesses. A wavefront-scope
// fence does exactly that and emits no instruction (in particular it does not drain outstanding global loads the way
// __syncthreads() does). The encode kernels run two wavefronts per workgroup (parser + emitter/helper); those two meet
// at explicit s_barriers (l1_group_barrier, l2_block_barrier), never through this macro.
#define ZZ_WAVE_SYNC()                                              \
    do {                                                            \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      \
        __builtin_amdgcn_wave_barrier();                            \
    } while (0)

__device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ uint32_t readlane(uint32_t v, int l)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(l));
}
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int l)
{
    uint32_t lo = readlane((uint32_t)v, l), hi = readlane((uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// ---- DPP scans --------------------------------------------------------------------------------
// update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl): lanes whose source lane is outside
// the row (or masked off) keep `old`; with old = 0 that is the additive identity.
#define ZZ_DPP_ROW_SHR(n) (0x110 + (n))
#define ZZ_DPP_ROW_BCAST15 0x142
#define ZZ_DPP_ROW_BCAST31 0x143

// inclusive prefix sum across the 64 lanes
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x)
{
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, ZZ_DPP_ROW_SHR(1), 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, ZZ_DPP_ROW_SHR(2), 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, ZZ_DPP_ROW_SHR(4), 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, ZZ_DPP_ROW_SHR(8), 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, ZZ_DPP_ROW_BCAST15, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, ZZ_DPP_ROW_BCAST31, 0xc, 0xf, false);
    return (uint32_t)v;
}

// wave-wide sums (result valid in every lane)
__device__ __forceinline__ uint32_t wave_sum(uint32_t x)
{
    return readlane(wave_scan_incl(x), 63);
}
__device__ __forceinline__ uint64_t wave_sum64(uint64_t x)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// Lanes holding the same 6-bit key as this lane (keys are lane ids, so six ballots decide it for all lanes at once).
// Per bit four instructions: "my bit as a mask" (v_bfe_i32), the ballot of that mask, and one v_bitop3 per half:
// ~(ballot ^ mybit) & set. The six ballots are issued back to back into six SGPR pairs: a VALU instruction that reads an
// SGPR the VALU wrote needs two wait states on gfx950, and ballot-then-use per bit would spend them on s_nops.
__device__ __forceinline__ uint64_t wave_match6(uint32_t key)
{
    uint32_t m0, m1, m2, m3, m4, m5;
    uint64_t b0, b1, b2, b3, b4, b5;
    asm("v_bfe_i32 %0, %12, 0, 1\n\tv_bfe_i32 %1, %12, 1, 1\n\tv_bfe_i32 %2, %12, 2, 1\n\t"
        "v_bfe_i32 %3, %12, 3, 1\n\tv_bfe_i32 %4, %12, 4, 1\n\tv_bfe_i32 %5, %12, 5, 1\n\t"
        "v_cmp_ne_u32_e64 %6, 0, %0\n\tv_cmp_ne_u32_e64 %7, 0, %1\n\tv_cmp_ne_u32_e64 %8, 0, %2\n\t"
        "v_cmp_ne_u32_e64 %9, 0, %3\n\tv_cmp_ne_u32_e64 %10, 0, %4\n\tv_cmp_ne_u32_e64 %11, 0, %5\n\t"
        "s_nop 1"                                                              // (the compiler's next instruction may read %11)
        : "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3), "=&v"(m4), "=&v"(m5),
          "=&s"(b0), "=&s"(b1), "=&s"(b2), "=&s"(b3), "=&s"(b4), "=&s"(b5)
        : "v"(key));
    uint32_t lo, hi;
    lo = __builtin_amdgcn_bitop3_b32((uint32_t)b0, m0, ~0u, 0x82);             // ~(a ^ b) & c
    hi = __builtin_amdgcn_bitop3_b32((uint32_t)(b0 >> 32), m0, ~0u, 0x82);
    lo = __builtin_amdgcn_bitop3_b32((uint32_t)b1, m1, lo, 0x82); hi = __builtin_amdgcn_bitop3_b32((uint32_t)(b1 >> 32), m1, hi, 0x82);
    lo = __builtin_amdgcn_bitop3_b32((uint32_t)b2, m2, lo, 0x82); hi = __builtin_amdgcn_bitop3_b32((uint32_t)(b2 >> 32), m2, hi, 0x82);
    lo = __builtin_amdgcn_bitop3_b32((uint32_t)b3, m3, lo, 0x82); hi = __builtin_amdgcn_bitop3_b32((uint32_t)(b3 >> 32), m3, hi, 0x82);
    lo = __builtin_amdgcn_bitop3_b32((uint32_t)b4, m4, lo, 0x82); hi = __builtin_amdgcn_bitop3_b32((uint32_t)(b4 >> 32), m4, hi, 0x82);
    lo = __builtin_amdgcn_bitop3_b32((uint32_t)b5, m5, lo, 0x82); hi = __builtin_amdgcn_bitop3_b32((uint32_t)(b5 >> 32), m5, hi, 0x82);
    return ((uint64_t)hi << 32) | lo;
}

// The same for keys of NB bits (the 13-bit hashes of the extended levels' counting sort): one ballot per bit. Left to the
// compiler's scheduling -- its callers are not on a packet's critical path.
template <int NB> __device__ __forceinline__ uint64_t wave_match_bits(uint32_t key)
{
    uint32_t lo = ~0u, hi = ~0u;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const uint32_t m = (uint32_t)((int32_t)(key << (31 - b)) >> 31);      // my bit b as a mask
        const uint64_t B = ballot(m != 0);
        lo = __builtin_amdgcn_bitop3_b32((uint32_t)B, m, lo, 0x82);            // ~(ballot ^ mybit) & set
        hi = __builtin_amdgcn_bitop3_b32((uint32_t)(B >> 32), m, hi, 0x82);
    }
    return ((uint64_t)hi << 32) | lo;
}

// A 64-bit scalar mask used directly as the lane predicate of a select (lane l takes `a` where bit l is set): one
// v_cndmask, where the compiler would shift the mask by the lane id and test a bit (three or four instructions).
// `m` must have been written by SCALAR instructions (inline-asm "=s" outputs, or uniform integer arithmetic): an SGPR
// written by the VALU needs two wait states before a VALU instruction reads it, and the compiler cannot see into the asm.
__device__ __forceinline__ uint32_t sel_lanes(uint64_t m, uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
__device__ __forceinline__ uint32_t keep_lanes(uint64_t m, uint32_t a)      // a where the bit is set, else 0
{
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(a), "s"(m));
    return r;
}

// ---- loads ------------------------------------------------------------------------------------
// gfx950 runs with unaligned global access enabled (amdhsa), so a byte-addressed 4/8-byte load is one
// global_load_dword/dwordx2. `end` is one past the last readable byte: a load that would cross it is
// assembled byte-wise and zero-padded (only the tail of the last packet ever takes that path).
__device__ __forceinline__ uint64_t load64(const uint8_t* p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ uint32_t load32(const uint8_t* p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint64_t load64_safe(const uint8_t* p, const uint8_t* end)
{
    if (p + 8 <= end) return load64(p);
    uint64_t v = 0;
    for (int i = 0; i < 8; ++i)
        if (p + i < end) v |= (uint64_t)p[i] << (8 * i);
    return v;
}
__device__ __forceinline__ uint32_t load32_safe(const uint8_t* p, const uint8_t* end)
{
    if (p + 4 <= end) return load32(p);
    uint32_t v = 0;
    for (int i = 0; i < 4; ++i)
        if (p + i < end) v |= (uint32_t)p[i] << (8 * i);
    return v;
}

// ---- cooperative byte copy ------------------------------------------------------------------------
// Copies n bytes src -> dst with `nthreads` threads (tid in [0,nthreads)); dst and src may have any
// alignment. Stores are 16-byte aligned dwordx4, loads are byte-addressed dwordx4.
__device__ __forceinline__ void coop_copy(uint8_t* dst, const uint8_t* src, uint64_t n, uint32_t tid,
                                          uint32_t nthreads)
{
    if (n == 0) return;
    uint64_t head = (16 - ((uintptr_t)dst & 15)) & 15;
    if (head > n) head = n;
    if (tid < head) dst[tid] = src[tid];
    uint64_t body = (n - head) >> 4;
    const uint8_t* s = src + head;
    uint4* d = (uint4*)(dst + head);
    uint64_t i = tid;
    // four independent 16-byte loads in flight per thread, then four aligned stores
    for (; i + 3 * (uint64_t)nthreads < body; i += 4 * (uint64_t)nthreads) {
        uint4 v0, v1, v2, v3;
        __builtin_memcpy(&v0, s + (i << 4), 16);
        __builtin_memcpy(&v1, s + ((i + nthreads) << 4), 16);
        __builtin_memcpy(&v2, s + ((i + 2 * (uint64_t)nthreads) << 4), 16);
        __builtin_memcpy(&v3, s + ((i + 3 * (uint64_t)nthreads) << 4), 16);
        d[i] = v0; d[i + nthreads] = v1; d[i + 2 * (uint64_t)nthreads] = v2; d[i + 3 * (uint64_t)nthreads] = v3;
    }
    for (; i < body; i += nthreads) {
        uint4 v;
        __builtin_memcpy(&v, s + (i << 4), 16);
        d[i] = v;
    }
    uint64_t done = head + (body << 4);
    uint64_t tail = n - done;
    if (tid < tail) dst[done + tid] = src[done + tid];
}

}  // namespace zz
