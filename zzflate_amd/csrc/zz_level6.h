// zz_level6.h -- the extended levels 4, 5, 6: bounded hash chains, one-step lazy matching, code lengths
// by package-merge. NOT in the reference, which has one slot per hash, a greedy parse, a frequency-floor length limiter and
// rejects level > 3 (encoder.h:41-43,76; encoder.cpp:388-424; huffman.cpp:122-154; zzflate.cpp:201,230-234); what this
// file must be bit-exact with is the definition of these levels in the oracle, which it restates for a GPU:
//
//   chains   every position q of the window in front of the packet and of the packet itself, up to target = n - 16, is
//            entered under a 13-bit hash of its FOUR bytes, ascending; the candidates of q are the nearest DEPTH earlier
//            positions with q's hash, as long as they are less than 32768 back. All positions are entered whatever the
//            parse does, so the chains are parse-independent -- and stored FLATTENED: the positions sorted by (hash,
//            position) in one array (a counting sort: histogram of the hashes, exclusive scan, then every block of 64
//            positions takes consecutive places in its buckets, in lane order). The DEPTH entries in front of a
//            position's own place are its chain, nearest last: one 2*DEPTH-byte read instead of DEPTH dependent hops.
//            Entries in front of a bucket's first belong to the bucket before (other four bytes: no match of four) or have
//            not been written yet (the array is zeroed per packet: position -32768, out of reach): the chain ends there.
//   match    16 bytes at q against 16 bytes at each candidate; the longest wins, the nearest among equals; >= 4 is a match;
//   lazy     q defers (stays a literal) if its match is shorter than 16 and q+1 has a longer one, unless (q & 63) == 63;
//   parse    greedy over the positions with a match that do not defer; a length of 16 is extended to its true value (<= 258)
//            only when the parse reaches it -- the only thing here that depends on the parse;
//   codes    optimal length-limited code lengths by package-merge in its list form (pm_lengths_w), wave-parallel: it replaces
//            the heap replay of levels 2,3 (which exists only because ties must fall as libstdc++'s heap lets them fall).
//
// Two kernels. Everything up to the parse is parse-independent and wants the whole sorted array (128 KiB of 16-bit entries,
// 65,536 scattered two-byte stores per packet) in LDS and then the 64 KiB of window and packet for the compares, i.e. a CU to
// itself: k_l6_matches, one packet per 16-wavefront workgroup, leaves per position "match length (lazy rule applied) and
// distance" in a 4-byte word. The parse, the records,
// the histograms, the codes and the emission are sequential per packet and want MANY packets per CU: the level-2 kernel
// (zz_level2.h), whose parser reads those words (l6_parse_pass) where levels 2,3 probe their table.
// (Round 3's first form did all of it in the level-2 kernel with the array in global memory: 291 bytes of fabric traffic per
// input byte, bound by that -- profiles/README.md.)
#pragma once

namespace zz {

#define ZZ_L6_BIAS 32768u                          // array entries are position + 32768
#define ZZ_L6_PAD 8u                               // entries in front of the sorted array (a chain read never starts below it)
#define ZZ_L6_SORT_ENTRIES (65536u + 64u)
#define ZZ_L6_SORT_BYTES (ZZ_L6_SORT_ENTRIES * 2u)  // positions sorted by (hash, position)
#define ZZ_L6_TAIL 16u                              // target = n - 16: the 16 bytes compared at a position lie inside the data
#define ZZ_L6_CAP 16u

__device__ __forceinline__ uint32_t l6_hash4(uint32_t four_bytes) { return (four_bytes * 2654435761u) >> (32 - ZZ_HASH_BITS); }
__device__ __forceinline__ uint32_t l6_target(uint32_t n) { return n > ZZ_L6_TAIL ? n - ZZ_L6_TAIL : 0u; }
__device__ __forceinline__ uint32_t l6_trips(uint32_t n) { return (l6_target(n) + 63u) >> 6; }

// ===== kernel 1: per position, the best match of its chain ====================================================================
// One packet per workgroup of 16 wavefronts, 160 KiB of LDS. Per packet:
//   1  all: array and counters to zero; histogram of the hashes of positions -W .. target-1 (LDS atomics on the packed 16-bit
//      counters); exclusive scan -> every bucket's first place;
//   2  the sort, in rounds of 30 blocks of 64 positions, ascending, one workgroup barrier per round, four rounds in flight:
//      - the 15 WORKER wavefronts prepare round t: the hashes of their two blocks, as the counter's address and shift -> ring;
//      - wavefront 0, the PLACER, takes round t-1 from the ring: one returning add per block on the buckets' counters, and what
//        it returned back into the ring slot -- nothing else. The LDS serves the lanes of one instruction that add to one
//        address in ascending lane order, and one wavefront's instructions in issue order (tools/ubench_lds_atomic_order.hip;
//        zz_debug_lds_atomic_order behind a GPU test), so what the add returns IS the position's place: no ranking of equal
//        hashes, no wait between blocks;
//      - the workers take round t-2's returns from the ring: sorted[place] = position;
//      - and, for the packet's own positions of round t-3, copy the chain -- the DEPTH entries in front of the place; all
//        earlier positions stand in the array since the barrier before, later ones land behind them or in other buckets -- to
//        the workgroup's scratch in global memory (2*DEPTH bytes per position, coalesced);
//   3  the array is dead: the window and the packet (<= 64 KiB) take its place in LDS, and all 16 wavefronts compare -- per
//      position 16 bytes against 16 bytes at each candidate, gathered from LDS (three or five words and a funnel shift each; from
//      global memory every candidate was a 128-byte line for 16 bytes of it, and the load path, not the CU, set the pace);
//      best of the chain, the lazy rule inside the block, one word per position.
// m[q] = length << 16 | distance for a position that starts a match if the parse reaches it, 0 otherwise.
#define ZZ_L6M_THREADS 1024u
#define ZZ_L6M_PLACERS 1u                           // wavefront 0 (two, each with the buckets of one parity, were slower: 34.6 against 30.7 ms at level 6)
#define ZZ_L6M_ROUND 30u                            // blocks per round: two per worker wavefront
struct zz_l6m_params {
    zz_packet_params pk;
    uint32_t* m;            // one word per input byte of the packets k0 .. k1-1
    uint16_t* chains;       // gridDim.x * 32768 * DEPTH entries
    uint32_t* work;         // packets handed out beyond the first gridDim.x (zero at launch)
    uint32_t k0, k1;
};

typedef __attribute__((address_space(3))) const uint8_t* zz_lds_bytes;
// 16 bytes at any byte offset of an LDS image, as two little-endian words
__device__ __forceinline__ void lds_gather16(zz_lds_bytes base, uint32_t off, uint64_t& lo, uint64_t& hi)
{
    typedef __attribute__((address_space(3))) const uint64_t* p8;
    const uint32_t a = off & ~7u, sh = (off & 3u) << 3;
    const bool up = (off & 4u) != 0;
    const uint64_t x0 = *(p8)(base + a), x1 = *(p8)(base + a + 8), x2 = *(p8)(base + a + 16);
    const uint32_t x0l = (uint32_t)x0, x0h = (uint32_t)(x0 >> 32), x1l = (uint32_t)x1, x1h = (uint32_t)(x1 >> 32), x2l = (uint32_t)x2, x2h = (uint32_t)(x2 >> 32);
    const uint32_t e0 = up ? x0h : x0l, e1 = up ? x1l : x0h, e2 = up ? x1h : x1l, e3 = up ? x2l : x1h, e4 = up ? x2h : x2l;
    const uint32_t w0 = __builtin_amdgcn_alignbit(e1, e0, sh), w1 = __builtin_amdgcn_alignbit(e2, e1, sh);
    const uint32_t w2 = __builtin_amdgcn_alignbit(e3, e2, sh), w3 = __builtin_amdgcn_alignbit(e4, e3, sh);
    lo = ((uint64_t)w1 << 32) | w0;
    hi = ((uint64_t)w3 << 32) | w2;
}

// the same in two steps, for a candidate: most differ from the position within their first eight bytes, and the third read
// and half of the shifting are only for those that do not. Returns the number of equal leading bytes, 16 = all.
// ZZ_L6_WORDS4 = 1 (default): the candidate's words are read at FOUR-byte alignment (ds_read2_b32 + ds_read_b32, then ds_read2_b32), so
// the five selects between the halves of 8-byte words go: the compare step is bound by the vector ALUs' issue rate (about 22 of
// their instructions per candidate; reading LESS from the LDS -- a four-byte filter in front, no reads behind a full sixteen -- was
// slower, fewer instructions is faster: profiles/r05_ab_l6_*.txt). 0 = round 3's form, two or three aligned 8-byte reads.
#ifndef ZZ_L6_WORDS4
#define ZZ_L6_WORDS4 1
#endif
// ZZ_L6_WORDS4 = 2: ONE byte-addressed ds_read_b64 per eight bytes and no funnel shifts (three vector instructions fewer per candidate):
// bit-exact and 22 % SLOWER at level 6 (43.0 -> 33.5 GB/s): the LDS serves a misaligned 8-byte read at a fraction of an aligned one's rate
// (profiles/r05_ab_l6_unaligned_reads_and_packed_key.txt). ZZ_L6_KEY = 1 (default): the chain's best as ONE max over
// length << 16 | 32767 - distance instead of compare + max + select: + 0.4 %.
#ifndef ZZ_L6_KEY
#define ZZ_L6_KEY 1
#endif
__device__ __forceinline__ uint32_t lds_match16(zz_lds_bytes base, uint32_t off, uint64_t w, uint64_t w2)
{
#if ZZ_L6_WORDS4 == 2
    // the candidate's bytes by ONE byte-addressed 8-byte LDS read (the LDS serves unaligned reads: the compiler emits ds_read_b64 for an
    // align-1 access on gfx950), no funnel shifts, no address arithmetic
    typedef uint64_t __attribute__((aligned(1))) u64u;
    typedef __attribute__((address_space(3))) const u64u* p8u;
    const uint64_t x = *(p8u)(base + off);
    const uint32_t d0 = (uint32_t)x ^ (uint32_t)w, d1 = (uint32_t)(x >> 32) ^ (uint32_t)(w >> 32);
    const uint32_t f0 = ffbl_or_ones(d0), f1 = add_sat_k<32>(ffbl_or_ones(d1));
    uint32_t bits = f0 < f1 ? f0 : f1;                                   // >= 64: the first eight bytes are equal
    if (bits >= 64u) {
        const uint64_t y = *(p8u)(base + off + 8u);
        const uint32_t d2 = (uint32_t)y ^ (uint32_t)w2, d3 = (uint32_t)(y >> 32) ^ (uint32_t)(w2 >> 32);
        const uint32_t f2 = ffbl_or_ones(d2), f3 = add_sat_k<32>(ffbl_or_ones(d3));
        const uint32_t t = f2 < f3 ? f2 : f3;
        bits = 64u + (t < 64u ? t : 64u);
    }
    return bits >> 3;
#elif ZZ_L6_WORDS4
    typedef __attribute__((address_space(3))) const uint32_t* p4;
    const p4 p = (p4)(base + (off & ~3u));
    const uint32_t sh = (off & 3u) << 3;
    const uint32_t e0 = p[0], e1 = p[1], e2 = p[2];
    const uint32_t d0 = __builtin_amdgcn_alignbit(e1, e0, sh) ^ (uint32_t)w, d1 = __builtin_amdgcn_alignbit(e2, e1, sh) ^ (uint32_t)(w >> 32);
    const uint32_t f0 = ffbl_or_ones(d0), f1 = add_sat_k<32>(ffbl_or_ones(d1));
    uint32_t bits = f0 < f1 ? f0 : f1;                                   // >= 64: the first eight bytes are equal
    if (bits >= 64u) {
        const uint32_t e3 = p[3], e4 = p[4];
        const uint32_t d2 = __builtin_amdgcn_alignbit(e3, e2, sh) ^ (uint32_t)w2, d3 = __builtin_amdgcn_alignbit(e4, e3, sh) ^ (uint32_t)(w2 >> 32);
        const uint32_t f2 = ffbl_or_ones(d2), f3 = add_sat_k<32>(ffbl_or_ones(d3));
        const uint32_t t = f2 < f3 ? f2 : f3;
        bits = 64u + (t < 64u ? t : 64u);
    }
    return bits >> 3;
#else
    typedef __attribute__((address_space(3))) const uint64_t* p8;
    const uint32_t a = off & ~7u, sh = (off & 3u) << 3;
    const bool up = (off & 4u) != 0;
    const uint64_t x0 = *(p8)(base + a), x1 = *(p8)(base + a + 8);
    const uint32_t x0l = (uint32_t)x0, x0h = (uint32_t)(x0 >> 32), x1l = (uint32_t)x1, x1h = (uint32_t)(x1 >> 32);
    const uint32_t e0 = up ? x0h : x0l, e1 = up ? x1l : x0h, e2 = up ? x1h : x1l;
    const uint32_t d0 = __builtin_amdgcn_alignbit(e1, e0, sh) ^ (uint32_t)w, d1 = __builtin_amdgcn_alignbit(e2, e1, sh) ^ (uint32_t)(w >> 32);
    const uint32_t f0 = ffbl_or_ones(d0), f1 = add_sat_k<32>(ffbl_or_ones(d1));
    uint32_t bits = f0 < f1 ? f0 : f1;                                   // >= 64: the first eight bytes are equal
    if (bits >= 64u) {
        const uint64_t x2 = *(p8)(base + a + 16);
        const uint32_t x2l = (uint32_t)x2, x2h = (uint32_t)(x2 >> 32);
        const uint32_t e3 = up ? x2l : x1h, e4 = up ? x2h : x2l;
        const uint32_t d2 = __builtin_amdgcn_alignbit(e3, e2, sh) ^ (uint32_t)w2, d3 = __builtin_amdgcn_alignbit(e4, e3, sh) ^ (uint32_t)(w2 >> 32);
        const uint32_t f2 = ffbl_or_ones(d2), f3 = add_sat_k<32>(ffbl_or_ones(d3));
        const uint32_t t = f2 < f3 ? f2 : f3;
        bits = 64u + (t < 64u ? t : 64u);
    }
    return bits >> 3;
#endif
}

template <int DEPTH>
__global__ __launch_bounds__(ZZ_L6M_THREADS) void k_l6_matches(zz_l6m_params Q)
{
    const zz_packet_params& P = Q.pk;
    __shared__ __attribute__((aligned(16))) uint16_t sorted[ZZ_L6_SORT_ENTRIES];     // step 3: the window and the packet, as bytes
    __shared__ __attribute__((aligned(16))) uint32_t Tw[ZZ_HASH_SIZE / 2 + 4];      // 8192 16-bit counters, two per word (+ the one lanes without a position use)
    __shared__ __attribute__((aligned(16))) uint32_t ring[2][ZZ_L6M_ROUND][ZZ_WAVE];
    __shared__ uint32_t wtot[ZZ_L6M_THREADS / ZZ_WAVE];
    __shared__ uint32_t nextk, nextblk;
    const uint32_t tid = threadIdx.x;
    const int lane = lane_id();
    const uint32_t wave = uniform(tid >> 6);
    uint16_t* const chains = Q.chains + (uint64_t)blockIdx.x * (32768u * DEPTH);
    ZZ_PROF_DECL

    uint32_t k = Q.k0 + blockIdx.x;
    while (k < Q.k1) {
        const uint64_t off = (uint64_t)k * P.packet_size;
        const uint32_t len = (uint32_t)((P.n - off) < P.packet_size ? (P.n - off) : P.packet_size);
        const bool is_final = P.last_is_final && k == P.npk - 1;
        const uint32_t n = is_final ? len : len - 1;      // bytes of the compressing AddData (zz_level2.h)
        const uint8_t* src = P.src + off;
        const uint64_t before = P.halo + off;
        const int32_t W = (int32_t)(before < P.warm ? before : P.warm);
        const uint32_t target = l6_target(n);
        uint32_t* const mrow = Q.m + (uint64_t)(k - Q.k0) * P.packet_size;
        if (target) {
            const int32_t lo = -W, hi = (int32_t)target;
            ZZ_T(13);
            // ---- 1: zero, histogram, scan ----
            {
                uint4* z = (uint4*)sorted;
                const uint32_t n16 = ((ZZ_L6_PAD + (uint32_t)W + target) * 2u + 15u) >> 4;
                for (uint32_t i = tid; i < n16; i += ZZ_L6M_THREADS) z[i] = make_uint4(0, 0, 0, 0);
                ((uint4*)Tw)[tid] = make_uint4(0, 0, 0, 0);
            }
            __syncthreads();
            ZZ_T(6);
            for (int32_t p0 = lo + (int32_t)tid; p0 < hi; p0 += 8 * (int32_t)ZZ_L6M_THREADS) {      // eight loads in flight
                uint32_t v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int32_t pos = p0 + u * (int32_t)ZZ_L6M_THREADS;
                    v[u] = load32(src + (pos < hi ? pos : lo));
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int32_t pos = p0 + u * (int32_t)ZZ_L6M_THREADS;
                    const uint32_t h = l6_hash4(v[u]);
                    if (pos < hi) atomicAdd(&Tw[h >> 1], 1u << ((h & 1u) << 4));
                }
            }
            __syncthreads();
            ZZ_T(7);
            {   // exclusive scan over the 8192 counters: a thread owns eight consecutive ones (W + target < 65536: places fit 16 bits)
                uint4 w = ((uint4*)Tw)[tid];
                const uint32_t c0 = w.x & 0xFFFF, c1 = w.x >> 16, c2 = w.y & 0xFFFF, c3 = w.y >> 16;
                const uint32_t c4 = w.z & 0xFFFF, c5 = w.z >> 16, c6 = w.w & 0xFFFF, c7 = w.w >> 16;
                const uint32_t s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
                const uint32_t incl = wave_scan_incl(s);
                if (lane == 63) wtot[wave] = incl;
                __syncthreads();
                uint32_t e = incl - s;
                for (uint32_t i = 0; i < wave; ++i) e += wtot[i];
                const uint32_t a1 = e + c0, a2 = a1 + c1, a3 = a2 + c2, a4 = a3 + c3, a5 = a4 + c4, a6 = a5 + c5, a7 = a6 + c6;
                w.x = (e & 0xFFFF) | (a1 << 16); w.y = (a2 & 0xFFFF) | (a3 << 16);
                w.z = (a4 & 0xFFFF) | (a5 << 16); w.w = (a6 & 0xFFFF) | (a7 << 16);
                ((uint4*)Tw)[tid] = w;
            }
            __syncthreads();
            ZZ_T(8);
            // ---- 2: the rounds ----
            // (blocks are aligned to the packet: the first window block may be partial; blocks below 0 hold window positions only)
            const uint32_t Wr = ((uint32_t)W + 63u) & ~63u;
            const int32_t g0 = -(int32_t)Wr;
            const uint32_t nblk = (Wr + ((target + 63u) & ~63u)) >> 6;
            const uint32_t NR = (nblk + ZZ_L6M_ROUND - 1u) / ZZ_L6M_ROUND;
            uint32_t vnext[2];                         // workers: the four bytes at their positions of the next round
            auto fetch = [&](uint32_t t, uint32_t (&v)[2]) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const uint32_t bi = t * ZZ_L6M_ROUND + 2u * (wave - ZZ_L6M_PLACERS) + (uint32_t)u;
                    const int32_t pos = g0 + (int32_t)(bi << 6) + lane;
                    v[u] = load32(src + ((bi < nblk && pos >= lo && pos < hi) ? pos : lo));
                }
            };
            // the placer's round: thirty returning adds, thirty stores of what they returned -- nothing else
            auto place_round = [&](uint32_t r) {
                uint32_t* const slots = &ring[r & 1][0][0];
                uint32_t inf[ZZ_L6M_ROUND], old[ZZ_L6M_ROUND];
#pragma unroll
                for (uint32_t s = 0; s < ZZ_L6M_ROUND; ++s) inf[s] = slots[s * ZZ_WAVE + (uint32_t)lane];
#pragma unroll
                for (uint32_t s = 0; s < ZZ_L6M_ROUND; ++s)
                    old[s] = atomicAdd((uint32_t*)((uint8_t*)Tw + (inf[s] & 0xFFFFu)), 1u << (inf[s] >> 16));   // (a lane without a position: a counter of its own)
#pragma unroll
                for (uint32_t s = 0; s < ZZ_L6M_ROUND; ++s) slots[s * ZZ_WAVE + (uint32_t)lane] = old[s];
            };
            if (wave >= ZZ_L6M_PLACERS) fetch(0, vnext);
            else __builtin_amdgcn_s_setprio(3);        // the placer's stream of adds is the serial part of a packet
            uint32_t inf1[2] = { 0x80000000u, 0x80000000u }, inf2[2] = { 0x80000000u, 0x80000000u };   // workers: what they sent the placer one / two rounds ago
            uint32_t plc[2] = { 0, 0 };                                                                // ... and the places of their blocks of three rounds ago
            for (uint32_t t = 0; t <= NR + 2; ++t) {
                if (wave < ZZ_L6M_PLACERS) {
                    if (t >= 1 && t <= NR) place_round(t - 1);
                } else {
                    const uint32_t s0 = 2u * (wave - ZZ_L6M_PLACERS);
                    // round t-3: the chains of the packet's own positions (every earlier position stands in the array since the last barrier)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const uint32_t bi = (t - 3) * ZZ_L6M_ROUND + s0 + (uint32_t)u;
                        if (t >= 3 && bi < nblk && bi >= (Wr >> 6)) {
                            const uint32_t q = ((bi - (Wr >> 6)) << 6) + (uint32_t)lane;
                            if (q < target) {
                                // (entries plc - DEPTH .. plc - 1, nearest last: 2*DEPTH bytes at two-byte alignment -- read as aligned
                                // words and shifted; a 16-byte LDS read off its alignment is replayed at 64 cycles)
                                uint64_t c[2];
                                lds_gather16((zz_lds_bytes)(const uint8_t*)sorted, 2u * (ZZ_L6_PAD + plc[u] - DEPTH), c[0], c[1]);
                                __builtin_memcpy(chains + (uint64_t)q * DEPTH, c, 2 * DEPTH);
                            }
                        }
                    }
                    // round t-2: what the placer's adds returned is the place; the position goes into the array
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const uint32_t bi = (t - 2) * ZZ_L6M_ROUND + s0 + (uint32_t)u;
                        if (t >= 2 && bi < nblk) {
                            const uint32_t old = ring[t & 1][s0 + u][lane];
                            plc[u] = __builtin_amdgcn_ubfe(old, inf2[u] >> 16, 16);
                            if ((int32_t)inf2[u] >= 0) sorted[ZZ_L6_PAD + plc[u]] = (uint16_t)(g0 + (int32_t)(bi << 6) + lane + (int32_t)ZZ_L6_BIAS);
                        }
                    }
                    ZZ_WAVE_SYNC();
                    // round t: the hashes, as the placer wants them: byte address of the counter's word | shift << 16
#pragma unroll
                    for (int u = 0; u < 2; ++u) { inf2[u] = inf1[u]; inf1[u] = 0x80000000u | (ZZ_HASH_SIZE * 2u); }
                    if (t < NR) {
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const uint32_t bi = t * ZZ_L6M_ROUND + s0 + (uint32_t)u;
                            const int32_t pos = g0 + (int32_t)(bi << 6) + lane;
                            const bool act = bi < nblk && pos >= lo && pos < hi;
                            const uint32_t h = l6_hash4(vnext[u]);
                            if (act) inf1[u] = ((h >> 1) << 2) | ((h & 1u) << 20);
                            ring[t & 1][s0 + u][lane] = inf1[u];
                        }
                        fetch(t + 1, vnext);
                    }
                }
                __syncthreads();
                if (t * ZZ_L6M_ROUND < (Wr >> 6) + 3 * ZZ_L6M_ROUND) { ZZ_T(9); } else { ZZ_T(12); }   // rounds without / with chain copies
            }
            if (wave < ZZ_L6M_PLACERS) __builtin_amdgcn_s_setprio(0);
            // ---- 3: the window and the packet into LDS (position p at byte 32768 + p: an array entry IS its position's offset) ----
            {
                uint8_t* const L = (uint8_t*)sorted;
                const uint64_t avail64 = (uint64_t)((P.src + P.n) - src);
                const int32_t avail = (int32_t)(avail64 < 65536u ? avail64 : 65536u);           // bytes readable from src on
                const int32_t c0 = (32768 - W) >> 4, c1 = (32768 + (int32_t)n + 8 + 15) >> 4;
                for (int32_t c = c0 + (int32_t)tid; c < c1; c += (int32_t)ZZ_L6M_THREADS) {
                    const int32_t p = (c << 4) - 32768;
                    uint4 v;
                    if (p >= -W && p + 16 <= avail) __builtin_memcpy(&v, src + p, 16);
                    else {
                        uint8_t bts[16];
                        for (int i = 0; i < 16; ++i) bts[i] = (p + i >= -W && p + i < avail) ? src[p + i] : (uint8_t)0;
                        __builtin_memcpy(&v, bts, 16);
                    }
                    *(uint4*)(L + ((uint32_t)c << 4)) = v;
                }
            }
            if (tid == 0) nextblk = ZZ_L6M_THREADS / ZZ_WAVE;      // the blocks of the compare step beyond every wavefront's first
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // the chains: no stale line of the previous packet's in the L1
            ZZ_T(15);
            {
                const zz_lds_bytes L = (zz_lds_bytes)(const uint8_t*)sorted;
                const uint32_t npb = (target + 63u) >> 6;
                struct chain_t { uint16_t c[DEPTH]; };
                auto ldchain = [&](uint32_t b) -> chain_t {
                    const uint32_t q = (b << 6) + (uint32_t)lane;
                    chain_t ch;
                    __builtin_memcpy(ch.c, chains + (uint64_t)(q < target ? q : 0u) * DEPTH, 2 * DEPTH);
                    return ch;
                };
                // blocks come from a counter (their costs differ with the data: with a fixed share per wavefront the slowest one took
                // half as long again as the fastest), and a block's chains are requested one block ahead of their use
                auto grab = [&]() -> uint32_t {
                    uint32_t v = 0;
                    if (lane == 0) v = atomicAdd(&nextblk, 1u);
                    return uniform(v);
                };
                uint32_t b = wave, bn = b < npb ? grab() : npb;
                chain_t nextc = ldchain(b < npb ? b : 0u);
                for (; b < npb; b = bn, bn = bn < npb ? grab() : npb) {
                    const uint32_t q = (b << 6) + (uint32_t)lane;
                    const bool act = q < target;
                    const uint32_t qa = act ? q : 0u;
                    const chain_t cur = nextc;
                    nextc = ldchain(bn < npb ? bn : b);
                    const uint16_t* c = cur.c;
                    uint64_t w, w2;
                    lds_gather16(L, ZZ_L6_BIAS + qa, w, w2);                        // q + 16 <= n: inside the data
                    uint32_t best = 0, bdist = 0;
#if ZZ_L6_KEY
                    // longest wins, the nearest among equals: one max over length << 16 | 32767 - distance (distances grow along the chain)
                    uint32_t key = 0;
                    const uint32_t K = 32767u - (ZZ_L6_BIAS + q);
#pragma unroll
                    for (int i = 0; i < DEPTH; ++i) {                               // i = 0: the nearest
                        const uint32_t co = c[DEPTH - 1 - i];                      // the candidate's offset in the image
                        const uint32_t e = co + K;                                  // 32767 - d; d < 32768 as below
                        if (act && e < 32768u) {
                            const uint32_t k = (lds_match16(L, co, w, w2) << 16) | e;
                            key = k > key ? k : key;
                        }
                    }
                    best = key >> 16; bdist = 32767u - (key & 0xFFFFu);
#else
#pragma unroll
                    for (int i = 0; i < DEPTH; ++i) {                               // i = 0: the nearest
                        const uint32_t co = c[DEPTH - 1 - i];                      // the candidate's offset in the image
                        const uint32_t d = (ZZ_L6_BIAS + q) - co;                   // 0 < d < 32768: a candidate (d != 0: the entries
                        if (act && d < 32768u) {                                    // in front of q's place are other positions)
                            const uint32_t ln = lds_match16(L, co, w, w2);
                            if (ln > best) { best = ln; bdist = d; }
                        }
                    }
#endif
                    if (best < 4) best = 0;
                    // lazy: the next position's length (lane 63 never defers)
                    const uint32_t nxt = (uint32_t)__shfl_down((int)best, 1);
                    const bool defer = best != 0 && best < ZZ_L6_CAP && lane != 63 && nxt > best;
                    if (act) mrow[q] = (best != 0 && !defer) ? (best << 16) | bdist : 0u;
                }
            }
            ZZ_T(14);
        }
        __syncthreads();
        if (tid == 0) nextk = Q.k0 + gridDim.x + atomicAdd(Q.work, 1u);
        __syncthreads();
        k = nextk;
    }
    ZZ_PROF_FLUSH(P);
}

// ===== kernel 2's parser: the greedy parse over those words; hands every block's matches to the helper wavefront in the
// level-2 format (zz_level2.h: ZZ_L2_HB_PACK, one s_barrier per block) =========================================================
__device__ __forceinline__ void l6_parse_pass(uint32_t* hb, const uint8_t* src, const uint8_t* end, const l1p_src& TS, uint32_t n, const uint32_t* mrow)
{
    const int lane = lane_id();
    const uint32_t target = l6_target(n);
    const uint32_t nblk = (target + 63u) >> 6;
    uint32_t nextpos = 0;                   // first position the parse has not decided
    uint32_t slotsel = 0;
    auto ldm = [&](uint32_t b) -> uint32_t {
        const uint32_t q = (b << 6) + (uint32_t)lane;
        return (b < nblk && q < target) ? mrow[q] : 0u;
    };
    if (nblk == 0) return;
    uint32_t m0 = ldm(0), m1 = ldm(1), m2 = ldm(2);        // three blocks' words in flight
    for (uint32_t b = 0; b < nblk; ++b) {
        const uint32_t mN = ldm(b + 3);
        const uint32_t base = b << 6;
        const uint32_t q = base + (uint32_t)lane;
        const uint32_t best = m0 >> 16, bdist = m0 & 0xFFFFu;
        const uint64_t E = ballot(best != 0);
        // the parse: from match to match
        uint64_t evmask = 0;
        uint32_t tlen = best;               // per lane: the match's length (extended where the parse reached a capped one)
        uint32_t p = nextpos > base ? nextpos - base : 0u;
        while (p < 64) {
            const uint64_t m = E & (~0ull << p);
            if (!m) break;
            const int e = __builtin_ctzll(m);
            uint32_t len = readlane(best, e);
            if (len == ZZ_L6_CAP) {
                const uint32_t qe = base + (uint32_t)e;
                const uint32_t maxlen = (n - qe) < ZZ_MAX_LEN ? (n - qe) : ZZ_MAX_LEN;
                len = l1p_extend_match(TS, qe, (int32_t)(qe - readlane(bdist, e)), maxlen, ZZ_L6_CAP);   // (4-byte loads: may look 3 bytes past n)
                if (lane == e) tlen = len;
            }
            evmask |= 1ull << e;
            p = (uint32_t)e + len;
        }
        nextpos = base + (p < 64 ? 64u : p);                 // (no match left in this block: the rest are literals)
        // hand-over (every lane stores: no lane mask to set up)
        uint32_t* slot = hb + slotsel;
        slotsel ^= ZZ_L2_HB_WORDS;
        slot[lane] = ZZ_L2_HB_PACK(q, tlen < 3 ? 3u : tlen, bdist, base);
        slot[64] = (uint32_t)evmask; slot[65] = (uint32_t)(evmask >> 32);
        slot[66] = (uint32_t)evmask; slot[67] = (uint32_t)(evmask >> 32);     // all of them carry start, length, distance
        slot[68] = 0;
        l2_block_barrier();
        m0 = m1; m1 = m2; m2 = mN;
    }
}

// ---- package-merge, wave-parallel ------------------------------------------------------------------------------------------
// Optimal code lengths <= maxlen for the symbols with a non-zero count (the oracle restates it in the same list form, with the same
// ties): leaves = those symbols sorted by (count, symbol); list 1 = the leaves; list l+1 = merge(leaves, packages of list l),
// a leaf first where weights are equal, cut at 2m-2 items; from the last list the first 2m-2 items are taken, from every
// earlier list two per package taken from its successor; a symbol's length = the number of lists its leaf was taken from. Every
// list holds the leaves in the same order, so "the leaves taken from list l" is a count a_l and the symbol of rank r gets
// #{l : r < a_l}. Per list: packages = sums of pairs (parallel over pairs), the merged place of a leaf = its rank + the packages
// strictly lighter (binary search), of a package = its index + the leaves not heavier; one bit per place says "leaf".
struct pm_scratch {
    uint16_t* sym;      // [288] symbols by rank
    uint32_t* w;        // [288] their counts
    uint32_t* pk;       // [288] packages of the current list (also the sort keys before the lists exist)
    uint32_t* cur;      // [576] the current list's weights
    uint64_t* bm;       // [16 * 9] per list: bit k = item k is a leaf
    uint32_t* misc;     // [32] list lengths [0..16), leaves taken a_l [16..32)
};
__device__ inline void pm_lengths_w(pm_scratch& S, const uint32_t* freqs, int n, int maxlen, uint8_t* out)
{
    const int lane = lane_id();
    // the non-zero symbols, in symbol order, as keys (count << 9 | symbol)
    uint32_t m = 0;
    for (int i0 = 0; i0 < n; i0 += ZZ_WAVE) {
        const int i = i0 + lane;
        const uint32_t f = i < n ? freqs[i] : 0;
        const uint64_t nz = ballot(f != 0);
        if (f != 0) S.pk[m + mbcnt(nz)] = (f << 9) | (uint32_t)i;
        if (i < n) out[i] = 0;
        m += (uint32_t)__builtin_popcountll(nz);
    }
    ZZ_WAVE_SYNC();
    if (m == 0) return;
    if (m == 1) { if (lane == 0) out[S.pk[0] & 511u] = 1; ZZ_WAVE_SYNC(); return; }
    // rank sort (the keys are distinct)
    for (uint32_t t0 = 0; t0 < m; t0 += ZZ_WAVE) {
        const uint32_t t = t0 + (uint32_t)lane;
        const uint32_t key = t < m ? S.pk[t] : 0xFFFFFFFFu;
        uint32_t rank = 0;
        for (uint32_t k = 0; k < m; ++k) rank += S.pk[k] < key ? 1u : 0u;
        if (t < m) { S.sym[rank] = (uint16_t)(key & 511u); S.w[rank] = key >> 9; S.cur[rank] = key >> 9; }
    }
    for (int i = lane; i < 16 * 9; i += ZZ_WAVE) S.bm[i] = 0;
    ZZ_WAVE_SYNC();
    const uint32_t lim = 2 * m - 2;
    // list 1 = the leaves
    for (uint32_t k = (uint32_t)lane; k < m; k += ZZ_WAVE) atomicOr((unsigned long long*)&S.bm[k >> 6], 1ull << (k & 63));
    if (lane == 0) S.misc[0] = m;
    uint32_t ncur = m;
    for (int l = 1; l < maxlen; ++l) {
        const uint32_t np = ncur >> 1;
        ZZ_WAVE_SYNC();
        for (uint32_t b = (uint32_t)lane; b < np; b += ZZ_WAVE) S.pk[b] = S.cur[2 * b] + S.cur[2 * b + 1];
        ZZ_WAVE_SYNC();
        uint64_t* bm = S.bm + 9 * l;
        for (uint32_t a0 = 0; a0 < m; a0 += ZZ_WAVE) {                      // leaves: packages strictly lighter go first
            const uint32_t a = a0 + (uint32_t)lane;
            if (a < m) {
                const uint32_t wa = S.w[a];
                uint32_t lo = 0, hi = np;
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (S.pk[mid] < wa) lo = mid + 1; else hi = mid; }
                const uint32_t pos = a + lo;
                if (pos < lim) { S.cur[pos] = wa; atomicOr((unsigned long long*)&bm[pos >> 6], 1ull << (pos & 63)); }
            }
        }
        for (uint32_t b0 = 0; b0 < np; b0 += ZZ_WAVE) {                     // packages: leaves not heavier go first
            const uint32_t b = b0 + (uint32_t)lane;
            if (b < np) {
                const uint32_t wp = S.pk[b];
                uint32_t lo = 0, hi = m;
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (S.w[mid] <= wp) lo = mid + 1; else hi = mid; }
                const uint32_t pos = b + lo;
                if (pos < lim) S.cur[pos] = wp;
            }
        }
        ncur = m + np < lim ? m + np : lim;
        if (lane == 0) S.misc[l] = ncur;
    }
    ZZ_WAVE_SYNC();
    // how many leaves are taken from each list, last list first
    uint32_t need = lim;
    for (int l = maxlen - 1; l >= 0; --l) {
        const uint32_t len = S.misc[l];
        if (need > len) need = len;
        uint32_t cnt = 0;
        if (lane < 9) {
            const uint64_t wbits = S.bm[9 * l + lane];
            const int32_t rem = (int32_t)need - 64 * lane;
            const uint64_t mask = rem >= 64 ? ~0ull : rem <= 0 ? 0ull : ((1ull << rem) - 1);
            cnt = (uint32_t)__builtin_popcountll(wbits & mask);
        }
        const uint32_t a = wave_sum(cnt);
        if (lane == 0) S.misc[16 + l] = a;
        need = 2 * (need - a);
    }
    ZZ_WAVE_SYNC();
    for (uint32_t r = (uint32_t)lane; r < m; r += ZZ_WAVE) {
        uint32_t len = 0;
        for (int l = 0; l < maxlen; ++l) len += r < S.misc[16 + l] ? 1u : 0u;
        out[S.sym[r]] = (uint8_t)len;
    }
    ZZ_WAVE_SYNC();
}

}  // namespace zz
