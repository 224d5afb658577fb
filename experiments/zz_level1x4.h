// zz_level1x4.h -- level 1 with FOUR wavefronts per packet.
//
// Same algorithm and same bytes as zz_level1.h (WriteBlockFixedHuff, encoder.cpp:329-373), different mapping.
// Measured on MI355X (profiles/, tools/prof_phases.py): only ~9 packets fit a CU (16 KiB hash table each), a
// wave issues one instruction per ~4 cycles, and per 64-position group the serial chain
// probe -> walk -> repair spends most of its time in work that is parallel over positions. So a packet gets a
// 256-thread workgroup and speculates over a *supergroup* of 256 consecutive positions:
//
//   P1  all four waves: hash, probe + speculative insert (three workgroup barriers separate "everyone read",
//       "everyone wrote", "losers marked"), candidate bytes, lengths against the table candidate and against
//       the nearest earlier same-hash position of the supergroup (found through a 256-entry hash list in LDS);
//   P2  the walk, the only serial part: wave 0 replays the reference's decisions over its 64 positions (the
//       hand-written scalar loop of zz_level1.h), publishes where it ended and which lanes it visited, then
//       wave 1, 2, 3; a match may run over into the next wave's positions;
//   P3  all four waves: table repair, tokens -> fixed-Huffman fragments, one prefix sum across the workgroup,
//       ds_or into a shared bit ring, coalesced flush.
#pragma once
#include "zz_level1.h"

namespace zz {

#define ZZ_X4_WAVES 4
#define ZZ_X4_THREADS 256
#define ZZ_X4_RING 512            // words; one supergroup adds at most 256*31 bits = 248 words

// per-lane walk info for a supergroup (candidate lane index needs 8 bits)
#define ZZ_XI_LENA(i) ((i) & 15u)
#define ZZ_XI_LENB(i) (((i) >> 4) & 15u)
#define ZZ_XI_Q(i) (((i) >> 8) & 255u)
#define ZZ_XI_DUP 0x10000u
#define ZZ_XI_HARD 0x20000u
#define ZZ_XI_EXTA 0x40000u
#define ZZ_XI_EXTB 0x80000u

struct x4_shared {
    uint16_t T[ZZ_HASH_SIZE];          // hashtable (encoder.h:76) as pos+1; 0 = empty; 0xFFFF = "several lanes" marker
    uint32_t ring[ZZ_X4_RING];
    uint16_t Hs[ZZ_X4_THREADS];        // this supergroup's hashes (0xFFFF for inactive lanes)
    uint64_t w_mst[ZZ_X4_WAVES];       // per wave: match-start lanes
    uint64_t w_cov[ZZ_X4_WAVES];       //           lanes inside matches
    uint32_t w_end[ZZ_X4_WAVES];       //           first undecided position after the wave's walk (supergroup-relative)
    uint32_t w_bits[ZZ_X4_WAVES];      //           bits this wave appends
    uint64_t ad_a[ZZ_X4_WAVES], ad_c[ZZ_X4_WAVES];
};

// LDS loads land in VGPRs even when every lane reads the same word; these make the uniformity explicit so that the
// values can live in SGPRs (the scalar walk needs them there)
__device__ __forceinline__ uint32_t x4_u32(const uint32_t& v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t x4_u64(const uint64_t& v)
{
    const uint64_t t = v;
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(t >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
}

// lanes of wave `wv` (64 of them) that the walk visited: in front of the wave's end, not inside a match, or a match start
__device__ __forceinline__ uint64_t x4_visited(const x4_shared& S, int wv, uint32_t nact)
{
    const uint32_t lo = 64u * (uint32_t)wv;
    const uint32_t act = nact > lo ? (nact - lo < 64 ? nact - lo : 64) : 0;
    const uint64_t actmask = act == 64 ? ~0ull : ((1ull << act) - 1);
    const uint32_t end = x4_u32(S.w_end[wv]);
    const uint32_t rel = end > lo ? end - lo : 0;
    const uint64_t front = rel >= 64 ? ~0ull : ((1ull << rel) - 1);
    return (actmask & front & ~x4_u64(S.w_cov[wv])) | x4_u64(S.w_mst[wv]);
}

template <bool SAFE>
__device__ __forceinline__ void x4_encode_body(const zz_packet_params& P, x4_shared& S, uint32_t* out32, uint32_t& bitpos, uint32_t& flushed,
                                               const uint8_t* src, const uint8_t* end, uint32_t n)
{
    const uint32_t tid = threadIdx.x;
    const int lane = (int)(tid & 63);
    const int wv = __builtin_amdgcn_readfirstlane((int)(tid >> 6));   // wave-uniform, and the compiler knows it
    uint32_t cur = 0;
    ZZ_PROF_DECL
    uint64_t w = tid < n ? ld64<SAFE>(src + tid, end) : 0;
    while (cur < n) {
        ZZ_T(0); ZZ_C(10, 1);
        const uint32_t nact = (n - cur) < ZZ_X4_THREADS ? (n - cur) : ZZ_X4_THREADS;
        const uint32_t p = cur + tid;
        const bool active = tid < nact;
        const uint32_t wlo = 64u * (uint32_t)wv;
        const uint32_t wact = nact > wlo ? (nact - wlo < 64 ? nact - wlo : 64) : 0;   // active lanes of this wave
        const uint64_t actmask = wact == 64 ? ~0ull : ((1ull << wact) - 1);

        // ---- P1: probe ---------------------------------------------------------------------------------------
        const uint32_t h = calc_hash3((uint32_t)(w >> 8));              // bytes p+1..p+3 (encoder.cpp:344)
        uint32_t old = 0;
        if (active) old = S.T[h];                                       // encoder.cpp:345
        S.Hs[tid] = active ? (uint16_t)h : (uint16_t)0xFFFF;
        uint64_t wc = 0;
        if (active && old) wc = ld64<SAFE>(src + (old - 1), end);       // encoder.cpp:350
        __syncthreads();                                                // every lane has read the old table
        if (active) S.T[h] = (uint16_t)(p + 1);                         // encoder.cpp:346, speculative
        __syncthreads();
        const bool lost = active && S.T[h] != (uint16_t)(p + 1);
        if (lost) S.T[h] = 0xFFFF;                                      // tell the lane whose write survived
        __syncthreads();
        const bool multi = active && S.T[h] == 0xFFFF;                  // my hash occurs more than once

        ZZ_T(1);
        // which positions of the supergroup share my hash?
        uint64_t ms0 = 0, ms1 = 0, ms2 = 0, ms3 = 0;                    // member lanes per wave (multi lanes only)
        uint32_t info = 0;
        uint64_t mm = ballot(multi);
        while (mm) {
            const int l0 = __builtin_ctzll(mm);
            const uint32_t hv = readlane(h, l0);
            const uint64_t m0 = ballot(S.Hs[lane] == hv), m1 = ballot(S.Hs[64 + lane] == hv);
            const uint64_t m2 = ballot(S.Hs[128 + lane] == hv), m3 = ballot(S.Hs[192 + lane] == hv);
            const uint64_t own = wv == 0 ? m0 : wv == 1 ? m1 : wv == 2 ? m2 : m3;
            if ((own >> lane) & 1) {
                ms0 = m0; ms1 = m1; ms2 = m2; ms3 = m3;
                const uint32_t cnt = (uint32_t)(__builtin_popcountll(m0) + __builtin_popcountll(m1) +
                                                __builtin_popcountll(m2) + __builtin_popcountll(m3));
                // nearest earlier member
                const uint64_t below = own & ((1ull << lane) - 1);
                int q = -1;
                if (below) q = (int)wlo + (63 - __builtin_clzll(below));
                else if (wv > 2 && m2) q = 128 + (63 - __builtin_clzll(m2));
                else if (wv > 1 && m1) q = 64 + (63 - __builtin_clzll(m1));
                else if (wv > 0 && m0) q = 63 - __builtin_clzll(m0);
                if (q >= 0) {
                    info = ZZ_XI_DUP | ((uint32_t)q << 8);
                    if (cnt > 2) info |= ZZ_XI_HARD;
                }
            }
            mm &= ~own;
        }

        ZZ_T(2);
        // lengths against both possible candidates, capped at 8 ("8 or more")
        const uint32_t left = active ? n - p : 0;                       // bytes left in the block (D1 clamp)
        const uint32_t cap8 = left < 8 ? left : 8;
        uint64_t x = ~0ull;
        uint32_t la = 0;
        if (active && old) {
            x = w ^ wc;
            la = x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8;
            if (la > cap8) la = cap8;
        }
        info |= la;
        if (la == 8 && left > 8) info |= ZZ_XI_EXTA;
        if (info & ZZ_XI_DUP) {
            const uint64_t xq = w ^ ld64<SAFE>(src + cur + ZZ_XI_Q(info), end);
            uint32_t lb = xq ? (uint32_t)__builtin_ctzll(xq) >> 3 : 8;
            if (lb > cap8) lb = cap8;
            info |= lb << 4;
            if (lb == 8 && left > 8) info |= ZZ_XI_EXTB;
        }
        const uint64_t E = ballot(active && ((info & ZZ_XI_HARD) || ZZ_XI_LENA(info) >= 4 || ZZ_XI_LENB(info) >= 4));
        const uint64_t Ms = ballot(active && la >= 4 && !(info & (ZZ_XI_DUP | ZZ_XI_EXTA)));

        ZZ_T(3);
        // ---- P2: the walk, wave after wave (encoder.cpp:341-368) -----------------------------------------------
        uint64_t mst = 0, cov = 0, usedB = 0;
        uint32_t ovlen = 0, ovcand1 = 0;
        for (int turn = 0; turn < ZZ_X4_WAVES; ++turn) {
            if (wv == turn) {
                ZZ_T(4);
                const uint32_t prev_end = turn ? x4_u32(S.w_end[turn - 1]) : 0;
                uint32_t pos = prev_end > wlo ? prev_end - wlo : 0;     // a match of the previous wave may reach in here
                if (wact == 0) pos = 0;
                if (pos) cov = pos >= 64 ? ~0ull : ((1ull << pos) - 1);   // lanes the previous wave's match covers
                while (pos < wact) {
                    l1_fast_walk(E, Ms, la, pos, mst, cov);
                    if (pos >= wact) break;
                    const uint64_t Er = E & (~0ull << pos);
                    if (!Er) { pos = wact; break; }
                    const int e = __builtin_ctzll(Er);
                    const uint64_t probed = ~cov | mst;                 // own lanes below e the parse has visited
                    const uint32_t inf = readlane(info, e);
                    const uint32_t pe = cur + wlo + (uint32_t)e;
                    const uint32_t maxlen = (n - pe) < ZZ_MAX_LEN ? (n - pe) : ZZ_MAX_LEN;
                    uint32_t mlen;
                    if (!(inf & ZZ_XI_HARD)) {
                        bool useB = false;
                        if (inf & ZZ_XI_DUP) {
                            const uint32_t q = ZZ_XI_Q(inf);
                            const int qw = (int)(q >> 6);
                            const uint64_t vis = qw == turn ? probed : x4_visited(S, qw, nact);
                            useB = (vis >> (q & 63)) & 1;
                        }
                        mlen = useB ? ZZ_XI_LENB(inf) : ZZ_XI_LENA(inf);
                        if (mlen >= 4) {
                            if (inf & (useB ? ZZ_XI_EXTB : ZZ_XI_EXTA)) {     // remain(), encoder.cpp:64-90
                                const uint32_t cand = useB ? cur + ZZ_XI_Q(inf) : readlane(old, e) - 1;
                                mlen = wave_extend_match<SAFE>(src, pe, cand, maxlen, end);
                                if (lane == e) ovlen = mlen;
                            }
                            if (useB) usedB |= 1ull << e;
                        }
                    } else {
                        // hash shared by 3+ positions: candidate = most recent visited member, else the table's
                        uint32_t cand1 = 0;
                        for (int qw = turn; qw >= 0 && !cand1; --qw) {
                            const uint64_t members = readlane64(qw == 0 ? ms0 : qw == 1 ? ms1 : qw == 2 ? ms2 : ms3, e);
                            uint64_t vis = qw == turn ? (probed & ((1ull << e) - 1)) : x4_visited(S, qw, nact);
                            const uint64_t Sx = members & vis;
                            if (Sx) cand1 = cur + 64u * (uint32_t)qw + (uint32_t)(63 - __builtin_clzll(Sx)) + 1;
                        }
                        uint64_t xe = ~0ull;
                        if (cand1) {
                            const uint64_t cbytes = ld64<SAFE>(src + (cand1 - 1), end);   // same address in every lane
                            xe = readlane64(w, e) ^ x4_u64(cbytes);
                        }
                        else {
                            cand1 = readlane(old, e);
                            if (cand1) xe = readlane64(x, e);
                        }
                        mlen = 0;
                        if ((uint32_t)xe == 0 && maxlen >= 4) {
                            if (xe != 0) mlen = (uint32_t)__builtin_ctzll(xe) >> 3;
                            else mlen = wave_extend_match<SAFE>(src, pe, cand1 - 1, maxlen, end);
                            if (mlen > maxlen) mlen = maxlen;
                        }
                        if (lane == e) { ovlen = mlen; ovcand1 = cand1; }
                    }
                    if (mlen > 3) {                                      // encoder.cpp:356
                        mst |= 1ull << e;
                        cov |= (mlen >= 64u - (uint32_t)e) ? (~0ull << e) : (((1ull << mlen) - 1) << e);
                        pos = (uint32_t)e + mlen;                        // encoder.cpp:361-362
                    } else {
                        pos = (uint32_t)e + 1;                           // a literal after all (encoder.cpp:367)
                    }
                }
                ZZ_T(5);
                if (lane == 0) {
                    S.w_mst[wv] = mst;
                    S.w_cov[wv] = cov;
                    const uint32_t my_end = wlo + pos;
                    S.w_end[wv] = my_end > prev_end ? my_end : prev_end;
                }
            }
            __syncthreads();
        }
        ZZ_T(6);
        const uint32_t sg_end = x4_u32(S.w_end[ZZ_X4_WAVES - 1]);               // >= nact
        const uint32_t next = cur + sg_end;
        const uint64_t wnext = next + tid < n ? ld64<SAFE>(src + next + tid, end) : 0;

        // ---- P3: table repair ----------------------------------------------------------------------------------
        const uint64_t visited = x4_visited(S, wv, nact);
        const bool is_committed = (visited >> lane) & 1;
        if (active && !is_committed) S.T[h] = (uint16_t)old;            // skipped lanes restore (also clears the marker)
        __syncthreads();
        if (multi && is_committed) {
            // among visited positions sharing a hash the highest one stays in the table
            bool later = ((wv == 0 ? ms0 : wv == 1 ? ms1 : wv == 2 ? ms2 : ms3) & visited & ((~0ull << lane) << 1)) != 0;
            if (wv < 1) later |= (ms1 & x4_visited(S, 1, nact)) != 0;
            if (wv < 2) later |= (ms2 & x4_visited(S, 2, nact)) != 0;
            if (wv < 3) later |= (ms3 & x4_visited(S, 3, nact)) != 0;
            if (!later) S.T[h] = (uint16_t)(p + 1);
        }

        ZZ_T(7);
        // ---- P3: tokens -> fragments -> shared ring ---------------------------------------------------------------
        uint32_t bits = 0, nb = 0;
        if (is_committed) {
            if (!((mst >> lane) & 1)) {
                fixed_code((uint32_t)(w & 0xFF), bits, nb);                          // codes_f[*sourcePtr], encoder.cpp:367
            } else {
                const bool b = (usedB >> lane) & 1;
                const uint32_t tlen = ovlen ? ovlen : (b ? ZZ_XI_LENB(info) : ZZ_XI_LENA(info));
                const uint32_t cand1 = (info & ZZ_XI_HARD) ? ovcand1 : (b ? cur + ZZ_XI_Q(info) + 1 : old);
                const uint32_t lc = fixed_lcode_packed(tlen);                        // lcodes_f[matchLength], encoder.cpp:358
                const uint32_t ll = lc >> 16;
                uint32_t bucket, eb, ev;
                dist_symbol(p + 1 - cand1, bucket, eb, ev);                          // WriteDistance, encoder.cpp:135-141
                bits = (lc & 0xFFFF) | (bitrev(bucket, 5) << ll) | (ev << (ll + 5));
                nb = ll + 5 + eb;
            }
        }
        const uint32_t incl = wave_scan_incl(nb);
        if (lane == 63) S.w_bits[wv] = incl;
        __syncthreads();                                                // also orders the repair before the next probe
        uint32_t base = bitpos, total = 0;
#pragma unroll
        for (int i = 0; i < ZZ_X4_WAVES; ++i) {
            const uint32_t b = x4_u32(S.w_bits[i]);
            if (i < wv) base += b;
            total += b;
        }
        if (nb) {
            const uint32_t o = base + incl - nb, sh = o & 31, wd = o >> 5;
            atomicOr(&S.ring[wd & (ZZ_X4_RING - 1)], bits << sh);
            if (sh + nb > 32) atomicOr(&S.ring[(wd + 1) & (ZZ_X4_RING - 1)], bits >> (32 - sh));
        }
        bitpos += total;
        __syncthreads();
        {
            const uint32_t full = bitpos >> 5;
            const uint32_t wd = flushed + tid;                          // at most 249 words are pending
            if (wd < full) {
                out32[wd] = S.ring[wd & (ZZ_X4_RING - 1)];
                S.ring[wd & (ZZ_X4_RING - 1)] = 0;
            }
            flushed = full;
        }
        cur = next;
        w = wnext;
        ZZ_T(8);
    }
#ifdef ZZ_PROF
    if ((threadIdx.x & 63) == 0 && P.prof) for (int _i = 0; _i < 16; ++_i) atomicAdd(&P.prof[_i], prof_acc[_i]);
#endif
}

// append a few uniform bits (thread 0 writes; every thread tracks bitpos)
__device__ __forceinline__ void x4_put(x4_shared& S, uint32_t& bitpos, uint32_t bits, uint32_t nb)
{
    if (threadIdx.x == 0 && nb) {
        const uint32_t sh = bitpos & 31, wd = bitpos >> 5;
        atomicOr(&S.ring[wd & (ZZ_X4_RING - 1)], bits << sh);
        if (sh + nb > 32) atomicOr(&S.ring[(wd + 1) & (ZZ_X4_RING - 1)], bits >> (32 - sh));
    }
    bitpos += nb;
}

__global__ __launch_bounds__(ZZ_X4_THREADS) void k_encode_l1x4(zz_packet_params P)
{
    __shared__ x4_shared S;
    const uint32_t tid = threadIdx.x;
    const int lane = (int)(tid & 63), wv = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t k = blockIdx.x;
    const uint64_t off = (uint64_t)k * P.packet_size;
    const uint32_t len = (uint32_t)((P.n - off) < P.packet_size ? (P.n - off) : P.packet_size);
    const bool is_final = P.last_is_final && k == P.npk - 1;
    const uint32_t n = is_final ? len : len - 1;   // bytes of the compressing AddData (zzflate.cpp:113,116)
    const uint8_t* src = P.src + off;
    const uint8_t* end = P.src + P.n;
    uint32_t* out32 = (uint32_t*)(P.slots + (uint64_t)k * P.slot_stride);

    {   // cold table (encoder.cpp:533-536), empty ring
        uint4* t4 = (uint4*)S.T;
        for (uint32_t i = tid; i < sizeof(S.T) / 16; i += ZZ_X4_THREADS) t4[i] = make_uint4(0, 0, 0, 0);
        for (uint32_t i = tid; i < ZZ_X4_RING; i += ZZ_X4_THREADS) S.ring[i] = 0;
    }
    if (P.cks_kind == ZZ_CKS_ADLER) {
        // Adler-32 partial of the packet, a quarter per wave (sums are position-weighted, so they simply add)
        uint32_t A = 0;
        uint64_t C = 0;
        const uint32_t nch = len >> 4;
        for (uint32_t c = tid; c < nch; c += ZZ_X4_THREADS) {
            uint4 v;
            __builtin_memcpy(&v, src + ((uint64_t)c << 4), 16);
            const uint32_t s0 = __builtin_amdgcn_sad_u8(v.x, 0u, 0u), s1 = __builtin_amdgcn_sad_u8(v.y, 0u, 0u);
            const uint32_t s2 = __builtin_amdgcn_sad_u8(v.z, 0u, 0u), s3 = __builtin_amdgcn_sad_u8(v.w, 0u, 0u);
            auto w3 = [](uint32_t x) { return ((x >> 8) & 0xFF) + 2 * ((x >> 16) & 0xFF) + 3 * (x >> 24); };
            const uint32_t t = w3(v.x) + (w3(v.y) + 4 * s1) + (w3(v.z) + 8 * s2) + (w3(v.w) + 12 * s3);
            const uint32_t s = s0 + s1 + s2 + s3;
            A += s;
            C += (uint64_t)(c << 4) * s + t;
        }
        const uint32_t i = (nch << 4) + tid;
        if (i < len) { const uint32_t d = src[i]; A += d; C += (uint64_t)i * d; }
        const uint64_t At = wave_sum64(A), Ct = wave_sum64(C);
        if (lane == 0) { S.ad_a[wv] = At; S.ad_c[wv] = Ct; }
    }
    __syncthreads();
    if (P.cks_kind == ZZ_CKS_ADLER && tid == 0) {
        const uint64_t At = S.ad_a[0] + S.ad_a[1] + S.ad_a[2] + S.ad_a[3];
        const uint64_t Ct = S.ad_c[0] + S.ad_c[1] + S.ad_c[2] + S.ad_c[3];
        zz_cks c;
        c.a = (uint32_t)(At % ZZ_ADLER_MOD);
        c.b = (uint32_t)(((uint64_t)len * At - Ct) % ZZ_ADLER_MOD);
        P.cks[k] = c;
    }

    uint32_t bitpos = 0, flushed = 0;
    if (n > 0) {
        x4_put(S, bitpos, (is_final ? 1u : 0u) | (1u << 1), 3);          // StartBlock(FixedHuffman, final), encoder.cpp:338
        __syncthreads();
        if (k + 2 >= P.npk) x4_encode_body<true>(P, S, out32, bitpos, flushed, src, end, n);
        else x4_encode_body<false>(P, S, out32, bitpos, flushed, src, end, n);
        __syncthreads();
        x4_put(S, bitpos, 0, 7);                                          // codes_f[256] (encoder.cpp:371)
    }
    if (!is_final) {
        // one stored byte = byte alignment (zzflate.cpp:118-120, encoder.cpp:482-502)
        x4_put(S, bitpos, 0, 3);
        bitpos = (bitpos + 7) & ~7u;
        x4_put(S, bitpos, 0xFFFE0001u, 32);
        x4_put(S, bitpos, src[len - 1], 8);
    } else if (n == 0) {
        x4_put(S, bitpos, 1u | (1u << 1), 3);                             // empty input: one empty fixed block (D8)
        x4_put(S, bitpos, 0, 7);
    }
    bitpos = (bitpos + 7) & ~7u;                                          // Flush, outputbitstream.h:105-124
    const uint32_t bytes = bitpos >> 3;
    const uint32_t words = (bytes + 3) >> 2;
    __syncthreads();
    for (uint32_t wd = flushed + tid; wd < words; wd += ZZ_X4_THREADS) out32[wd] = S.ring[wd & (ZZ_X4_RING - 1)];
    if (tid == 0) {
        P.sizes[k] = bytes;
        if (bytes > P.slot_stride) atomicOr(P.err, 1u);
    }
}

}  // namespace zz
