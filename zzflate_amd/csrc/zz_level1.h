// zz_level1.h -- level 1: greedy single-probe LZ77 + fixed Huffman, one packet per wavefront.
//
// Restates WriteBlockFixedHuff (encoder.cpp:329-373) for a packet (zzflate.cpp:101-125) and is bit-exact
// with it. The reference walks the packet one position at a time: hash bytes i+1..i+3 (CalcHash,
// encoder.cpp:11-17), take the single candidate stored under that hash, overwrite it with i, compare from
// i, emit a match if >= 4 bytes agree, else a literal; positions inside a match are skipped and never
// inserted. That is a serial dependency chain, so the wave speculates over a *group* of 64 consecutive
// positions at once:
//
//   1. every lane loads 8 bytes at its position, hashes, reads the pre-group candidate from the LDS table
//      and speculatively writes its own position (one LDS round trip); reading the slot back tells which
//      lanes share a hash inside the group ("dup" lanes: their true candidate depends on the parse);
//   2. every lane XORs its 8 bytes with the 8 bytes at the pre-group candidate (one memory round trip);
//   3. a scalar walk over ballot masks replays the reference's decisions in order: the next event is the
//      first lane that either has a >= 4 byte match (non-dup) or is a dup lane, whose candidate is
//      resolved from the lanes committed so far (bytes are already in registers); matches longer than 8
//      are extended by all 64 lanes at once (4 bytes per lane);
//   4. the table is repaired: lanes the parse skipped restore the old entry, and among committed lanes
//      sharing a hash the highest position wins -- exactly the state the serial loop leaves;
//   5. tokens are turned into fixed-Huffman fragments and appended through the bit ring (zz_emit.h).
#pragma once
#include "zz_checksum.h"
#include "zz_emit.h"

namespace zz {

// compare src[pe+8 ..) with src[cand+8 ..) with the whole wave, 4 bytes per lane; returns the match
// length (>= 8) clamped to maxlen. Only called when the first 8 bytes are known equal.
__device__ __forceinline__ uint32_t wave_extend_match(const uint8_t* src, uint32_t pe, uint32_t cand,
                                                      uint32_t maxlen, const uint8_t* end)
{
    const uint32_t o = 8 + 4 * (uint32_t)lane_id();
    uint32_t d = 0;
    const bool act = o < maxlen;
    if (act) d = load32_safe(src + pe + o, end) ^ load32_safe(src + cand + o, end);
    const uint64_t neq = ballot(act && d != 0);
    if (!neq) return maxlen;
    const int k = __builtin_ctzll(neq);
    const uint32_t dk = readlane(d, k);
    const uint32_t len = 8 + 4 * (uint32_t)k + ((uint32_t)__builtin_ctz(dk) >> 3);
    return len < maxlen ? len : maxlen;
}

__device__ __forceinline__ uint32_t calc_hash3(uint32_t three_bytes)   // encoder.cpp:11-17
{
    return ((three_bytes & 0xFFFFFFu) * 0x00d68664u) >> (32 - ZZ_HASH_BITS);
}

__global__ __launch_bounds__(ZZ_WAVE) void k_encode_l1(zz_packet_params P)
{
    __shared__ uint16_t T[ZZ_HASH_SIZE];          // hashtable (encoder.h:76) as pos+1, 0 = empty
    __shared__ uint32_t ring_words[ZZ_RING_WORDS];
    __shared__ uint32_t lcodes[ZZ_MAX_LEN + 1];   // lcodes_f (fixedhuffmanluts.cpp:8-46), packed

    const int lane = lane_id();
    const uint32_t k = blockIdx.x;
    const uint64_t off = (uint64_t)k * P.packet_size;
    const uint32_t len = (uint32_t)((P.n - off) < P.packet_size ? (P.n - off) : P.packet_size);
    const bool is_final = P.last_is_final && k == P.npk - 1;
    const uint32_t n = is_final ? len : len - 1;   // bytes of the compressing AddData (zzflate.cpp:113,116)
    const uint8_t* src = P.src + off;
    const uint8_t* end = P.src + P.n;              // one past the last readable byte
    uint8_t* out = P.slots + (uint64_t)k * P.slot_stride;

    // cold table (encoder.cpp:533-536)
    {
        uint4* t4 = (uint4*)T;
        for (int i = lane; i < (int)(sizeof(T) / 16); i += ZZ_WAVE) t4[i] = make_uint4(0, 0, 0, 0);
        for (int l = lane; l <= ZZ_MAX_LEN; l += ZZ_WAVE) lcodes[l] = l >= 3 ? fixed_lcode_packed(l) : 0;
    }
    bitring ring;
    ring_init(ring, ring_words, out);   // includes the barrier that publishes T and lcodes

    if (P.cks_kind == ZZ_CKS_ADLER) {
        zz_cks c = wave_adler(src, len);
        if (lane == 0) P.cks[k] = c;
    }

    if (n > 0) {
        // StartBlock(FixedHuffman, final): encoder.cpp:143-147,338
        ring_append_uniform(ring, (is_final ? 1u : 0u) | (1u << 1), 3);

        uint32_t cur = 0;
        while (cur < n) {
            const uint32_t nact = (n - cur) < ZZ_WAVE ? (n - cur) : ZZ_WAVE;
            const uint64_t actmask = nact == 64 ? ~0ull : ((1ull << nact) - 1);
            const uint32_t p = cur + lane;
            const bool active = lane < (int)nact;

            // (1) load, hash, probe + speculative insert
            const uint64_t w = active ? load64_safe(src + p, end) : 0;
            const uint32_t h = calc_hash3((uint32_t)(w >> 8));          // bytes p+1..p+3 (encoder.cpp:344)
            uint32_t old = 0;
            if (active) {
                old = T[h];                                             // encoder.cpp:345
                T[h] = (uint16_t)(p + 1);                               // encoder.cpp:346
            }
            __syncthreads();
            const bool lost = active && T[h] != (uint16_t)(p + 1);
            uint64_t lostmask = ballot(lost);
            uint64_t dupmask = 0;      // lanes with an earlier same-hash lane in this group
            uint64_t multimask = 0;    // lanes whose hash occurs more than once in this group
            uint64_t myset = 0;        // per lane: all lanes of the group sharing my hash (0 if unique)
            while (lostmask) {
                const int l0 = __builtin_ctzll(lostmask);
                const uint32_t hv = readlane(h, l0);
                const uint64_t set = ballot(active && h == hv);
                if (active && h == hv) myset = set;
                dupmask |= set & (set - 1);        // all but the lowest lane of the set
                multimask |= set;
                lostmask &= ~set;
            }

            // (2) compare with the pre-group candidate
            uint64_t x = ~0ull;
            if (active && old) x = w ^ load64_safe(src + (old - 1), end);   // encoder.cpp:350
            // a match needs 4 equal bytes AND 4 bytes left in the block (D1 clamp)
            const bool m4 = active && old && (uint32_t)x == 0 && p + 4 <= n;
            const uint64_t M = ballot(m4) & ~dupmask;
            uint64_t E = (M | dupmask) & actmask;

            // (3) the walk
            uint64_t committed = 0;     // probed lanes (literals and match starts)
            uint32_t tlen = 0, tdist = 0;
            uint32_t pos = 0;
            for (;;) {
                const uint64_t Er = E & ~((1ull << pos) - 1);
                if (!Er) { committed |= actmask & ~((1ull << pos) - 1); pos = nact; break; }
                const int e = __builtin_ctzll(Er);
                committed |= ((1ull << e) - 1) & ~((1ull << pos) - 1);   // literals pos..e-1 (encoder.cpp:367)
                committed |= 1ull << e;
                const uint32_t pe = cur + (uint32_t)e;
                uint32_t cand = 0;
                uint64_t xe = ~0ull;
                if ((M >> e) & 1) {
                    cand = readlane(old, e) - 1;
                    xe = readlane64(x, e);
                } else {
                    // dup lane: candidate = most recent committed lane with my hash, else the table's
                    const uint64_t S = readlane64(myset, e) & committed & ((1ull << e) - 1);
                    if (S) {
                        const int c = 63 - __builtin_clzll(S);
                        cand = cur + (uint32_t)c;
                        xe = readlane64(w, e) ^ readlane64(w, c);
                    } else {
                        const uint32_t o = readlane(old, e);
                        if (o) { cand = o - 1; xe = readlane64(x, e); }
                    }
                }
                uint32_t mlen = 0;
                const uint32_t maxlen = (n - pe) < ZZ_MAX_LEN ? (n - pe) : ZZ_MAX_LEN;
                if ((uint32_t)xe == 0 && maxlen >= 4) {
                    if (xe != 0) mlen = (uint32_t)__builtin_ctzll(xe) >> 3;          // ZeroCount, gcc.h:10-13
                    else mlen = wave_extend_match(src, pe, cand, maxlen, end);       // remain(), encoder.cpp:64-90
                    if (mlen > maxlen) mlen = maxlen;
                }
                if (mlen > 3) {                                                      // encoder.cpp:356
                    if (lane == e) { tlen = mlen; tdist = pe - cand; }
                    pos = (uint32_t)e + mlen;                                        // encoder.cpp:361-362
                } else {
                    pos = (uint32_t)e + 1;
                    E &= ~(1ull << e);
                }
                if (pos >= nact) break;
            }

            // (4) table repair
            const bool is_committed = (committed >> lane) & 1;
            if (active && !is_committed) T[h] = (uint16_t)old;
            if (multimask) {
                __syncthreads();
                const bool winner = is_committed && myset && (((myset & committed) >> lane) >> 1) == 0;
                if (winner) T[h] = (uint16_t)(p + 1);
            }

            // (5) fixed-Huffman fragments
            uint32_t bits = 0, nb = 0;
            if (is_committed) {
                if (tlen == 0) {
                    fixed_code((uint32_t)(w & 0xFF), bits, nb);                      // codes_f[*sourcePtr]
                } else {
                    const uint32_t lc = lcodes[tlen];                                // lcodes_f[matchLength]
                    const uint32_t ll = lc >> 16;
                    uint32_t bucket, eb, ev;
                    dist_symbol(tdist, bucket, eb, ev);                              // WriteDistance, encoder.cpp:135-141
                    bits = (lc & 0xFFFF) | (bitrev(bucket, 5) << ll) | (ev << (ll + 5));
                    nb = ll + 5 + eb;
                }
            }
            ring_append(ring, bits, nb);   // barriers inside also order (4) before the next group's probe
            cur += pos;
        }
        // EOB: codes_f[256] = 7 zero bits (encoder.cpp:371)
        ring_append_uniform(ring, 0, 7);
    }
    if (!is_final) {
        // SetLevel(0); AddData(e-1, e): one stored byte = byte alignment (zzflate.cpp:118-120,
        // encoder.cpp:482-502): BFINAL=0 BTYPE=00, pad, LEN=1, NLEN=0xFFFE, the byte
        ring_append_uniform(ring, 0, 3);
        ring_pad_to_byte(ring);
        ring_append_uniform(ring, 0xFFFE0001u, 32);
        ring_append_uniform(ring, src[len - 1], 8);
    } else if (n == 0) {
        // empty final packet (only for empty input): one empty fixed block (D8 divergence, documented)
        ring_append_uniform(ring, 1u | (1u << 1), 3);
        ring_append_uniform(ring, 0, 7);
    }
    const uint32_t bytes = ring_finish(ring);
    if (lane == 0) {
        P.sizes[k] = bytes;
        if (bytes > P.slot_stride) atomicOr(P.err, 1u);
    }
}

}  // namespace zz
