// zz_stream2.h -- the reference's sequential whole-buffer stream at levels 2,3 (threaded=false: zzflate.cpp:84-95,
// Encoder::AddData -> WriteDeflateBlock -> WriteBlock2Pass, encoder.cpp:217-303,506-552) on ONE wavefront.
//
// A compatibility mode (bit-identical to the reference, pinned to its whole-file goldens), not a throughput mode:
// the stream is one dependency chain -- the hash table is carried from block to block (FixHashTable, :320-327;
// here entries are absolute positions, so nothing is rebased), blocks end after 20,000 records or 500,000 bytes
// (:44 maxRecords, :518-522), each block is token pass -> histograms -> Huffman -> header + body or stored
// fallback. The pieces are those of zz_level2.h with 32-bit positions; the per-position status bitmaps and the
// token list live in global scratch because a block can span 500,000 positions.
//
// Token pass differences from the packet kernel: batches of 16,384 start wherever the previous one ended
// (:225-234), so positions are inserted in 64-wide chunks that simply continue where the last one stopped;
// positions past the batch end that a match covers are inserted by an insert-only sweep; and when fewer than 64
// records remain before the 20,000 cut the chunk width drops to one position, so that nothing beyond the cut is
// ever inserted.
#pragma once
#include "zz_level2.h"

namespace zz {

struct st_token { uint32_t start; uint16_t dist, len; };
#define ZZ_ST_MAX_RECORDS 20000u                 // encoder.h:44
#define ZZ_ST_MAX_BLOCK 500000u                  // encoder.cpp:518-522
#define ZZ_ST_WORDS ((ZZ_ST_MAX_BLOCK + 63) / 64 + 1)
// global scratch: tokens | cov bitmap | mst bitmap | mcount
#define ZZ_ST_SCRATCH_BYTES (ZZ_ST_MAX_RECORDS * 8 + ZZ_ST_WORDS * 8 * 2 + ZZ_ST_WORDS * 4 + 64)

struct st_state {
    uint32_t* T;            // LDS: absolute position + 1 (0 = empty)
    const uint8_t* base;    // whole input
    const uint8_t* end;     // one past the last readable byte
    uint64_t bs;            // absolute position of the block start
    st_token* tokens;
    unsigned long long* cov;
    unsigned long long* mst;
};

// insert positions [lo, hi) (block-relative) of one 64-wide chunk starting at `cb`; returns each lane's candidate
// as absolute position + 1 (0 = none / too far: distance >= 32768 is skipped, encoder.cpp:392)
__device__ __forceinline__ uint32_t st_insert_chunk(st_state& S, uint32_t cb, uint32_t lo, uint32_t hi, uint32_t& hout)
{
    const int lane = lane_id();
    const uint32_t q = cb + lane;
    const bool ins = q >= lo && q < hi;
    const uint64_t qa = S.bs + q;                                          // absolute
    const uint32_t w4 = ins ? load32_safe(S.base + qa, S.end) : 0;
    const uint32_t h = calc_hash3(w4);                                    // CalcHash(source + j), :388
    hout = h;
    uint32_t old = 0;
    if (ins) { old = S.T[h]; S.T[h] = (uint32_t)qa + 1; }                 // :389-390 / :474-480
    ZZ_WAVE_SYNC();
    uint32_t rb = 0;
    if (ins) rb = S.T[h];
    uint64_t lostmask = ballot(ins && rb != (uint32_t)qa + 1);
    uint32_t cand1 = old;
    while (lostmask) {
        const int l0 = __builtin_ctzll(lostmask);
        const uint32_t hv = readlane(h, l0);
        const bool mine = ins && h == hv;
        const uint64_t set = ballot(mine);
        const uint64_t below = set & ((1ull << lane) - 1);
        if (mine && below) cand1 = (uint32_t)(S.bs + cb) + (63 - __builtin_clzll(below)) + 1;   // nearest earlier member
        ZZ_WAVE_SYNC();
        if (mine && (set >> lane) >> 1 == 0) S.T[h] = (uint32_t)qa + 1;                          // highest member wins
        lostmask &= ~set;
    }
    ZZ_WAVE_SYNC();
    if (!ins || cand1 == 0 || (uint32_t)qa + 1 - cand1 >= 0x8000u) cand1 = 0;                    // :392
    return cand1;
}

// publish a found token: list entry + status bits (all lanes call; `have` selects the lane that holds it)
__device__ __forceinline__ void st_publish(st_state& S, bool have, uint32_t idx, uint32_t ms, uint32_t dist, uint32_t mlen)
{
    if (have) {
        st_token t; t.start = ms; t.dist = (uint16_t)dist; t.len = (uint16_t)mlen;
        S.tokens[idx] = t;
        const uint32_t last = ms + mlen - 1;
        for (uint32_t wi = ms >> 6; wi <= (last >> 6); ++wi) {
            const uint32_t lo = wi == (ms >> 6) ? (ms & 63) : 0;
            const uint32_t hi = wi == (last >> 6) ? (last & 63) : 63;
            const unsigned long long mask = ((hi == 63 ? 0ull : (1ull << (hi + 1))) - 1) & ~((1ull << lo) - 1);
            atomicOr(&S.cov[wi], mask);
        }
        atomicOr(&S.mst[ms >> 6], 1ull << (ms & 63));
    }
}

// FirstPass over one block (encoder.cpp:217-248 loop + :375-440). Returns the block length; *ntok_out tokens.
__device__ __forceinline__ uint32_t st_token_pass(st_state& S, uint32_t byteCount, uint32_t* ntok_out)
{
    const int lane = lane_id();
    uint32_t ntok = 0, nrec = 0, length = 0;
    int64_t target = (int64_t)byteCount - ZZ_MAX_LEN;                     // :222
    if (target < 0) target = 0;
    while (target > 0 && nrec < ZZ_ST_MAX_RECORDS) {                       // :225
        const uint32_t bstart = length;
        const uint32_t bend = length + (uint32_t)(target < ZZ_BATCH_LEN ? target : ZZ_BATCH_LEN);
        uint32_t B = bstart + 1;                                          // backRefEnd (:380)
        uint32_t nextProbe = bstart + 1;                                  // j (:383); the batch's first byte is never probed
        uint32_t ins_end = bstart + 1;                                    // positions of this batch below ins_end are in the table
        bool cut = false;                                                 // the record array filled up (:426-430)
        bool matched = false;       // this batch has seen a match: B is the end of one, not the batch's initial bstart + 1
        while (!cut) {
            // near the record limit chunks are one position wide, so that nothing past the cut is ever inserted
#ifdef ZZ_ST_ALWAYS_CAREFUL
            const bool careful = true;
#else
            const bool careful = nrec + 64 >= ZZ_ST_MAX_RECORDS;
#endif
            if (careful && matched && B >= ins_end) {
                // AddHashEntries (:418): everything the last match covers, up to and including backRefEnd
                for (uint32_t sb = ins_end; sb <= B; sb += 64) {
                    uint32_t h2;
                    (void)st_insert_chunk(S, sb, sb, (B + 1 < sb + 64 ? B + 1 : sb + 64), h2);
                }
                ins_end = B + 1;
            }
            const uint32_t cb = ins_end;
            if (cb >= bend) break;
            const uint32_t hi = cb + (careful ? 1u : 64u) < bend ? cb + (careful ? 1u : 64u) : bend;
            uint32_t h;
            const uint32_t cand1 = st_insert_chunk(S, cb, cb, hi, h);
            ins_end = hi;
            const uint32_t q = cb + lane;
            if (hi > nextProbe) {
                const bool has = cand1 != 0 && q < hi;
                const uint64_t qa = S.bs + q;
                const uint64_t ca = (uint64_t)cand1 - 1;                  // absolute candidate position
                uint32_t fwd8 = 0, bwd8 = 0, room = 0;
                if (has) {
                    const uint64_t x = load64_safe(S.base + qa, S.end) ^ load64_safe(S.base + ca, S.end);   // :399
                    room = ca < ZZ_MAX_LEN ? (uint32_t)ca : ZZ_MAX_LEN;   // D4 + D11 caps
                    if (room >= 8) {
                        const uint64_t y = load64(S.base + qa - 8) ^ load64(S.base + ca - 8);
                        bwd8 = y ? (uint32_t)__builtin_clzll(y) >> 3 : 8;
                    } else {
                        while (bwd8 < room && S.base[qa - 1 - bwd8] == S.base[ca - 1 - bwd8]) bwd8++;
                    }
                    fwd8 = x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8;
                }
                const uint32_t broom = bwd8 < room ? bwd8 : room;
                for (;;) {
                    const uint32_t pend = q - B;                          // j - backRefEnd (:404)
                    const uint32_t bq = broom < pend ? broom : pend;
                    const uint64_t m = ballot(has && q >= nextProbe && fwd8 + bq >= 4);   // :406-407
                    if (!m) break;
                    const int e = __builtin_ctzll(m);
                    const uint32_t qe = cb + (uint32_t)e;
                    uint32_t fwd = readlane(fwd8, e), bw = readlane(bq, e);
                    const uint32_t ce1 = readlane(cand1, e);
                    const uint8_t* blk = S.base + S.bs;                   // block-relative addressing for the extenders
                    const int64_t crel = (int64_t)ce1 - 1 - (int64_t)S.bs;
                    if (fwd == 8) {
                        // forward extension, 4 bytes per lane (remain(), :64-90)
                        const uint32_t o = 8 + 4 * (uint32_t)lane;
                        uint32_t d = 0;
                        const bool act = o < ZZ_MAX_LEN;
                        if (act) d = load32_safe(blk + qe + o, S.end) ^ load32_safe(blk + crel + o, S.end);
                        const uint64_t neq = ballot(act && d != 0);
                        if (!neq) fwd = ZZ_MAX_LEN;
                        else {
                            const int k = __builtin_ctzll(neq);
                            fwd = 8 + 4 * (uint32_t)k + ((uint32_t)__builtin_ctz(readlane(d, k)) >> 3);
                            if (fwd > ZZ_MAX_LEN) fwd = ZZ_MAX_LEN;
                        }
                    }
                    if (bw == 8) {
                        const uint32_t re = readlane(room, e), pe = qe - B;
                        const uint32_t blim = re < pe ? re : pe;
                        if (blim > 8) bw = wave_extend_back(blk, (int64_t)qe, crel, blim);   // :92-102
                    }
                    uint32_t mlen = fwd + bw;
                    if (mlen > ZZ_MAX_LEN) mlen = ZZ_MAX_LEN;             // :412-415
                    const uint32_t ms = qe - bw;                          // :416
                    st_publish(S, lane == e, ntok, ms, (uint32_t)(S.bs + qe - (ce1 - 1)), mlen);   // :420
                    ntok++;
                    nrec++;
                    B = ms + mlen;                                        // :422
                    nextProbe = B + 1;                                    // :424
                    matched = true;
                    if (nrec == ZZ_ST_MAX_RECORDS) { cut = true; break; } // :426-430
                    if (nextProbe >= hi) break;
                }
            }
        }
        // what the last match covers beyond the inserted range (an overrun past the batch end, or the cut)
        if (matched && B >= ins_end) {
            for (uint32_t sb = ins_end; sb <= B; sb += 64) {
                uint32_t h2;
                (void)st_insert_chunk(S, sb, sb, (B + 1 < sb + 64 ? B + 1 : sb + 64), h2);
            }
        }
        uint32_t newEnd;
        if (cut || B > bend) newEnd = B;                                  // :435-436 (cut: end = 0)
        else { nrec++; newEnd = bend; }                                   // :438 closing literal record
        target -= (int64_t)newEnd - (int64_t)length;                      // :229
        length = newEnd;
    }
    if (target <= 0 && nrec < ZZ_ST_MAX_RECORDS) length = byteCount;      // :236-245 trailing literals
    *ntok_out = ntok;
    return length;
}

// stored block through the bit ring (keeps one output path): encoder.cpp:482-502
__device__ __forceinline__ void st_stored_block(bitring& ring, const uint8_t* p, uint32_t len, bool final)
{
    const int lane = lane_id();
    ring_append_uniform(ring, final ? 1u : 0u, 3);
    ring_pad_to_byte(ring);
    ring_append_uniform(ring, (len & 0xFFFF) | ((~len & 0xFFFF) << 16), 32);
    for (uint32_t i = 0; i < len; i += 256) {
        const uint32_t o = i + 4 * (uint32_t)lane;
        uint32_t v = 0, nb = 0;
        if (o < len) {
            const uint32_t k = len - o < 4 ? len - o : 4;
            for (uint32_t j = 0; j < k; ++j) v |= (uint32_t)p[o + j] << (8 * j);
            nb = 8 * k;
        }
        ring_append(ring, v, nb);
    }
}

struct zz_st_params {
    zz_packet_params pk;      // src, n, slots (one big output slot), sizes[0], err
    uint8_t* scratch;         // ZZ_ST_SCRATCH_BYTES (per range)
    zz_stream_ctl ctl;        // callback form: log of the EnsureOutputLength calls (the host cuts the chunks from it)
    // The reference's own threaded=true split (zzflate.cpp:67-78,97-125): range_step != 0 and one workgroup per range. Range r is
    // src[r * range_step, min((r + 1) * range_step, n)), encoded by a fresh encoder (cold table) into slot r; a range that is
    // not the last compresses all but its last byte with BFINAL clear and closes with that byte as a stored block
    // (SetLevel(0); AddData(e - 1, e)): byte-aligned, so the ranges' outputs concatenate. Positions stay relative to the
    // whole input: a candidate's room for backward extension counts the bytes in front of the range too, as the reference's
    // pointer arithmetic does (SURVEY.md App. B D4 -- the packet kernels' `before`).
    uint64_t range_step;
};

__global__ __launch_bounds__(ZZ_WAVE) void k_stream_l2(zz_st_params Q)
{
    const zz_packet_params& P = Q.pk;
    __shared__ uint32_t T[ZZ_HASH_SIZE];
    __shared__ __attribute__((aligned(16))) uint8_t hs[8192];          // Huffman scratch (layout of zz_level2.h)
    __shared__ uint32_t symF[320];
    __shared__ uint32_t metaF[20];
    __shared__ uint32_t codes[288];
    __shared__ uint32_t dcodes[32];
    __shared__ uint32_t ring_words[ZZ_RING_WORDS];
    __shared__ uint32_t misc[64];
    uint32_t* distF = symF + 288;
    huff_scratch H;
    H.rec_freq = (uint32_t*)(hs);
    H.rec_id = (uint16_t*)(hs + 1152);
    H.t_freq = (uint32_t*)(hs + 1728);
    H.t_left = (uint16_t*)(hs + 4032);
    H.t_right = (uint16_t*)(hs + 5184);
    H.t_bits = (uint8_t*)(hs + 6336);
    uint8_t* lens = (uint8_t*)(hs + 6912);
    uint8_t* metaLens = (uint8_t*)(hs + 7232);
    uint32_t* metaCodes = (uint32_t*)(hs + 7264);
    uint16_t* rle = (uint16_t*)(hs + 7344);

    const int lane = lane_id();
    st_state S;
    S.T = T; S.base = P.src; S.end = P.src + P.n;
    // (one workgroup per range; the sequential stream is the one-range case)
    const uint32_t rg = blockIdx.x;
    const uint64_t rstart = Q.range_step ? (uint64_t)rg * Q.range_step : 0;
    const bool rlast = !Q.range_step || rg + 1 == gridDim.x;
    const uint64_t rend = rlast ? P.n : rstart + Q.range_step;
    const uint64_t cend = rlast ? rend : rend - 1;                       // what the compressing AddData sees (zzflate.cpp:116)
    uint8_t* const scratch = Q.scratch + (uint64_t)rg * ZZ_ST_SCRATCH_BYTES;
    S.tokens = (st_token*)scratch;
    S.cov = (unsigned long long*)(scratch + ZZ_ST_MAX_RECORDS * 8);
    S.mst = S.cov + ZZ_ST_WORDS;
    uint32_t* mcount = (uint32_t*)(S.mst + ZZ_ST_WORDS);

    for (int i = lane; i < ZZ_HASH_SIZE; i += ZZ_WAVE) T[i] = 0;       // cold table (encoder.cpp:533-536)
    bitring ring;
    ring_init(ring, ring_words, P.slots + (uint64_t)rg * P.slot_stride);
    uint32_t* const out0 = ring.out32;

    // "bytes stored" as EnsureOutputLength sees them (outputbitstream.h:167-176): exact at the last Flush (the end of a
    // stored block, outputbitstream.h:155-160), whole 64-bit words of the packer since then
    uint64_t flush_bits = 0;
    uint32_t nlog = 0;
    auto stored_now = [&]() -> uint64_t {
        const uint64_t bits = (uint64_t)(ring.out32 - out0) * 32 + ring.bitpos;
        return (flush_bits >> 3) + (((bits - flush_bits) >> 6) << 3);
    };
    uint64_t pos = rstart;                                               // AddData loop (encoder.cpp:539-552)
    while (pos < cend) {
        const uint64_t remaining = cend - pos;
        bool final = rlast;
        uint32_t byteCount = (uint32_t)remaining;
        if (remaining > ZZ_ST_MAX_BLOCK) { byteCount = ZZ_ST_MAX_BLOCK; final = false; }   // :518-522
        S.bs = pos;
        const uint8_t* src = P.src + pos;
        const uint32_t nwords = (byteCount + 63) / 64 + 1;
        for (uint32_t i = lane; i < nwords; i += ZZ_WAVE) { S.cov[i] = 0; S.mst[i] = 0; }
        for (int i = lane; i < 320; i += ZZ_WAVE) symF[i] = 0;
        if (lane < 20) metaF[lane] = 0;
        __syncthreads();

        uint32_t ntok = 0;
        const uint32_t length = st_token_pass(S, byteCount, &ntok);
        __syncthreads();

        // histograms (encoder.cpp:442-471)
        const uint32_t nblk = (length + 63) >> 6;
        {
            uint32_t carry = 0;
            for (uint32_t b0 = 0; b0 < nblk; b0 += 64) {
                const uint32_t b = b0 + lane;
                const uint32_t cnt = b < nblk ? (uint32_t)__builtin_popcountll(S.mst[b]) : 0;
                const uint32_t incl = wave_scan_incl(cnt);
                if (b < nblk) mcount[b] = carry + incl - cnt;
                carry += readlane(incl, 63);
            }
        }
        __syncthreads();
        for (uint32_t b = 0; b < nblk; ++b) {
            const unsigned long long cw = S.cov[b], mw = S.mst[b];
            const uint32_t q = (b << 6) + lane;
            if (q < length) {
                if (!((cw >> lane) & 1)) atomicAdd(&symF[src[q]], 1u);
                else if ((mw >> lane) & 1) {
                    const st_token t = S.tokens[mcount[b] + __builtin_popcountll(mw & ((1ull << lane) - 1))];
                    uint32_t sym, eb, ev, bucket;
                    length_symbol(t.len, sym, eb, ev);
                    atomicAdd(&symF[sym], 1u);
                    dist_symbol(t.dist, bucket, eb, ev);
                    atomicAdd(&distF[bucket], 1u);
                }
            }
        }
        __syncthreads();

        // code construction (lane 0): encoder.cpp:255-271
        if (lane == 0) {
            symF[256] += 1;
            int64_t bits = 0;
            calc_lengths(H, symF, 286, 15, lens);
            int nrec = rle_lengths(lens, 286, rle, 0, metaF);
            for (int i = 0; i < 286; ++i) {
                uint32_t eb = i < 265 || i == 285 ? 0 : (uint32_t)(i - 261) >> 2;
                bits += (int64_t)symF[i] * (lens[i] + eb);
            }
            calc_lengths(H, distF, 30, 15, lens + 288);
            nrec = rle_lengths(lens + 288, 30, rle, nrec, metaF);
            for (int i = 0; i < 30; ++i) {
                uint32_t eb = i < 4 ? 0 : (uint32_t)(i - 2) >> 1;
                bits += (int64_t)distF[i] * (lens[288 + i] + eb);
            }
            calc_lengths(H, metaF, 19, 7, metaLens);
            int64_t total = 3 + 5 + 5 + 4 + 3 * 19 + bits;
            for (int i = 0; i < nrec; ++i) {
                const uint32_t v = rle[i] & 0xFF;
                total += metaLens[v] + (v == 16 ? 2 : v == 17 ? 3 : v == 18 ? 7 : 0);
            }
            misc[0] = (uint32_t)((total + 8) / 8);
            misc[2] = (uint32_t)nrec;
        }
        __syncthreads();
        const uint32_t required = misc[0], nrec = misc[2];

        if (required >= length) {
            // UncompressedFallback (encoder.cpp:305-317): stored blocks of at most 65535 bytes, BFINAL as passed in
            uint32_t written = 0;
            while (written < length) {
                const uint32_t c = length - written < 0xFFFF ? length - written : 0xFFFF;
                zz_log_ensure(Q.ctl, nlog, stored_now(), 6 + (uint64_t)c);          // encoder.cpp:488
                st_stored_block(ring, src + written, c, final && c == length - written);
                flush_bits = (uint64_t)(ring.out32 - out0) * 32 + ring.bitpos;       // WriteBytes flushed the packer
                written += c;
            }
        } else {
            if (lane == 0) {
                generate_codes(lens, 286, codes, misc + 16);
                generate_codes(lens + 288, 30, dcodes, misc + 16);
                generate_codes(metaLens, 19, metaCodes, misc + 16);
            }
            __syncthreads();
            zz_log_ensure(Q.ctl, nlog, stored_now(), required);                      // encoder.cpp:276
            const bool bfinal = length < byteCount ? false : final;     // :280
            ring_append_uniform(ring, (bfinal ? 1u : 0u) | (2u << 1) | (29u << 3) | (29u << 8) | (15u << 13), 17);
            {
                const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
                uint32_t v = 0;
                for (int i = 0; i < 19; ++i) if (lane == i) v = metaLens[order[i]];
                ring_append(ring, v, lane < 19 ? 3 : 0);
            }
            for (uint32_t r0 = 0; r0 < nrec; r0 += 64) {
                const uint32_t i = r0 + lane;
                uint32_t bits = 0, nb = 0;
                if (i < nrec) {
                    const uint32_t v = rle[i] & 0xFF, pay = rle[i] >> 8;
                    const uint32_t mc = metaCodes[v];
                    nb = mc >> 16; bits = mc & 0xFFFF;
                    if (v == 16) { bits |= (pay - 3) << nb; nb += 2; }
                    else if (v == 17) { bits |= (pay - 3) << nb; nb += 3; }
                    else if (v == 18) { bits |= (pay - 11) << nb; nb += 7; }
                }
                ring_append(ring, bits, nb);
            }
            for (uint32_t b = 0; b < nblk; ++b) {                       // WriteRecords (:149-169) by position
                const unsigned long long cw = S.cov[b], mw = S.mst[b];
                const uint32_t q = (b << 6) + lane;
                uint64_t bits = 0; uint32_t nb = 0;
                const unsigned long long live = ~cw | mw;
                if (q < length && ((live >> lane) & 1)) {
                    if (!((cw >> lane) & 1)) {
                        const uint32_t cd = codes[src[q]];
                        bits = cd & 0xFFFF; nb = cd >> 16;
                    } else {
                        const st_token t = S.tokens[mcount[b] + __builtin_popcountll(mw & ((1ull << lane) - 1))];
                        uint32_t sym, eb, ev, bucket, deb, dev;
                        length_symbol(t.len, sym, eb, ev);
                        const uint32_t lc = codes[sym];
                        uint32_t ln = lc >> 16;
                        uint64_t v = (lc & 0xFFFF) | ((uint64_t)ev << ln);
                        ln += eb;
                        dist_symbol(t.dist, bucket, deb, dev);
                        const uint32_t dc = dcodes[bucket];
                        v |= (uint64_t)(dc & 0xFFFF) << ln;
                        ln += dc >> 16;
                        v |= (uint64_t)dev << ln;
                        ln += deb;
                        bits = v; nb = ln;
                    }
                }
                if (live == 0) continue;
                ring_append64(ring, bits, nb);
            }
            const uint32_t cd = codes[256];
            ring_append_uniform(ring, cd & 0xFFFF, cd >> 16);           // :300
        }
        if (ring.flushed >= (1u << 24)) {                                // long streams: slide the ring's origin
            const uint32_t kw = ring.flushed & ~(uint32_t)(ZZ_RING_WORDS - 1);
            ring.out32 += kw; ring.bitpos -= kw * 32; ring.flushed -= kw;
        }
        pos += length;
        __syncthreads();
    }
    if (!rlast) st_stored_block(ring, P.src + rend - 1, 1, false);      // zzflate.cpp:118-120
    const uint64_t bytes = (uint64_t)(ring.out32 - out0) * 4 + ring_finish(ring);
    if (lane == 0) {
        P.sizes[rg] = (uint32_t)bytes;
        if (bytes > (uint64_t)P.slot_stride) atomicOr(P.err, 1u);
        if (Q.ctl.log_n) *Q.ctl.log_n = nlog;
    }
}

// Level 0 over the same ranges: stored blocks of at most 65535 bytes (encoder.cpp:482-502) straight to their final place --
// every size is known beforehand. One workgroup per range.
__host__ __device__ __forceinline__ uint64_t l0_chain_bytes(uint64_t len) { return len + 5 * ((len + 0xFFFE) / 0xFFFF); }
__host__ __device__ __forceinline__ uint64_t l0_range_bytes(uint64_t len, bool last) { return last ? l0_chain_bytes(len) : l0_chain_bytes(len - 1) + 6; }
__global__ __launch_bounds__(256) void k_ranges_l0(const uint8_t* src, uint64_t n, uint64_t step, uint8_t* dst)
{
    const uint32_t rg = blockIdx.x, nr = gridDim.x;
    uint64_t o = 0;
    for (uint32_t r = 0; r < rg; ++r) o += l0_range_bytes(step, false);
    const uint64_t rstart = (uint64_t)rg * step;
    const bool last = rg + 1 == nr;
    const uint64_t rend = last ? n : rstart + step;
    const uint64_t cend = last ? rend : rend - 1;
    uint8_t* out = dst + o;
    for (uint64_t p = rstart; p < cend; p += 0xFFFF) {
        const uint32_t c = (uint32_t)(cend - p < 0xFFFF ? cend - p : 0xFFFF);
        if (threadIdx.x == 0) {
            out[0] = (last && p + c == cend) ? 1 : 0;
            out[1] = (uint8_t)c; out[2] = (uint8_t)(c >> 8); out[3] = (uint8_t)~c; out[4] = (uint8_t)(~c >> 8);
        }
        coop_copy(out + 5, src + p, c, threadIdx.x, blockDim.x);
        out += 5 + c;
    }
    if (!last && threadIdx.x == 0) { out[0] = 0; out[1] = 1; out[2] = 0; out[3] = 0xFE; out[4] = 0xFF; out[5] = src[rend - 1]; }
}

}  // namespace zz
