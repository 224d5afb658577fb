// zz_level2.h -- levels 2 and 3 (one code path in the reference, encoder.cpp:506-527): two-pass dynamic
// Huffman, one packet per wavefront.
//
// Restates WriteBlock2Pass (encoder.cpp:217-303) for a packet (zzflate.cpp:101-125), bit-exact:
//
//   token pass   FirstPass + AddHashEntries (encoder.cpp:375-440, 474-480). At this level EVERY position a
//                match covers is inserted, in ascending order, so the candidate a probe at j sees is simply
//                "the previous position with the same 13-bit hash" -- independent of the parse (SURVEY.md 7,
//                hard part 2). The wave therefore inserts aligned blocks of 64 positions at once (LDS table
//                as pos+1, one read/write/read-back round trip, ballot loop over the hashes that occur more
//                than once in the block) and gets all 64 candidates in registers; only the greedy choice
//                "first probe position whose forward+backward match is >= 4" is serial, and it runs as a
//                ballot walk inside the block. Exact lengths (forward up to 258, backward up to the pending
//                literal count) are measured by the whole wave, 4 bytes per lane, for the chosen probe only.
//                The two positions the reference never inserts (packet byte 0 and the first byte of a batch
//                that the previous batch did not overrun, encoder.cpp:383,435-436) are skipped.
//   records      The parse is final 320 positions behind the probe front (a later match can extend backward over at
//                most 258 pending literals), so the token pass itself turns finished 64-position blocks into the
//                reference's record sequence (literal | match, encoder.cpp:420,428-431): covered / match-start
//                bits sit in a 16-word LDS window, live positions are compacted to a dense u16 record array.
//   histograms   GetFrequencies (encoder.cpp:442-471) falls out of the same pass: literals are counted when their
//                block is finished, length / distance symbols when a match is published (LDS atomics on packed
//                16-bit counters -- a packet has fewer than 32768 records).
//   code build   CalcLengths (huffman.cpp:122-154): heap Huffman + frequency-floor length limit. Code lengths
//                depend on libstdc++'s make_heap/pop_heap/push_heap element movements (ties!), so lane 0
//                replays bits/stl_heap.h (GCC 11: __push_heap :134-148, __adjust_heap :223-248) on LDS arrays.
//                huffman::generate (huffman.h:49-81) and FromLengths (huffman.cpp:158-216) follow.
//   emission     dynamic header (encoder.cpp:280-293: always 286/30/19 codes), body (WriteRecords :149-169
//                with merged length codes :121-133), or UncompressedFallback (:305-317) when the dynamic block
//                would not be smaller (:271-274).
#pragma once
#include <type_traits>
#include "zz_checksum.h"
#include "zz_emit.h"
#include "zz_level1.h"

namespace zz {

#define ZZ_L2_MAX_TOKENS 8192                       // every match covers >= 4 bytes of a <= 32768-byte packet
// per resident workgroup: matches as u32 (length symbol - 257 [27:23], length extra value [22:18], distance code
// [17:13], distance extra value [12:0]), records as u16 (literal byte | ZZ_L2_REC_MATCH)
#define ZZ_L2_SNAP_WORDS 164u     // a snapshot of the packed symbol counters [0, 158), then [158] the matches among the records so far, [159] the records, [160] the matches that have arrived
#define ZZ_L2_SCRATCH_BYTES (ZZ_L2_MAX_TOKENS * 4 + ZZ_MAX_PACKET * 2 + 2 * ZZ_L2_SNAP_WORDS * 4 + 768)
#define ZZ_L2_REC_MATCH 0x100u
#define ZZ_L2_REC_NONE 0x200u
#define ZZ_L2_WIN 16                                // bitmap window, words of 64 positions (11 are live at a time)
#define ZZ_L2_LAG 5                                 // a block is final once the probe front is 5 blocks ahead
#ifndef ZZ_L2_HELPER_DYNPRIO
#define ZZ_L2_HELPER_DYNPRIO 7u                     // matches per 64-position block from which the helper runs at the parser's priority
#endif
#define ZZ_L2_HIST_WORDS 160                        // 286 lit/len + 30 distance counters, two per word
#define ZZ_L2_LDS_BYTES (16384 + 560 + 2 * ZZ_L2_WIN * 8 + ZZ_L2_HIST_WORDS * 4)   // (+ 16: the slot non-inserting lanes use)

__device__ __forceinline__ uint32_t mbcnt(uint64_t m)   // set bits of m below this lane
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ void hist_add(uint32_t* histP, uint32_t idx) { atomicAdd(&histP[idx >> 1], 1u << ((idx & 1) << 4)); }

// One finished block of 64 positions -> records + literal counts; frees its window slot. `byte` = src[64F + lane].
__device__ __forceinline__ uint32_t l2_finish_block(uint64_t* covw, uint64_t* mstw, uint32_t* histP, uint16_t* recs,
                                                    uint32_t nrec, uint32_t F, uint32_t n, uint32_t byte, uint32_t* nmatch = nullptr)
{
    const int lane = lane_id();
    const uint32_t s = F & (ZZ_L2_WIN - 1);
    const uint64_t cw = covw[s], mw = mstw[s];
    ZZ_WAVE_SYNC();
    if (lane == 0) { covw[s] = 0; mstw[s] = 0; }
    const uint32_t p = (F << 6) + (uint32_t)lane;
    const bool lit = p < n && !((cw >> lane) & 1);
    const bool ms = p < n && ((mw >> lane) & 1);
    const uint64_t live = ballot(lit || ms);
    if (nmatch) *nmatch += (uint32_t)__builtin_popcountll(ballot(ms));
    if (lit) hist_add(histP, byte);
    if (lit || ms) recs[nrec + mbcnt(live)] = (uint16_t)(ms ? ZZ_L2_REC_MATCH : byte);
    return nrec + (uint32_t)__builtin_popcountll(live);
}

// ---- 64-bit fragments through the bit ring ----------------------------------------------------------------
__device__ __forceinline__ void ring_flush_all_full(bitring& r)
{
    while ((r.bitpos >> 5) > r.flushed) {
        const uint32_t full = r.bitpos >> 5;
        const uint32_t upto = full - r.flushed > 64 ? r.flushed + 64 : full;
        ZZ_WAVE_SYNC();
        const uint32_t w = r.flushed + lane_id();
        if (w < upto) {
            uint32_t v = r.ring[w & (ZZ_RING_WORDS - 1)];
            r.ring[w & (ZZ_RING_WORDS - 1)] = 0;
            if (w == r.hold) *r.holdp = v;      // shared with the other emitter's last word: merged at the end
            else r.out32[w] = v;
        }
        r.flushed = upto;
        ZZ_WAVE_SYNC();
    }
}
// a ring that continues a bit stream at bit `bitpos`; its first word is held back (see bitring::hold)
__device__ __forceinline__ void ring_init_at(bitring& r, uint32_t* lds_ring, uint8_t* out, uint32_t bitpos, uint32_t* holdp)
{
    ring_init(r, lds_ring, out);
    r.bitpos = bitpos;
    r.flushed = bitpos >> 5;
    r.hold = bitpos >> 5;
    r.holdp = holdp;
}
// every lane appends nb <= 48 bits (so one append adds at most 96 words; the ring holds 128)
__device__ __forceinline__ void ring_append64(bitring& r, uint64_t bits, uint32_t nb)
{
    const uint32_t incl = wave_scan_incl(nb);
    const uint32_t total = readlane(incl, 63);
    const uint32_t o = r.bitpos + incl - nb;
    const uint32_t sh = o & 31;
    const uint32_t w = o >> 5;
    {   // all three words every time (ORs with zero past the fragment's end): no lane-mask regions
        const uint64_t b = nb ? bits : 0ull;
        const uint64_t x = b << sh;
        const uint32_t top = (uint32_t)(((b >> 32) << sh) >> 32);          // what the shift pushed out of 64 bits
        atomicOr(&r.ring[w & (ZZ_RING_WORDS - 1)], (uint32_t)x);
        atomicOr(&r.ring[(w + 1) & (ZZ_RING_WORDS - 1)], (uint32_t)(x >> 32));
        atomicOr(&r.ring[(w + 2) & (ZZ_RING_WORDS - 1)], top);
    }
    r.bitpos += total;
    ring_flush_all_full(r);
}

__device__ __forceinline__ void ring_append_uniform64(bitring& r, uint32_t bits, uint32_t nb)   // hold-aware twin of ring_append_uniform
{
    ring_append64(r, lane_id() == 0 ? bits : 0u, lane_id() == 0 ? nb : 0u);
}
// ring_finish for a ring that may still hold its first word back
__device__ __forceinline__ uint32_t ring_finish_hold(bitring& r)
{
    ring_pad_to_byte(r);
    const uint32_t bytes = r.bitpos >> 3;
    const uint32_t words = (bytes + 3) >> 2;
    ZZ_WAVE_SYNC();
    const uint32_t w = r.flushed + lane_id();
    if (w < words) {
        const uint32_t v = r.ring[w & (ZZ_RING_WORDS - 1)];
        if (w == r.hold) *r.holdp = v;
        else r.out32[w] = v;
    }
    return bytes;
}

// ---- Huffman code construction (lane 0, LDS scratch) ---------------------------------------------------------
struct huff_scratch {
    uint32_t* rec_freq;   // [288] heap records: frequency            (huffman.h:10-14)
    uint16_t* rec_id;     // [288]               tree index
    uint32_t* t_freq;     // [576] tree items                          (huffman.h:16-22)
    uint16_t* t_left;     // [576] leaf: symbol; internal: left child
    uint16_t* t_right;    // [576] 0xFFFF for a leaf
    uint8_t* t_bits;      // [576]
};

// bits/stl_heap.h __push_heap with comparator `greater` on frequency (huffman.cpp:55-62)
__device__ __forceinline__ void heap_push(huff_scratch& S, int hole, int top, uint32_t vf, uint16_t vi)
{
    int parent = (hole - 1) / 2;
    while (hole > top && S.rec_freq[parent] > vf) {
        S.rec_freq[hole] = S.rec_freq[parent];
        S.rec_id[hole] = S.rec_id[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    S.rec_freq[hole] = vf;
    S.rec_id[hole] = vi;
}
// bits/stl_heap.h __adjust_heap
__device__ __forceinline__ void heap_adjust(huff_scratch& S, int hole, int len, uint32_t vf, uint16_t vi)
{
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (S.rec_freq[child] > S.rec_freq[child - 1]) child--;
        S.rec_freq[hole] = S.rec_freq[child];
        S.rec_id[hole] = S.rec_id[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        S.rec_freq[hole] = S.rec_freq[child - 1];
        S.rec_id[hole] = S.rec_id[child - 1];
        hole = child - 1;
    }
    heap_push(S, hole, top, vf, vi);
}

// huffman.cpp:67-120 CalculateTree; returns the maximum leaf depth
__device__ inline int calculate_tree(huff_scratch& S, const uint32_t* freqs, int n, uint32_t minFreq, int* ntree)
{
    int nrec = 0, nt = 0;
    for (int i = 0; i < n; ++i) {
        uint32_t f = freqs[i];
        if (f != 0) {
            if (f < minFreq) f = minFreq;
            S.rec_freq[nrec] = f;
            S.rec_id[nrec] = (uint16_t)i;
            nrec++;
        }
        S.t_freq[nt] = f; S.t_left[nt] = (uint16_t)i; S.t_right[nt] = 0xFFFF; S.t_bits[nt] = 0;
        nt++;
    }
    // make_heap (stl_heap.h:339-360)
    if (nrec >= 2) {
        for (int parent = (nrec - 2) / 2;; --parent) {
            heap_adjust(S, parent, nrec, S.rec_freq[parent], S.rec_id[parent]);
            if (parent == 0) break;
        }
    }
    while (nrec >= 2) {
        // pop_heap (stl_heap.h:253-265): last element becomes the value sifted from the root
        uint32_t af = S.rec_freq[0]; uint16_t ai = S.rec_id[0];
        nrec--;
        if (nrec >= 1) heap_adjust(S, 0, nrec, S.rec_freq[nrec], S.rec_id[nrec]);
        uint32_t bf = S.rec_freq[0]; uint16_t bi = S.rec_id[0];
        nrec--;
        if (nrec >= 1) heap_adjust(S, 0, nrec, S.rec_freq[nrec], S.rec_id[nrec]);
        const uint32_t sum = af + bf;
        S.t_freq[nt] = sum; S.t_left[nt] = ai; S.t_right[nt] = bi; S.t_bits[nt] = 0;
        nrec++;
        heap_push(S, nrec - 1, 0, sum, (uint16_t)nt);
        nt++;
    }
    int maxLength = 0;
    for (int i = nt - 1; i != 0; --i) {                      // huffman.cpp:108, index 0 skipped
        if (S.t_right[i] == 0xFFFF) { if (S.t_bits[i] > maxLength) maxLength = S.t_bits[i]; continue; }
        const uint8_t b = (uint8_t)(S.t_bits[i] + 1);
        S.t_bits[S.t_left[i]] = b;
        S.t_bits[S.t_right[i]] = b;
    }
    *ntree = nt;
    return maxLength;
}

// huffman.cpp:122-154 CalcLengths
__device__ inline void calc_lengths(huff_scratch& S, const uint32_t* freqs, int n, int maxlen, uint8_t* out)
{
    uint32_t minFreq = 0;
    int nt = 0;
    for (;;) {
        const int mx = calculate_tree(S, freqs, n, minFreq, &nt);
        if (mx <= maxlen) {
            for (int i = 0; i < n; ++i)   // leaves are tree items 0..n-1 in symbol order
                out[i] = S.t_freq[i] == 0 ? 0 : (S.t_bits[i] > 1 ? S.t_bits[i] : 1);
            return;
        }
        uint32_t total = 0;
        for (int i = 0; i < n; ++i) total += freqs[i];
        const uint32_t step = total >> maxlen;
        minFreq += step > 1 ? step : 1;
    }
}

// ---- the same construction, wave-cooperative (packet kernel) ------------------------------------------------------
// Leaf setup, the length limit test and the output are parallel over symbols; only the heap replay runs on lane 0,
// on records packed as frequency << 10 | tree index (one LDS word per heap slot: half the dependent LDS round trips
// of the two-array form above). Frequencies stay below 2^22: a packet has < 32768 records and the floor grows by
// at most a few units per retry. Leaves are tree items 0..n-1, so "is a leaf" is an index test and internal items
// keep their children in one word (left | right << 16).
#define ZZ_HKEY_SHIFT 10
__device__ __forceinline__ void hkey_push(uint32_t* hk, int hole, int top, uint32_t v)
{
    int parent = (hole - 1) / 2;
    while (hole > top) {
        const uint32_t pk = hk[parent];
        if (!((pk >> ZZ_HKEY_SHIFT) > (v >> ZZ_HKEY_SHIFT))) break;
        hk[hole] = pk;
        hole = parent;
        parent = (hole - 1) / 2;
    }
    hk[hole] = v;
}
__device__ __forceinline__ void hkey_adjust(uint32_t* hk, int hole, int len, uint32_t v)
{
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        uint32_t a = hk[child];
        const uint32_t b = hk[child - 1];
        if ((a >> ZZ_HKEY_SHIFT) > (b >> ZZ_HKEY_SHIFT)) { child--; a = b; }
        hk[hole] = a;
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        hk[hole] = hk[child - 1];
        hole = child - 1;
    }
    hkey_push(hk, hole, top, v);
}

// CalculateTree (huffman.cpp:67-120) for all lanes; returns the maximum leaf depth (wave-uniform)
__device__ inline int calculate_tree_w(huff_scratch& S, const uint32_t* freqs, int n, uint32_t minFreq)
{
    const int lane = lane_id();
    uint32_t* hk = S.rec_freq;                 // [288] heap records
    uint32_t* t_lr = (uint32_t*)S.t_left;      // [576] children of internal items (t_left and t_right are adjacent)
    uint8_t* t_bits = S.t_bits;
    uint32_t nrec = 0;
    for (int i0 = 0; i0 < n; i0 += ZZ_WAVE) {  // :70-83 records of the non-zero symbols, in symbol order
        const int i = i0 + lane;
        uint32_t f = i < n ? freqs[i] : 0;
        const uint64_t nz = ballot(f != 0);
        if (f != 0) {
            if (f < minFreq) f = minFreq;
            hk[nrec + mbcnt(nz)] = (f << ZZ_HKEY_SHIFT) | (uint32_t)i;
        }
        if (i < n) t_bits[i] = 0;
        nrec += (uint32_t)__builtin_popcountll(nz);
    }
    ZZ_WAVE_SYNC();
    int nt = n;
#ifndef ZZ_L2_MAKE_HEAP_W
#define ZZ_L2_MAKE_HEAP_W 1
#endif
#if ZZ_L2_MAKE_HEAP_W
    // make_heap (stl_heap.h:339-360) sifts the parents down one after the other, last parent first. A sift stays inside its parent's
    // subtree, the parents of one level of the heap have disjoint subtrees, and every deeper level's parents (higher indices) come
    // before: level by level, deepest first, a lane per parent is the same sequence of moves.
    if (nrec >= 2) {
        const int nr = (int)nrec, last = (nr - 2) / 2;
        for (int L = 31 - __builtin_clz((uint32_t)last + 1u); L >= 0; --L) {
            const int first = (1 << L) - 1, end = last < (2 << L) - 2 ? last : (2 << L) - 2;
            for (int p0 = first; p0 <= end; p0 += ZZ_WAVE) {
                const int p = p0 + lane;
                if (p <= end) hkey_adjust(hk, p, nr, hk[p]);
            }
            ZZ_WAVE_SYNC();
        }
    }
#endif
    if (lane == 0) {
        int nr = (int)nrec;
#if !ZZ_L2_MAKE_HEAP_W
        if (nr >= 2) {                         // make_heap (stl_heap.h:339-360)
            for (int parent = (nr - 2) / 2;; --parent) {
                hkey_adjust(hk, parent, nr, hk[parent]);
                if (parent == 0) break;
            }
        }
#endif
        while (nr >= 2) {                      // :92-104, pop_heap x2 (stl_heap.h:253-265) + push_heap
            const uint32_t a = hk[0];
            nr--;
            if (nr >= 1) hkey_adjust(hk, 0, nr, hk[nr]);
            const uint32_t b = hk[0];
            nr--;
            if (nr >= 1) hkey_adjust(hk, 0, nr, hk[nr]);
            const uint32_t sum = (a >> ZZ_HKEY_SHIFT) + (b >> ZZ_HKEY_SHIFT);
            const uint32_t mask = (1u << ZZ_HKEY_SHIFT) - 1;
            t_lr[nt] = (a & mask) | ((b & mask) << 16);
            nr++;
            hkey_push(hk, nr - 1, 0, (sum << ZZ_HKEY_SHIFT) | (uint32_t)nt);
            nt++;
        }
        if (nt > n) t_bits[nt - 1] = 0;
        for (int i = nt - 1; i >= n; --i) {    // :108-118 depths, root first; leaves are handled below
            const uint32_t lr = t_lr[i];
            const uint8_t b = (uint8_t)(t_bits[i] + 1);
            t_bits[lr & 0xFFFF] = b;
            t_bits[lr >> 16] = b;
        }
    }
    ZZ_WAVE_SYNC();
    uint32_t mx = 0;                           // :109-111 deepest leaf; item 0 is never looked at (the loop stops at 1)
    for (int i = lane; i < n; i += ZZ_WAVE)
        if (i != 0 && t_bits[i] > mx) mx = t_bits[i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const uint32_t y = (uint32_t)__shfl_xor((int)mx, o); mx = y > mx ? y : mx; }
    return (int)uniform(mx);
}

// CalcLengths (huffman.cpp:122-154) for all lanes
__device__ inline void calc_lengths_w(huff_scratch& S, const uint32_t* freqs, int n, int maxlen, uint8_t* out)
{
    const int lane = lane_id();
    uint32_t minFreq = 0;
    for (;;) {
        const int mx = calculate_tree_w(S, freqs, n, minFreq);
        if (mx <= maxlen) {
            for (int i = lane; i < n; i += ZZ_WAVE) {
                const uint32_t b = S.t_bits[i];
                out[i] = freqs[i] == 0 ? 0 : (uint8_t)(b > 1 ? b : 1);
            }
            ZZ_WAVE_SYNC();
            return;
        }
        uint32_t part = 0;
        for (int i = lane; i < n; i += ZZ_WAVE) part += freqs[i];
        const uint32_t step = wave_sum(part) >> maxlen;
        minFreq += step > 1 ? step : 1;
    }
}

// huffman.h:49-81 generate: canonical codes, stored bit-reversed, packed (len << 16) | bits; 0 for unused.
// work: 32 words of LDS (bl_count[16], next_code[16]) -- runtime-indexed private arrays would go to scratch.
__device__ inline void generate_codes(const uint8_t* lengths, int n, uint32_t* codes, uint32_t* work)
{
    uint32_t* bl_count = work;
    uint32_t* next_code = work + 16;
    for (int b = 0; b < 16; ++b) bl_count[b] = 0;
    for (int i = 0; i < n; ++i) bl_count[lengths[i]]++;
    bl_count[0] = 0;
    uint32_t c = 0;
    next_code[0] = 0;
    for (int b = 1; b < 16; ++b) { c = (c + bl_count[b - 1]) << 1; next_code[b] = c; }
    for (int i = 0; i < n; ++i) {
        const uint32_t len = lengths[i];
        uint32_t v = 0;
        if (len) {
            const uint32_t nc = next_code[len];
            next_code[len] = nc + 1;
            v = (len << 16) | bitrev(nc, len);
        }
        codes[i] = v;
    }
}

// generate (huffman.h:49-81) for all lanes: symbols of one length get consecutive codes in symbol order, so a
// symbol's code is next_code[len] + (number of earlier symbols with that length) -- counted with ballots, 64
// symbols at a time. work: 32 words of LDS.
__device__ inline void generate_codes_w(const uint8_t* lengths, int n, uint32_t* codes, uint32_t* work)
{
    const int lane = lane_id();
    uint32_t* bl_count = work;
    uint32_t* next_code = work + 16;
    if (lane < 16) bl_count[lane] = 0;
    ZZ_WAVE_SYNC();
    for (int i = lane; i < n; i += ZZ_WAVE) { const uint32_t l = lengths[i]; if (l) atomicAdd(&bl_count[l], 1u); }
    ZZ_WAVE_SYNC();
    {
        const uint32_t cnt = lane < 16 ? bl_count[lane] : 0;
        uint32_t c = 0, mine = 0;
        for (int b = 1; b < 16; ++b) { c = (c + readlane(cnt, b - 1)) << 1; if (lane == b) mine = c; }
        if (lane < 16) next_code[lane] = mine;
    }
    ZZ_WAVE_SYNC();
    for (int i0 = 0; i0 < n; i0 += ZZ_WAVE) {
        const int i = i0 + lane;
        const uint32_t len = i < n ? lengths[i] : 0;
        uint32_t v = 0;
        uint64_t todo = ballot(len != 0);
        while (todo) {
            const uint32_t lv = readlane(len, __builtin_ctzll(todo));
            const uint64_t same = ballot(len == lv);
            const uint32_t nc = next_code[lv];
            ZZ_WAVE_SYNC();
            if (len == lv) v = (len << 16) | bitrev(nc + mbcnt(same), len);
            if (lane == 0) next_code[lv] = nc + (uint32_t)__builtin_popcountll(same);
            ZZ_WAVE_SYNC();
            todo &= ~same;
        }
        if (i < n) codes[i] = v;
    }
    ZZ_WAVE_SYNC();
}

// huffman.cpp:158-216 FromLengths/AddRecords: RLE of one code-length array into (value, payload) records,
// packed value | payload << 8; meta frequencies accumulate. Returns the new record count.
__device__ inline int rle_add(uint16_t* recs, int nv, uint32_t* metaF, int value, int count)
{
    if (count == 0) return nv;
    if (value == 0) {
        while (count >= 3) {
            int w = count < 138 ? count : 138;
            count -= w;
            int sym = w < 11 ? 17 : 18;
            recs[nv++] = (uint16_t)(sym | (w << 8)); metaF[sym]++;
        }
    } else {
        recs[nv++] = (uint16_t)value; metaF[value]++;
        count--;
        while (count >= 3) {
            int w = count < 6 ? count : 6;
            count -= w;
            recs[nv++] = (uint16_t)(16 | (w << 8)); metaF[16]++;
        }
    }
    for (int i = 0; i < count; ++i) { recs[nv++] = (uint16_t)value; metaF[value]++; }
    return nv;
}
__device__ inline int rle_lengths(const uint8_t* lengths, int n, uint16_t* recs, int nv, uint32_t* metaF)
{
    int cur = -1, count = 0;
    for (int i = 0; i < n; ++i) {
        if (lengths[i] == cur) { count++; continue; }
        nv = rle_add(recs, nv, metaF, cur, count);
        cur = lengths[i];
        count = 1;
    }
    return rle_add(recs, nv, metaF, cur, count);
}

// The same for all lanes, a lane per run of equal lengths: what a run of c lengths v becomes is a closed form of AddRecords'
// loops (:191-216) -- v == 0: c = 138 q + r gives q records (18, 138), then (17 or 18, r) if r >= 3, else r plain zeros; v != 0: the
// length itself, then c - 1 = 6 q + r gives q records (16, 6), then (16, r) if r >= 3, else r times the length again -- so a run
// knows how many records it writes, a prefix sum over the runs places them, and the meta frequencies take one atomic per kind.
#ifndef ZZ_L2_RLE_W
#define ZZ_L2_RLE_W 1
#endif
template <int N>
__device__ __forceinline__ int rle_lengths_w(const uint8_t* lengths, uint16_t* recs, int nv, uint32_t* metaF)
{
    constexpr int K = (N + 63) / 64;
    const int lane = lane_id();
    uint32_t base = (uint32_t)nv;
    uint32_t carry = 0;                                      // where the run that reaches into this block of 64 began
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
        const uint32_t i = 64u * k + (uint32_t)lane;
        const bool in = i < (uint32_t)N;
        const uint32_t val = in ? lengths[i] : 0xFFu;
        const uint32_t pv = (in && i > 0) ? lengths[i - 1] : 0xFFu;
        const uint32_t nx = i + 1 < (uint32_t)N ? lengths[i + 1] : 0xFFu;
        const uint64_t stm = ballot(in && val != pv);        // runs that begin here
        const uint64_t below = stm & ((2ull << lane) - 1ull);
        const uint32_t first = below ? 64u * k + 63u - (uint32_t)__builtin_clzll(below) : carry;
        const bool e = in && val != nx;                      // the lane at a run's last position writes the run
        const uint32_t c = e ? i + 1u - first : 1u;
        const uint32_t q = val == 0 ? c / 138u : (c - 1u) / 6u;
        const uint32_t r = val == 0 ? c - 138u * q : (c - 1u) - 6u * q;
        const uint32_t cnt = e ? (val != 0 ? 1u : 0u) + q + (r >= 3u ? 1u : r) : 0u;
        const uint32_t incl = wave_scan_incl(cnt);
        if (e) {
            uint16_t* o = recs + base + (incl - cnt);
            if (val == 0) {
                for (uint32_t j = 0; j < q; ++j) *o++ = (uint16_t)(18u | (138u << 8));
                if (q) atomicAdd(&metaF[18], q);
                if (r >= 3u) { const uint32_t sym = r < 11u ? 17u : 18u; *o++ = (uint16_t)(sym | (r << 8)); atomicAdd(&metaF[sym], 1u); }
                else { for (uint32_t j = 0; j < r; ++j) *o++ = 0; if (r) atomicAdd(&metaF[0], r); }
            } else {
                *o++ = (uint16_t)val;
                for (uint32_t j = 0; j < q; ++j) *o++ = (uint16_t)(16u | (6u << 8));
                if (r >= 3u) *o++ = (uint16_t)(16u | (r << 8));
                else for (uint32_t j = 0; j < r; ++j) *o++ = (uint16_t)val;
                atomicAdd(&metaF[val], 1u + (r >= 3u ? 0u : r));
                if (q + (r >= 3u ? 1u : 0u)) atomicAdd(&metaF[16], q + (r >= 3u ? 1u : 0u));
            }
        }
        base += readlane(incl, 63);
        if (stm) carry = 64u * k + 63u - (uint32_t)__builtin_clzll(stm);
    }
    ZZ_WAVE_SYNC();
    return (int)base;
}

// backward twin of wave_extend_match: number of equal bytes going down from src[a-1] / src[b-1], at most
// maxlen (> 8; the first 8 are known equal). 4 bytes per lane.
__device__ __forceinline__ uint32_t wave_extend_back(const uint8_t* src, int64_t a, int64_t b, uint32_t maxlen)
{
    const uint32_t o = 8 + 4 * (uint32_t)lane_id();          // bytes a-o-4 .. a-o-1
    uint32_t d = 0;
    const bool act = o < maxlen;
    if (act) {
        // the caller guarantees b - maxlen >= start of readable memory; a 4-byte load may reach up to 3 bytes
        // below a-maxlen, which is still readable unless it crosses the buffer start: assemble byte-wise there
        if (o + 4 <= maxlen) d = load32(src + a - o - 4) ^ load32(src + b - o - 4);
        else {
            for (uint32_t i = 0; i < 4; ++i)
                if (o + i < maxlen) d |= (uint32_t)(src[a - o - 1 - i] ^ src[b - o - 1 - i]) << (8 * (3 - i));
        }
    }
    const uint64_t neq = ballot(act && d != 0);
    if (!neq) return maxlen;
    const int k = __builtin_ctzll(neq);
    const uint32_t dk = readlane(d, k);
    const uint32_t len = 8 + 4 * (uint32_t)k + ((uint32_t)__builtin_clz(dk) >> 3);   // top byte = nearest
    return len < maxlen ? len : maxlen;
}

// One hop of the level-2 walk: a match at candidate lane e if it is "plain" (strong, lengths exact), then e = the
// first candidate at or after the next probe position. winfo: fwd8 [4:0], need [7:5], "8 or more backward possible"
// bit 8, "16 or more forward" bit 9, plain bit 10, strong bit 11, next candidate [21:16] (0 = none: lane 0 never
// follows a match), lane + fwd8 [29:23]. The CU's one scalar unit is shared by all its wavefronts (tools/ubench_valu.hip),
// so every scalar instruction saved here is saved for the whole CU: s_bfe sets SCC itself, no compare.
#define ZZ_L2_HOP \
                        "v_readlane_b32 %[inf], %[winfo], %[e]\n\t" \
                        "s_bitcmp1_b32 %[inf], 10\n\t" \
                        "s_cbranch_scc0 5f\n\t"                     /* weak or flagged */ \
                        "s_bitset1_b64 %[ev], %[e]\n\t"             /* a match is found at this probe (:406-407) */ \
                        "s_bfe_u32 %[Brel], %[inf], 0x70017\n\t"    /* backRefEnd (:422), relative to the block */ \
                        "s_bfe_u32 %[e], %[inf], 0x60010\n\t"       /* hop: the first candidate at or after j = backRefEnd + 1 (:424); 0 = none, and SCC says so */

// ---- token pass ------------------------------------------------------------------------------------------------
// FirstPass + AddHashEntries over the whole packet (encoder.cpp:217-248, 375-440, 474-480). Returns the number
// of matches written to `tokens` (ascending start) and, through nrec_out, the number of records in `recs`.
// Hand-over from the parsing wavefront to the helper wavefront, one slot per 64-position block, two slots:
// word l = lane l's match (ZZ_L2_HB_*), words 64,65 = mask of lanes that found one. The helper picks a slot up behind
// barrier i; the parser overwrites it after barrier i + 1.
// (A lane the C++ path finished carries start, length, distance: ZZ_L2_HB_PACK; a lane the scalar loop only marked
// carries its forward length, its usable backward length and the distance -- the helper works out where the match
// starts: words 66,67 = mask of the finished lanes, word 68 = backRefEnd when the block was entered.)
#define ZZ_L2_HB_WORDS 70u
#define ZZ_L2_HB_PACK(ms, mlen, dist, base) (((ms) + 258u - (base)) | (((mlen) - 3u) << 9) | ((dist) << 17))
#define ZZ_L2_HB_PACK_FAST(fwd, broom, dist) ((fwd) | ((broom) << 5) | ((dist) << 17))
__device__ __forceinline__ void l2_block_barrier()
{
    // this wave's LDS traffic must have landed; its global loads (prefetches) stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ uint32_t l2_probe_blocks(uint32_t n)     // trips of the token pass = hand-over barriers
{
    const uint32_t target = n > ZZ_MAX_LEN ? n - ZZ_MAX_LEN : 0;
    return (target + 63) >> 6;
}

// BIAS = 32768: warm window (the extended levels 4..6, see zz_api.hip): table entries are position + 1 + BIAS, the
// positions -32768 .. -1 in front of the packet were hashed into the table by the caller; candidates 32768 or more back
// are ignored (encoder.cpp:392).
// (ffbh_or_ones, zz_level1.h: "-1 for 0" as ffbl_or_ones has it, here for equal TRAILING bytes, the bytes in front of a position)
__device__ __forceinline__ uint32_t sub_from4_sat(uint32_t v)           // max(4 - v, 0)
{
    uint32_t r;
    asm("v_sub_u32_e64 %0, 4, %1 clamp" : "=v"(r) : "v"(v));
    return r;
}

// TS: the same bytes as `src`, with loads that would leave the shard turned to the copy of its end (l1p_src, zz_level1p.h):
// one instance of this function instead of a second, bounds-checked one -- that one's byte-wise loads cost the kernel 18
// spilled VGPRs (76 bytes of scratch per lane until round 4).
template <uint32_t BIAS = 0>
__device__ __forceinline__ void l2_token_pass(uint16_t* T, uint32_t* hb, const uint8_t* src, const uint8_t* end, const l1p_src& TS,
                                              uint32_t n, uint64_t before, unsigned long long* prof = nullptr)
{
    const int lane = lane_id();
    const uint64_t below_me = (1ull << lane) - 1, above_me = ~((2ull << lane) - 1);
    ZZ_PROF_DECL
    const uint32_t target = n > ZZ_MAX_LEN ? n - ZZ_MAX_LEN : 0;     // :222 last 258 bytes never searched
    uint32_t B = 1;                 // backRefEnd (:380)
    uint32_t nextProbe = 1;         // j (:383)
    uint32_t batchEnd = target < ZZ_BATCH_LEN ? target : ZZ_BATCH_LEN;
    uint32_t skipPos = 0;           // position that must not be inserted in the current block (0 = byte 0)
    // bytes in front of the packet, as far as they matter: a candidate at offset c has min(before + c, 258) bytes of room
    // behind it (D4 + D11 caps), and offsets below lo8 have fewer than the 8 bytes the backward compare loads
    const uint32_t bcap = (uint32_t)(before < ZZ_MAX_LEN ? before : ZZ_MAX_LEN);
    const uint32_t lo8 = before >= 8 ? 0u : 8u - (uint32_t)before;
    const uint8_t* const srcm8 = src - 8;
    // this lane's own bytes: 16 from its position (hash + forward compare) and the 8 in front (backward compare)
    uint64_t wa = 0, wa2 = 0, wb = 0;
    if ((uint32_t)lane < n) {
        l1p_ld128<true>(TS, lane, wa, wa2);
        if (before + (uint32_t)lane >= 8) wb = load64(src + (int64_t)lane - 8);
    }
    uint32_t slotsel = 0;           // word offset of the hand-over slot in use (alternates)
    // One block of 64 positions. Every position of a block lies below target = n - 258, so its own 16-byte loads, the
    // next block's and every candidate's stay inside the packet: no bounds checks and no clamps anywhere here.
    // INTERIOR: not the packet's first block, no position excluded from the table, the whole block inside the current
    // batch and some position of it probed -- the common case; the lane conditions and their scalar bookkeeping (every
    // lane-mask region is three scalar instructions, and the CU's one scalar unit is what this kernel is short of)
    // drop out of that copy of the code.
    auto block = [&](auto interior_tag, const uint32_t base) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
        ZZ_T(6);
        const uint32_t q = base + lane;
        const bool ins = INTERIOR ? true : (q != skipPos && q != 0);
        const bool doProbe = INTERIOR ? true : (base + 64 > nextProbe && nextProbe < batchEnd);   // some position of this block is probed
        const uint32_t h = calc_hash3((uint32_t)wa);                  // CalcHash(source + j), :388
        // lanes that must not insert (byte 0, a batch's first byte) read and write a slot of their own behind the kernel's
        // other LDS data instead: no lane mask around the table accesses
        const uint32_t hs = ins ? h : (uint32_t)(ZZ_L2_LDS_BYTES / 2);
        uint32_t old = T[hs];                                         // :389
        T[hs] = (uint16_t)(q + 1 + BIAS);                             // :390 / :474-480
        if (!ins) old = 0;
        if (BIAS && q + 1 + BIAS - old >= 32768u) old = 0;            // :392 (inside a cold packet every candidate is in reach)
        // the table candidate's bytes are requested at once (a lane whose candidate turns out to sit in this very
        // block takes that lane's registers instead): 16 at the candidate (:399), 8 in front of it (:92-102).
        // Every lane loads (a lane without a candidate: the packet's first bytes, result not looked at).
        uint64_t ca = 0, ca2 = 0, cpre = 0;
        if (doProbe) {
            if (BIAS == 0) {
                const uint32_t c0 = __builtin_elementwise_sub_sat(old, 1u);
                ld128<false>(src + c0, end, ca, ca2);
                cpre = load64(srcm8 + (c0 > lo8 ? c0 : lo8));          // (too close to the stream's start: fixed up below)
            } else {
                const int32_t c0 = old ? (int32_t)(old - 1 - BIAS) : 0;
                ld128<false>(src + c0, end, ca, ca2);
                cpre = load64(src + ((int64_t)before + c0 >= 8 ? (int64_t)c0 - 8 : 0));
            }
        }
        // next block's own bytes: in flight during the rest of this block
        uint64_t wan, wan2, wbn;
        ld128<false>(src + q + 64, end, wan, wan2);
        wbn = load64(src + q + 56);
        if (INTERIOR || base) l2_block_barrier();   // releases the previous block's matches to the helper (the table read above had to land anyway)
        ZZ_WAVE_SYNC();
        // Positions of this block that share a hash: the read-back names the lane whose store landed, the same
        // lane for every member of a set and a different one for different sets -- a 6-bit key. Six ballots give
        // every lane the mask of its set (no loop over sets); the candidate of a later member is the nearest
        // earlier member, and the table must end up holding the last member.
        const uint32_t rb = T[hs];
        uint32_t cand1 = old;                                         // candidate as pos+1, 0 = none
        bool inl = false;                                             // the candidate is lane `il` of this block
        uint32_t il = 0;
        const uint64_t lost = INTERIOR ? ballot(rb != q + 1 + BIAS) : ballot(ins && rb != q + 1 + BIAS);
        if (lost) {
            const uint32_t W = ins ? (rb - 1u - BIAS - base) & 63u : (uint32_t)lane;
            const uint64_t set = wave_match6(W);
            const uint64_t below = set & below_me;
            inl = ins && below != 0;
            il = 63u - (uint32_t)__builtin_clzll(below | 1ull);       // nearest earlier member (lane 0 where there is none: unused)
            cand1 = inl ? base + il + 1 + BIAS : old;
            ZZ_WAVE_SYNC();
            // last member wins: the highest lane of a set rewrites the slot unless its own store was the one that landed
            if (INTERIOR) {
                const uint64_t fix = ballot(W != (uint32_t)lane) & ballot((set & above_me) == 0);
                uint64_t saved;
                asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write_b16 %2, %3\n\ts_mov_b64 exec, %0"
                             : "=&s"(saved) : "s"(fix), "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)(T + h)), "v"(q + 1 + BIAS)
                             : "memory", "scc");
            } else if (ins && W != (uint32_t)lane && (set & above_me) == 0) T[h] = (uint16_t)(q + 1 + BIAS);
        }
        ZZ_WAVE_SYNC();
        skipPos = 0xFFFFFFFFu;   // only the block that contains it skips (byte 0 is excluded by q != 0)

        ZZ_T(7);
        uint32_t* slot = hb + slotsel;
        slotsel ^= ZZ_L2_HB_WORDS;
        if (doProbe) {
            // ---- quick compare info for all 64 probes of this block --------------------------------------
            const bool has = INTERIOR ? cand1 != 0 : (ins && cand1 != 0 && q < batchEnd);
            const int32_t c = (int32_t)(cand1 - 1 - BIAS);           // may lie in front of the packet (warm window)
            if (lost) {
                // candidates inside the block: their bytes come from the owning lane's registers
                const int qa = (int)(il << 2);
                const uint64_t sa = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)(wa >> 32)) << 32) | (uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)wa);
                const uint64_t sa2 = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)(wa2 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)wa2);
                const uint64_t sp = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)(wb >> 32)) << 32) | (uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)wb);
                if (inl) { ca = sa; ca2 = sa2; cpre = sp; }
            }
            uint32_t fwd8, bwd8, room;
            {
                if (BIAS == 0) {
                    const uint32_t r = bcap + (uint32_t)c;                 // bytes in front of the candidate, as far as they count
                    room = r < ZZ_MAX_LEN ? r : ZZ_MAX_LEN;                // D4 + D11 caps
                } else {
                    const uint64_t cb = (uint64_t)((int64_t)before + c);
                    room = cb < ZZ_MAX_LEN ? (uint32_t)cb : ZZ_MAX_LEN;
                }
                // equal bytes in front of the probe and in front of the candidate: leading zero bits of the XOR of the two
                // words that end there, 8 = "8 or more" (ffbh gives -1 for 0, the addition saturates, min3 caps)
                const uint64_t y = wb ^ cpre;
                bwd8 = umin3(ffbh_or_ones((uint32_t)(y >> 32)), add_sat_k<32>(ffbh_or_ones((uint32_t)y)), 64u) >> 3;
                // fewer than 8 bytes in front of the candidate -- only in the first bytes of a stream: byte by byte. (A cold packet's
                // candidates lie inside it, so a packet with 8 bytes in front of it never gets here; a warm window reaches back.)
                if ((BIAS != 0 || before < 8) && ballot(has && room < 8)) {
                    if (has && room < 8) {
                        bwd8 = 0;
                        while (bwd8 < room && src[(int64_t)q - 1 - bwd8] == src[(int64_t)c - 1 - bwd8]) bwd8++;
                    }
                }
                fwd8 = equal_bits128(wa ^ ca, wa2 ^ ca2, 128u) >> 3;   // 16 bytes at the probe and at the candidate (:399); 16 = "16 or more"
                if (!has) fwd8 = 0;
            }
            uint32_t broom = bwd8 < room ? bwd8 : room;
            if (!has) broom = 0;
            ZZ_DRAIN();
            ZZ_T(8);
            // ---- the greedy walk ----------------------------------------------------------------------------------
            // A probe at lane l is a match when fwd8 + min(l - backRefEnd, broom) >= 4 (:404-407). "Strong" lanes
            // (fwd8 >= 4) are matches whatever the parse did before; "weak" ones need `need` pending literals.
            // After a match backRefEnd = probe + fwd8 whatever the backward part was (:416,422), so the scalar loop
            // only hops from match to match (scalar code is slow here: tools/ubench_scalar.hip) and marks the lanes;
            // starts and lengths are computed for all marked lanes at once afterwards. Lengths of "16 or more"
            // forward / "8 or more" backward drop out to the C++ below.
            const uint64_t Amask = ballot(fwd8 + broom >= 4);                     // strong or weak
            const uint32_t inexact = (fwd8 & 16u) | (broom & 8u);                // "16 or more" forward (-> bit 9) | "8 or more" backward (-> bit 8)
            uint32_t winfo = fwd8 | (sub_from4_sat(fwd8) << 5) | (inexact << 5) | (((fwd8 + 28u) & 32u) << 6) |   // fwd8 >= 4: strong, bit 11
                             ((fwd8 >= 4 && inexact == 0) ? 0x400u : 0u) | (((uint32_t)lane + fwd8) << 23);
            {
                const uint32_t endl = (uint32_t)lane + fwd8 + 1;                  // first lane probed after a match here
                const uint64_t m = Amask >> (endl & 63u);                         // (endl >= 64: whatever this finds lies at 64 or beyond)
                const uint32_t nx = endl + (m ? (uint32_t)__builtin_ctzll(m) : 64u);
                winfo |= ((nx < 64u ? nx : 64u) & 63u) << 16;                     // (64 & 63 = 0 = none)
            }
            uint32_t tk = 0;                // start | len << 16, slow tokens only; the others are filled in below
            uint64_t evmask = 0, slowmask = 0;
            const uint32_t Bentry = B;
            // positions relative to the block while the walk runs: np = first lane that may be probed (< 64 on entry: some
            // position of this block is probed), Brel = backRefEnd
            uint32_t np = (int32_t)(nextProbe - base) > 0 ? nextProbe - base : 0u;
            int32_t Brel = (int32_t)(B - base);
            // (the scalar loop once in front and once at the bottom of the loop over what it cannot decide: the common case -- no
            // such token in the block -- leaves through one test, where a loop around a single instance paid seven scalar
            // instructions of flag bookkeeping behind every block's walk)
            uint32_t slow = 0;
            auto walk = [&]() {
                    uint32_t inf, t1, t2;
                    uint64_t tmp;
                    int32_t e;
                    asm volatile(
                        "1:\n\t"
                        "s_lshl_b64 %[tmp], -1, %[np]\n\t"
                        "s_and_b64 %[tmp], %[tmp], %[A]\n\t"        // candidates at or after np; SCC = there is one
                        "s_cbranch_scc0 3f\n\t"
                        "s_ff1_i32_b64 %[e], %[tmp]\n"
                        // plain lanes: two hops per taken branch (a taken branch costs about five scalar instructions)
                        "9:\n\t"
                        ZZ_L2_HOP "s_cbranch_scc0 10f\n\t"
                        ZZ_L2_HOP "s_cbranch_scc1 9b\n"
                        "10:\n\t"
                        "s_add_u32 %[np], %[Brel], 1\n\t"           // no candidate left: j = backRefEnd + 1 (:424)
                        "s_branch 3f\n"
                        "5:\n\t"
                        "s_bitcmp1_b32 %[inf], 11\n\t"
                        "s_cbranch_scc1 4f\n\t"                     // strong but flagged
                        "s_sub_i32 %[t1], %[e], %[Brel]\n\t"        // weak: pending literals j - backRefEnd (:404)
                        "s_bfe_u32 %[t2], %[inf], 0x30005\n\t"
                        "s_cmp_ge_i32 %[t1], %[t2]\n\t"
                        "s_cbranch_scc1 7f\n\t"
                        "s_add_u32 %[np], %[e], 1\n\t"              // no match at this probe: j++ (:430)
                        "s_cmp_lt_u32 %[np], 64\n\t"
                        "s_cbranch_scc1 1b\n\t"
                        "s_branch 3f\n"
                        "7:\n\t"
                        "s_and_b32 %[t1], %[inf], 0x300\n\t"        // "8 or more" backward possible | "16 or more" forward
                        "s_cmp_eq_u32 %[t1], 0\n\t"
                        "s_cbranch_scc0 4f\n"
                        "8:\n\t"
                        "s_bitset1_b64 %[ev], %[e]\n\t"             // a match is found at this probe (:406-407)
                        "s_bfe_u32 %[Brel], %[inf], 0x70017\n\t"
                        "s_bfe_u32 %[e], %[inf], 0x60010\n\t"
                        "s_cbranch_scc1 9b\n\t"
                        "s_branch 10b\n"
                        "4:\n\t"
                        "s_bitcmp1_b32 %[inf], 9\n\t"
                        "s_cbranch_scc1 2f\n\t"                     // forward "16 or more": extend in C++
                        "s_sub_i32 %[t1], %[e], %[Brel]\n\t"
                        "s_cmp_ge_i32 %[t1], 8\n\t"
                        "s_cbranch_scc0 8b\n"                        // fewer than 8 pending literals: the backward part is exact
                        "2:\n\t"
                        "s_mov_b32 %[np], %[e]\n\t"
                        "s_mov_b32 %[slow], 1\n"
                        "3:\n\t"
                        : [Brel] "+s"(Brel), [np] "+s"(np), [ev] "+s"(evmask), [slow] "+s"(slow), [inf] "=&s"(inf),
                          [t1] "=&s"(t1), [t2] "=&s"(t2), [tmp] "=&s"(tmp), [e] "=&s"(e)
                        : [winfo] "v"(winfo), [A] "s"(Amask)
                        : "scc");
            };
            walk();
            while (__builtin_expect(slow != 0, 0)) {
                // one token with a length of "8 or more" backward or "16 or more" forward, at lane np
                const int e = (int)np;
                const uint32_t qe = base + (uint32_t)e;
                uint32_t fwd = readlane(fwd8, e);
                const uint32_t pe = (uint32_t)(e - Brel);                // j - backRefEnd (:404)
                const uint32_t bre = readlane(broom, e);
                uint32_t bw = bre < pe ? bre : pe;
                {
                    const int32_t ce = (int32_t)readlane((uint32_t)c, e);
                    if (fwd == 16) fwd = l1p_extend_match(TS, qe, ce, ZZ_MAX_LEN, 16);    // remain(), :64-90
                    const uint32_t re = readlane(room, e);
                    const uint32_t blim = re < pe ? re : pe;
                    if (bw == 8 && blim > 8) bw = wave_extend_back(src, qe, ce, blim);            // :92-102
                }
                uint32_t mlen = fwd + bw;
                if (mlen > ZZ_MAX_LEN) mlen = ZZ_MAX_LEN;                                           // :412-415
                const uint32_t ms = qe - bw;                                                        // :416
                Brel = (int32_t)(ms + mlen - base);                                                 // :422
                if (lane == e) tk = ms | (mlen << 16);                                              // :420
                evmask |= 1ull << e;
                slowmask |= 1ull << e;
                np = (uint32_t)Brel + 1;                                                            // :424
                slow = 0;
                if (np < 64) walk();
            }
            B = base + (uint32_t)Brel;
            nextProbe = base + np;
            ZZ_T(9);
            // ---- hand this block's matches to the helper wavefront (which also fills in the starts and lengths of the
            // matches the scalar loop only marked). Every lane stores (a lane without a match: a word nobody looks at; the
            // masks and backRefEnd: the same words from all lanes) -- no lane mask to set up.
            const uint32_t dist = (uint32_t)((int32_t)q - c);
            slot[lane] = sel_lanes(slowmask, ZZ_L2_HB_PACK(tk & 0xFFFF, tk >> 16, dist, base), ZZ_L2_HB_PACK_FAST(fwd8, broom, dist));
            slot[64] = (uint32_t)evmask; slot[65] = (uint32_t)(evmask >> 32);
            slot[66] = (uint32_t)slowmask; slot[67] = (uint32_t)(slowmask >> 32);
            slot[68] = Bentry;
        } else {
            slot[64] = 0; slot[65] = 0;
        }
        ZZ_T(12);
        wa = wan; wa2 = wan2; wb = wbn;
    };
    for (uint32_t base = 0; base < target; base += 64) {
        if (base >= batchEnd) {
            // batch switch (:228-230, :435-438): the next batch starts at max(backRefEnd, end); its first
            // byte is inserted only if the last match covered it
            const uint32_t s2 = B > batchEnd ? B : batchEnd;
            skipPos = B >= batchEnd ? 0xFFFFFFFFu : batchEnd;
            B = s2 + 1;
            nextProbe = s2 + 1;
            // (a match that overruns the batch AND the search region ends the pass: encoder.cpp:225-234 leaves its loop with
            // target <= 0 -- found by tools/fuzz_gpu.py, seed 1234: period-375 data of 16,866 bytes got one match too many)
            const uint32_t rest = target > s2 ? target - s2 : 0u;
            batchEnd = s2 + (rest < ZZ_BATCH_LEN ? rest : ZZ_BATCH_LEN);
        }
        if (base >= 64 && skipPos == 0xFFFFFFFFu && base + 64 <= batchEnd && base + 64 > nextProbe) block(std::true_type{}, base);
        else block(std::false_type{}, base);
    }
    if (target) l2_block_barrier();       // the last block's matches
#ifdef ZZ_PROF
    if (lane == 0 && prof) { for (int _i = 6; _i < 10; ++_i) atomicAdd(&prof[_i], prof_acc[_i]); atomicAdd(&prof[12], prof_acc[12]); }
#endif
}

// The helper wavefront's side of the token pass: per block, publish the parser's matches (scratch, bitmap window,
// symbol counts) and finish the block that can no longer change. Returns the number of matches; the number of
// records through nrec_out.
__device__ __forceinline__ uint32_t l2_helper_pass(const uint32_t* hb, uint64_t* covw, uint64_t* mstw, uint32_t* histP,
                                                   uint32_t* tokens, uint16_t* recs, const uint8_t* src, uint32_t n,
                                                   uint32_t& nrec_out, uint32_t& adA, uint64_t& adC, const uint32_t trips)
{
    const int lane = lane_id();
    uint32_t nrec = 0, Fnext = 0, ntok = 0;
    // Adler-32 on the way: every byte below n passes through this wave once (as the byte of a finished block), so the
    // per-lane sums of d and position * d (wave_adler's A and C) cost two instructions here instead of a pass of their own
    adA = 0; adC = 0;
    for (uint32_t i = 0; i < trips; ++i) {
        const uint32_t base = i << 6;
        // the block that becomes final in this trip: its bytes are fetched before the wait
        uint32_t fbyte = 0;
        if (i >= ZZ_L2_LAG) {
            const uint32_t p = base + lane - 64 * ZZ_L2_LAG;        // < n: the probe front is at least 258 bytes from the end
            fbyte = src[p];
            adA += fbyte; adC += (uint64_t)p * fbyte;
        }
        l2_block_barrier();
        const uint32_t* slot = hb + (i & 1) * ZZ_L2_HB_WORDS;
        const uint32_t pk = slot[lane];
        const uint64_t evmask = ((uint64_t)uniform(slot[65]) << 32) | uniform(slot[64]);
        // This wavefront's work grows with the block's matches, the parser's hardly: an instrumented build shows it BUSY for
        // 1.66 M of the token pass's 1.83 M cycles per packet of text -- at issue priority 0 against the parser's 3 it was the one
        // the packet waited for. Where a block's matches are dense it takes the parser's priority for that block (text: 71.6 ->
        // 74.0 GB/s; always equal priorities: 73.9, but 69.1 against 70.3 on the twelve-family mix, where this costs 1 %).
        // (a one-armed test: with a priority in either arm the compiler's control-flow bookkeeping was eight instructions)
        const uint32_t nmatch = uniform((uint32_t)__builtin_popcountll(evmask));
        __builtin_amdgcn_s_setprio(0);
        if (nmatch >= ZZ_L2_HELPER_DYNPRIO) __builtin_amdgcn_s_setprio(3);
        if (evmask) {
            const uint64_t slowmask = ((uint64_t)uniform(slot[67]) << 32) | uniform(slot[66]);
            const uint32_t Bentry = uniform(slot[68]);
            const uint32_t q = base + (uint32_t)lane;
            const bool slow = (slowmask >> lane) & 1;
            uint32_t ms = base + (pk & 0x1FF) - 258u, mlen = ((pk >> 9) & 0xFF) + 3u;       // as the C++ path left them
            // matches the scalar loop only marked: backRefEnd after a match is probe + forward length (:416,422); the
            // backward part is limited by the literals pending since the previous match of this block (or since the
            // block was entered, :404-407)
            const uint32_t fwd8 = pk & 31, broom = (pk >> 5) & 15;
            const uint32_t endp = slow ? ms + mlen : q + fwd8;
            const uint64_t prev = evmask & ((1ull << lane) - 1);
            const int pl = prev ? 63 - __builtin_clzll(prev) : lane;
            const uint32_t pend_end = (uint32_t)__shfl((int)endp, pl);            // every lane takes part
            if (!slow) {
                const uint32_t pe = q - (prev ? pend_end : Bentry);
                const uint32_t bq = broom < pe ? broom : pe;
                ms = q - bq; mlen = fwd8 + bq;
            }
            if ((evmask >> lane) & 1) {
                const uint32_t dist = pk >> 17;
                uint32_t sym, leb, lev, bucket, deb, dev;            // GetFrequencies, :455-463
                length_symbol(mlen, sym, leb, lev);
                dist_symbol(dist, bucket, deb, dev);
                // the match as the emission pass wants it: symbols and extra-bit values
                tokens[ntok + mbcnt(evmask)] = ((sym - 257) << 23) | (lev << 18) | (bucket << 13) | dev;
                // covered / start bits: words (base>>6)-5 .. (base>>6)+5 of the LDS window
                // (straight line for the one or two words nearly every match touches; a loop only for matches of more than 64 bytes)
                const uint32_t last = ms + mlen - 1;
                const uint32_t w0 = ms >> 6, w1 = last >> 6;
                const uint64_t from_lo = ~0ull << (ms & 63), to_hi = ~0ull >> (63u - (last & 63));
                atomicOr((unsigned long long*)&covw[w0 & (ZZ_L2_WIN - 1)], (unsigned long long)(w1 == w0 ? (from_lo & to_hi) : from_lo));
                if (w1 != w0) {
                    atomicOr((unsigned long long*)&covw[w1 & (ZZ_L2_WIN - 1)], (unsigned long long)to_hi);
                    for (uint32_t wi = w0 + 1; wi < w1; ++wi) atomicOr((unsigned long long*)&covw[wi & (ZZ_L2_WIN - 1)], ~0ull);
                }
                atomicOr((unsigned long long*)&mstw[(ms >> 6) & (ZZ_L2_WIN - 1)], 1ull << (ms & 63));
                hist_add(histP, sym);
                hist_add(histP, 286 + bucket);
            }
            ntok += nmatch;
        }
        // later matches start at >= base + 64 - 258: block (base>>6) - 5 cannot change any more
        if (i >= ZZ_L2_LAG) {
            ZZ_WAVE_SYNC();
            nrec = l2_finish_block(covw, mstw, histP, recs, nrec, Fnext, n, fbyte);
            Fnext++;
        }
    }
    ZZ_WAVE_SYNC();
    for (const uint32_t nblk = (n + 63) >> 6; Fnext < nblk; ++Fnext) {       // the tail nobody probes (:222) + the lag
        const uint32_t p = (Fnext << 6) + (uint32_t)lane;
        const uint32_t d = p < n ? src[p] : 0u;
        adA += d; adC += (uint64_t)p * d;
        nrec = l2_finish_block(covw, mstw, histP, recs, nrec, Fnext, n, d);
    }
    nrec_out = nrec;
    return ntok;
}

// ---- block body: WriteRecords (encoder.cpp:149-169) over the dense records [r_begin, r_end), 64 per trip --------
// Loads run ahead of their use: records by two trips, the matches a trip needs (a gather: their index depends on
// the records) by one. `mc` = number of matches in front of r_begin.
__device__ __forceinline__ void l2_emit_records(bitring& ring, const uint16_t* recs, const uint32_t* tokens,
                                                const uint32_t* codes, const uint32_t* dcodes, uint32_t r_begin,
                                                uint32_t r_end, uint32_t mc)
{
    const uint32_t lane = (uint32_t)lane_id();
    uint32_t v = r_begin + lane < r_end ? recs[r_begin + lane] : ZZ_L2_REC_NONE;
    uint32_t vn = r_begin + 64 + lane < r_end ? recs[r_begin + 64 + lane] : ZZ_L2_REC_NONE;
    uint64_t mb = ballot(v == ZZ_L2_REC_MATCH);
    uint32_t t = 0;
    if (v == ZZ_L2_REC_MATCH) t = tokens[mc + mbcnt(mb)];
    for (uint32_t r0 = r_begin; r0 < r_end; r0 += 64) {
        const uint32_t rnn = r0 + 128 + lane;
        const uint32_t vnn = rnn < r_end ? recs[rnn] : ZZ_L2_REC_NONE;
        mc += (uint32_t)__builtin_popcountll(mb);
        const uint64_t mbn = ballot(vn == ZZ_L2_REC_MATCH);
        uint32_t tn = 0;
        if (vn == ZZ_L2_REC_MATCH) tn = tokens[mc + mbcnt(mbn)];
        // One straight line for literals, matches and the padding of the last trip (no lane-mask regions: with both kinds in
        // nearly every trip both sides ran anyway, plus the scalar bookkeeping of two regions). A literal is a match whose
        // token is 0: length symbol 0 -> no extra bits, distance part switched off.
        const bool ism = v == ZZ_L2_REC_MATCH;
        const uint32_t ls = t >> 23, lev = (t >> 18) & 31, bucket = (t >> 13) & 31, dev = t & 0x1FFF;   // (t = 0 unless a match)
        const uint32_t leb = (ls < 8 || ls == 28) ? 0 : (ls - 4) >> 2;          // luts.cpp:64
        const uint32_t deb = bucket < 4 ? 0 : (bucket - 2) >> 1;
        uint32_t lc = codes[ism ? 257 + ls : (v & 0x1FF)];
        if (v > ZZ_L2_REC_MATCH) lc = 0;                                        // ZZ_L2_REC_NONE: nothing
        uint32_t dc = dcodes[bucket];
        if (!ism) dc = 0;
        uint32_t ln = lc >> 16;
        uint64_t w = (lc & 0xFFFF) | ((uint64_t)lev << ln);                     // Merge, :121-124
        ln += leb;
        w |= (uint64_t)(dc & 0xFFFF) << ln;                                     // WriteDistance, :135-141
        ln += dc >> 16;
        w |= (uint64_t)dev << ln;
        ln += deb;
        ring_append64(ring, w, ln);
        v = vn; vn = vnn; t = tn; mb = mbn;
    }
}
// the same walk without output: bits the records [0, r_end) will take, and how many of them are matches
__device__ __forceinline__ uint32_t l2_count_bits(const uint16_t* recs, const uint32_t* tokens, const uint32_t* codes,
                                                  const uint32_t* dcodes, uint32_t r_end, uint32_t& matches)
{
    const uint32_t lane = (uint32_t)lane_id();
    uint32_t mc = 0, acc = 0;
    uint32_t v = lane < r_end ? recs[lane] : ZZ_L2_REC_NONE;
    uint32_t vn = 64 + lane < r_end ? recs[64 + lane] : ZZ_L2_REC_NONE;
    uint64_t mb = ballot(v == ZZ_L2_REC_MATCH);
    uint32_t t = 0;
    if (v == ZZ_L2_REC_MATCH) t = tokens[mbcnt(mb)];
    for (uint32_t r0 = 0; r0 < r_end; r0 += 64) {
        const uint32_t rnn = r0 + 128 + lane;
        const uint32_t vnn = rnn < r_end ? recs[rnn] : ZZ_L2_REC_NONE;
        mc += (uint32_t)__builtin_popcountll(mb);
        const uint64_t mbn = ballot(vn == ZZ_L2_REC_MATCH);
        uint32_t tn = 0;
        if (vn == ZZ_L2_REC_MATCH) tn = tokens[mc + mbcnt(mbn)];
        {   // (the same straight line as in l2_emit_records)
            const bool ism = v == ZZ_L2_REC_MATCH;
            const uint32_t ls = t >> 23, bucket = (t >> 13) & 31;
            uint32_t lc = codes[ism ? 257 + ls : (v & 0x1FF)];
            if (v > ZZ_L2_REC_MATCH) lc = 0;
            uint32_t dc = dcodes[bucket];
            if (!ism) dc = 0;
            acc += (lc >> 16) + ((ls < 8 || ls == 28) ? 0 : (ls - 4) >> 2) + (dc >> 16) + (bucket < 4 ? 0 : (bucket - 2) >> 1);
        }
        v = vn; vn = vnn; t = tn; mb = mbn;
    }
    matches = mc;
    return wave_sum(acc);
}

}  // namespace zz
#include "zz_level6.h"
#include "zz_level2p.h"
namespace zz {

struct zz_l2_params {
    zz_packet_params pk;
    uint8_t* scratch;      // gridDim.x * ZZ_L2_SCRATCH_BYTES
    uint32_t* work;        // packets handed out beyond the first gridDim.x (zero at launch)
    const uint32_t* m;     // extended levels: k_l6_matches' word per input byte of the packets k0 .. k1-1
    uint32_t k0, k1;       // the packets of this launch (levels 2,3: all of them)
};

#define ZZ_L2_THREADS (2 * ZZ_WAVE)
#ifndef ZZ_L2_SPLIT64
#define ZZ_L2_SPLIT64 39u      // 64ths of the records that wavefront 0 emits
#endif
// one wave's own memory traffic has landed (the code after the token pass runs on wavefront 0 alone: no s_barrier there)
#define ZZ_WAVE_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")

// Two wavefronts per packet. During the token pass wavefront 0 parses (hash table, candidates, the serial walk) and
// wavefront 1 keeps the books (matches to scratch, bitmap window, symbol counts, finished blocks to records, the
// Adler-32 sums), one s_barrier per 64-position block. Afterwards wavefront 0 builds the codes and both emit.
// BIAS = 32768: the same with a warm window (P.warm bytes in front of every packet are hashed into its table first).
// XD: the extended levels 4..6 (zz_level6.h) -- the parser walks the per-position matches that k_l6_matches left in Q.m instead
// of probing a table, and the code lengths come from package-merge; everything else is shared.
// XD needs no hash table: what sits behind it in LDS moves down to where the code-construction scratch and the helper's bit ring
// end, 13,760 bytes in all => eleven workgroups per CU instead of nine (22 wavefronts: six on two of the SIMDs => at most 80
// VGPRs): the encode kernel of level 6 12.1 -> 10.0 ms per GiB. (Twelve -- the helper's ring over the dead match-start bits
// and counters, 12,992 bytes -- ran no faster: 26.4 against 26.3 ms.)
// PP (levels 2,3 with cold packets: "k_encode_l2p"): THREE wavefronts -- the token pass on two parsers that take alternate
// blocks (zz_level2p.h), wavefront 1 stays the helper, wavefront 2 is the second parser and sits out the rest of the packet at
// the barriers. 27 wavefronts per CU: seven on one SIMD => at most 72 VGPRs.
template <uint32_t BIAS, bool XD = false, bool PP = false>
__global__ __launch_bounds__(PP ? ZZ_L2P_THREADS : ZZ_L2_THREADS, PP ? ZZ_L2P_WPE : (XD ? 6 : 5)) void k_encode_l2_t(zz_l2_params Q)
{
    static_assert(!PP || (BIAS == 0 && !XD), "the two-parser token pass exists for cold packets of levels 2,3");
    const zz_packet_params& P = Q.pk;
    // ---- LDS carve-up: 17,840 bytes => nine workgroups per CU (18 wavefronts: five per SIMD => at most 96 VGPRs).
    // The hash table is dead once the token pass is over, so the Huffman scratch, the 32-bit histograms and the
    // code tables all live inside it; the bit ring (used after the token pass) shares its space with the
    // hand-over slots (used during it); the bitmap window and the packed counters have their own.
    constexpr uint32_t HB = XD ? 12288u : 16384u;                 // where the part behind the hash table starts (XD: behind the helper's bit ring at 11520)
    constexpr uint32_t RING2 = 11520u;                            // the helper's bit ring of the emission
    constexpr uint32_t RING3 = 12288u;                            // PP: the second parser's (the dead hash table has room: 12080 .. 16384)
    __shared__ __attribute__((aligned(16))) uint8_t lds[HB + (ZZ_L2_LDS_BYTES - 16384) + 16];
    uint16_t* T = (uint16_t*)lds;                                 // 16384: hash table during the token pass
    uint32_t* symF = (uint32_t*)(lds + 8192);                     // 1280: 286 lit/len + pad | 30 dist at [288..318)
    uint32_t* distF = symF + 288;
    uint32_t* codes = (uint32_t*)(lds + 8192 + 1280);             // 1152: 286 lit/len codes
    uint32_t* dcodes = (uint32_t*)(lds + 8192 + 1280 + 1152);     // 128: 30 distance codes
    uint32_t* metaF = (uint32_t*)(lds + 8192 + 1280 + 1152 + 128);            // 80: 19 meta frequencies
    uint32_t* misc = (uint32_t*)(lds + 11264);                    // 256: lane-0 results [0..3], code-generation work area [16..48)
    uint32_t* ring_words = (uint32_t*)(lds + HB);                 // 512 (+48 pad)
    uint32_t* hb = (uint32_t*)(lds + HB);                         // 560: two hand-over slots, same bytes as the ring
    uint64_t* covw = (uint64_t*)(lds + HB + 560);                 // 128: covered bits of the 16 blocks around the probe front
    uint64_t* mstw = covw + ZZ_L2_WIN;                            // 128: match-start bits
    uint32_t* histP = (uint32_t*)(lds + HB + 560 + 2 * ZZ_L2_WIN * 8);        // 640: packed 16-bit counters
    __shared__ uint32_t xb[8];                                    // PP: backRefEnd, the next probe position and the hand-over counters (l2p_sync)
    // Huffman scratch inside the (dead) hash table
    huff_scratch S;
    S.rec_freq = (uint32_t*)(lds);                 // 1152
    S.rec_id = (uint16_t*)(lds + 1152);            // 576
    S.t_freq = (uint32_t*)(lds + 1728);            // 2304
    S.t_left = (uint16_t*)(lds + 4032);            // 1152
    S.t_right = (uint16_t*)(lds + 5184);           // 1152
    S.t_bits = (uint8_t*)(lds + 6336);             // 576
    uint8_t* lens = (uint8_t*)(lds + 6912);        // 320: lit/len lengths, then dist lengths at [288..318)
    uint8_t* metaLens = (uint8_t*)(lds + 7232);    // 32
    uint32_t* metaCodes = (uint32_t*)(lds + 7264); // 80
    uint16_t* rle = (uint16_t*)(lds + 7344);       // up to 316+30 records -> 704 bytes
    // XD: package-merge scratch in the same (dead) space, in front of `lens`
    pm_scratch PM;
    PM.sym = (uint16_t*)(lds);                     // 576
    PM.w = (uint32_t*)(lds + 576);                 // 1152
    PM.pk = (uint32_t*)(lds + 1728);               // 1152
    PM.cur = (uint32_t*)(lds + 2880);              // 2304
    PM.bm = (uint64_t*)(lds + 5184);               // 1152
    PM.misc = (uint32_t*)(lds + 6336);             // 128 (.. 6464 < 6912)

    const int lane = lane_id();
    const uint32_t wave = uniform(threadIdx.x >> 6);
    ZZ_PROF_DECL
    uint8_t* const my_scratch = Q.scratch + (uint64_t)blockIdx.x * ZZ_L2_SCRATCH_BYTES;
    uint32_t* tokens = (uint32_t*)my_scratch;                                   // matches in stream order
    uint16_t* recs = (uint16_t*)(my_scratch + ZZ_L2_MAX_TOKENS * 4);            // records in stream order
    uint32_t* snap = (uint32_t*)(my_scratch + ZZ_L2_MAX_TOKENS * 4 + ZZ_MAX_PACKET * 2);   // PP: the counters at the two places the body is cut (l2p_helper_pass)

    // Persistent workgroups: the first packet is the workgroup's index, every further one comes from a counter, so
    // that packets of unequal cost (stored fallback vs. dynamic block) do not leave workgroups idle at the end.
    auto next_packet = [&]() -> uint32_t {
        __syncthreads();                                   // both wavefronts are done with the packet (and with LDS)
        if (threadIdx.x == 0) ((uint32_t*)covw)[2] = Q.k0 + gridDim.x + atomicAdd(Q.work, 1u);
        __syncthreads();
        return uniform(((uint32_t*)covw)[2]);
    };
    // One packet, as seen by wavefront 0 (W0: parser, code construction, first part of the emission) or by wavefront 1
    // (helper). Each wavefront runs its OWN packet loop over this body (below): with one loop around both roles,
    // values that are invariant across packets are hoisted for both roles at once and live through each other's code,
    // and the kernel does not fit its 96 registers.
    auto packet = [&](const uint32_t k, auto roletag) {
        constexpr int ROLE = decltype(roletag)::value;       // 0: parser (+ codes, first part of the emission), 1: helper, 2 (PP): second parser
        constexpr bool W0 = ROLE == 0, PB = ROLE == 2;
        // Behind the token pass: CODER builds the codes and emits the first part of the body, MID the second (PP: PB the third).
        // Two wavefronts: the parser codes. PP: the HELPER codes and wavefront 0 takes the middle part -- a parser's token pass
        // wants every register it can get, and with the code construction in the same wavefront the compiler kept 31 packet-invariant
        // values in scratch and read eight of them back on every out-of-line token of the walk (parser 0's walk 2,541 cycles per
        // block of the mix against parser 1's 1,793: profiles/r05_phases_l2p_mix_first.txt); the helper's token-pass loop is small.
        constexpr bool CODER = (PP && ZZ_L2P_HELPER_CODES) ? ROLE == 1 : W0, MID = (PP && ZZ_L2P_HELPER_CODES) ? W0 : ROLE == 1;
        const uint64_t off = (uint64_t)k * P.packet_size;
        const uint32_t len = (uint32_t)((P.n - off) < P.packet_size ? (P.n - off) : P.packet_size);
        const bool is_final = P.last_is_final && k == P.npk - 1;
        const uint32_t n = is_final ? len : len - 1;     // bytes of the compressing AddData
        const uint8_t* src = P.src + off;
        const uint8_t* end = P.src + P.n;
        const uint64_t before = P.halo + off;            // input bytes in front of the packet (D4: stop at 0)
        uint8_t* out = P.slots + (uint64_t)k * P.slot_stride;

        __syncthreads();       // both wavefronts are done with the previous packet
        if (W0) {
            uint4* z = (uint4*)lds;
            if (!XD) for (int i = lane; i < 16384 / 16; i += ZZ_WAVE) z[i] = make_uint4(0, 0, 0, 0);   // T
        } else if (!PB) {
            uint32_t* zw = (uint32_t*)covw;
            for (int i = lane; i < 2 * ZZ_L2_WIN * 2 + ZZ_L2_HIST_WORDS; i += ZZ_WAVE) zw[i] = 0;   // window + counters
        }
        ZZ_WAVE_SYNC();
        if (PP && ZZ_L2P_FLAGS) {
            // the counters the wavefronts meet through during the token pass: cleared here, and one more barrier in front of their first
            // use (which also says "the table is clear" to the second parser and "window and counters are" to both)
            if (PB) { if (lane < 8) xb[lane] = lane < 2 ? 1u : 0u; }        // backRefEnd = 1 (:380), j = 1 (:383), nothing walked / entered / handed over
            __syncthreads();
        }
        ZZ_T(0);

        if (n > 0) {
            // ================= token pass (encoder.cpp:217-248, 375-471) ===========================================
            if (XD) {
                // the extended levels: every position's match is in Q.m already (k_l6_matches, zz_level6.h)
                if (W0) {
                    const uint32_t* mrow = Q.m + (uint64_t)(k - Q.k0) * P.packet_size;
                    l6_parse_pass(hb, src, end, l1p_make_src(P, src, end), n, mrow);
                }
            } else if (PP) {
                // two parsers: B0 also says "the table is clear" to the second one, "window and counters are" to both
                if (W0) {
                    if (ZZ_L2P_PRIO_P != 3) __builtin_amdgcn_s_setprio(ZZ_L2P_PRIO_P);
                    l2p_token_pass(T, hb, xb, covw, mstw, histP, src, end, l1p_make_src(P, src, end), n, before, 0u, P.err, P.dbg_viol, P.prof);
                    if (ZZ_L2P_PRIO_P != 3) __builtin_amdgcn_s_setprio(3);
                }
                if (PB) l2p_token_pass(T, hb, xb, covw, mstw, histP, src, end, l1p_make_src(P, src, end), n, before, 1u, P.err, P.dbg_viol, P.prof);
            } else if (W0) {
                if (BIAS) {
                    // warm window: every position of the last P.warm bytes in front of the packet under the hash of its
                    // own three bytes (CalcHash(source + j), :388), ascending, per hash the highest stays
                    uint32_t viol = 0;
                    warm_prehash<BIAS, false>(T, src, (int32_t)(before < P.warm ? before : P.warm), end, 0, (uint32_t)(ZZ_L2_LDS_BYTES / 2), viol);
                    if (viol | P.dbg_viol) atomicOr(P.err, ZZ_ERR_LDS_ORDER);     // (per lane, no ballot: this kernel has no scalar register to spare)
                }
                // 16-byte loads (own bytes, next block's prefetch) may run up to 15 bytes past the packet's last byte:
                // bounds-checked loads wherever that would leave the shard (by bytes: packets may be one byte long)
                l2_token_pass<BIAS>(T, hb, src, end, l1p_make_src(P, src, end), n, before, P.prof);
            }
            if (!W0 && !PB) {
                uint32_t nb = 0, adA = 0;
                uint64_t adC = 0;
                const uint32_t nt = PP ? l2p_helper_pass(hb, covw, mstw, histP, tokens, recs, src, n, nb, adA, adC, xb, P.err, snap, P.prof)
                                       : l2_helper_pass(hb, covw, mstw, histP, tokens, recs, src, n, nb, adA, adC, XD ? l6_trips(n) : l2_probe_blocks(n));
                if (lane == 0) { covw[0] = ((uint64_t)nt << 32) | nb; }       // the window is dead now
                if (P.cks_kind == ZZ_CKS_ADLER) {
                    if (lane == 0 && len > n) { const uint32_t d = src[n]; adA += d; adC += (uint64_t)n * d; }   // the byte of the alignment block
                    const uint64_t At = wave_sum64(adA), Ct = wave_sum64(adC);
                    if (lane == 0) {
                        zz_cks c;
                        c.a = (uint32_t)(At % ZZ_ADLER_MOD);
                        c.b = (uint32_t)(((uint64_t)len * At - Ct) % ZZ_ADLER_MOD);
                        P.cks[k] = c;
                    }
                }
            }
            __syncthreads();   // counts, window and the helper's global stores are complete
        }
        uint32_t* share = misc + 4;      // [0] 1 stored / 2 dynamic, [1] first record of the helper's part, [2] bits in front of
                                         // the records, [3] wavefront 0's last (partial) word, [4] the helper's first word
                                         // PP: [5] first record of the second parser's part, [6] the helper's last (partial) word,
                                         // [7] the second parser's first word, [8] index of the word the first two parts share
        if (PB) {
            // PP: the body goes out in THREE parts -- wavefront 0 the records [0, r1), the helper [r1, r2), this wavefront [r2, nbody),
            // the end of the block and the end of the packet. A part's place in the bit stream follows from a dry run over the records
            // in front of it; the words two parts share are put together here at the end.
            if (n > 0) {
                if (ZZ_L2P_DIST_ON_PB && !ZZ_L2P_HELPER_CODES) {
                    // the distance code's lengths (30 symbols) beside wavefront 0's literal / length code (286): the heap replay is one lane's
                    // chain of LDS round trips either way, and this wavefront has nothing else to do until the codes are there
                    __syncthreads();         // (D1) the counts are unpacked
                    huff_scratch S2;
                    S2.rec_freq = (uint32_t*)(lds + 12800); S2.rec_id = nullptr; S2.t_freq = nullptr;
                    S2.t_left = (uint16_t*)(lds + 12928); S2.t_right = nullptr; S2.t_bits = (uint8_t*)(lds + 13184);
                    calc_lengths_w(S2, distF, 30, 15, lens + 288);
                    __syncthreads();         // (D2) the distance lengths are in place
                }
                __syncthreads();             // (X) codes are ready, or the block went out stored
                if (uniform(share[0]) == 2) {
                    const uint32_t nbody = (uint32_t)covw[0], r2 = uniform(share[5]);
                    uint32_t m2 = 0;
                    const uint32_t* const cut = misc + 48;
                    const bool known = uniform(cut[4]) != 0;
                    if (known) m2 = uniform(cut[3]);
                    const uint32_t bits2 = known ? uniform(cut[1]) : l2_count_bits(recs, tokens, codes, dcodes, r2, m2);
                    bitring ring3;
                    ring_init_at(ring3, (uint32_t*)(lds + RING3), out, uniform(share[2]) + bits2, share + 7);
                    l2_emit_records(ring3, recs, tokens, codes, dcodes, r2, nbody, m2);
                    {   // codes[256] (:300)
                        const uint32_t cd = codes[256];
                        ring_append_uniform64(ring3, cd & 0xFFFF, cd >> 16);
                    }
                    if (!is_final) {
                        // one stored byte = byte alignment (zzflate.cpp:118-120)
                        ring_append_uniform64(ring3, 0, 3);
                        ring_pad_to_byte(ring3);
                        ring_append_uniform64(ring3, 0xFFFE0001u, 32);
                        ring_append_uniform64(ring3, src[len - 1], 8);
                    }
                    const uint32_t bytes = ring_finish_hold(ring3);
                    __syncthreads();         // (Y) the other parts' boundary words are in `share`
                    if (lane == 0) {
                        const uint32_t iA = share[8], iB = ring3.hold;
                        if (iA == iB) ring3.out32[iB] = share[3] | share[4] | share[6] | share[7];     // (the helper's part began and ended in one word)
                        else { ring3.out32[iA] = share[3] | share[4]; ring3.out32[iB] = share[6] | share[7]; }
                        P.sizes[k] = bytes;
                        if (bytes > P.slot_stride) atomicOr(P.err, 1u);
                    }
                }
            }
            return;
        }
        if (ROLE == 1 && n == 0 && P.cks_kind == ZZ_CKS_ADLER) {     // (n > 0: summed during the token pass)
            zz_cks c = wave_adler(src, len);
            if (lane == 0) P.cks[k] = c;
        }
        if (MID) {
            if (n > 0) {
                if (PP && ZZ_L2P_DIST_ON_PB && !ZZ_L2P_HELPER_CODES) { __syncthreads(); __syncthreads(); }      // (D1), (D2)
                __syncthreads();             // (X) codes are ready, or the block went out stored
                if (uniform(share[0]) == 2) {
                    // ... then the second part of the records, the end of the block and the end of the packet. Its
                    // place in the bit stream follows from a dry run over the first part.
                    const uint32_t nbody = (uint32_t)covw[0], r1 = uniform(share[1]);
                    uint32_t m1 = 0;
                    const uint32_t* const cut = misc + 48;
                    const bool known = PP && uniform(cut[4]) != 0;
                    if (known) m1 = uniform(cut[2]);
                    const uint32_t bits1 = known ? uniform(cut[0]) : l2_count_bits(recs, tokens, codes, dcodes, r1, m1);
                    bitring ring2;
                    ring_init_at(ring2, (uint32_t*)(lds + RING2), out, uniform(share[2]) + bits1, share + 4);
                    if (PP) {
                        // the middle part: its first word is held back like the last part's, its last (partial) word is handed on
                        const uint32_t r2 = uniform(share[5]);
                        l2_emit_records(ring2, recs, tokens, codes, dcodes, r1, r2, m1);
                        ZZ_WAVE_SYNC();
                        if (lane == 0) {
                            share[6] = (ring2.bitpos & 31) ? ring2.ring[(ring2.bitpos >> 5) & (ZZ_RING_WORDS - 1)] : 0u;
                            share[8] = ring2.hold;
                        }
                        __syncthreads();     // (Y)
                        return;
                    }
                    l2_emit_records(ring2, recs, tokens, codes, dcodes, r1, nbody, m1);
                    {   // codes[256] (:300)
                        const uint32_t cd = codes[256];
                        ring_append_uniform64(ring2, cd & 0xFFFF, cd >> 16);
                    }
                    if (!is_final) {
                        // one stored byte = byte alignment (zzflate.cpp:118-120)
                        ring_append_uniform64(ring2, 0, 3);
                        ring_pad_to_byte(ring2);
                        ring_append_uniform64(ring2, 0xFFFE0001u, 32);
                        ring_append_uniform64(ring2, src[len - 1], 8);
                    }
                    const uint32_t bytes = ring_finish_hold(ring2);
                    __syncthreads();         // (Y) wavefront 0's last word is in share[3]
                    if (lane == 0) {
                        ring2.out32[ring2.hold] = share[4] | share[3];
                        P.sizes[k] = bytes;
                        if (bytes > P.slot_stride) atomicOr(P.err, 1u);
                    }
                }
            }
            return;
        }
        // ---- the coding wavefront alone from here to the end of the packet ---------------------------------------------
        if (PP && ZZ_L2P_HELPER_CODES) __builtin_amdgcn_s_setprio(3);      // (the helper's token-pass priority is its own: l2p_helper_pass sets it per packet)
        bitring ring;
        ring_init(ring, ring_words, out);
        if (n > 0) {
            const uint32_t nbody = (uint32_t)covw[0], ntok = (uint32_t)(covw[0] >> 32);   // records (literals + matches), matches
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // this scratch held the previous packet's records: drop stale L1 lines
            // the hash table is dead now: its space is reused. Unpack the counters (GetFrequencies, :442-471).
            for (int i = lane; i < 320; i += ZZ_WAVE) {
                uint32_t v = 0;
                if (i < 286) v = (histP[i >> 1] >> ((i & 1) << 4)) & 0xFFFF;
                else if (i >= 288 && i < 318) { const int j = 286 + (i - 288); v = (histP[j >> 1] >> ((j & 1) << 4)) & 0xFFFF; }
                symF[i] = v;
            }
            if (lane < 20) metaF[lane] = 0;
            ZZ_T(1); ZZ_C(10, 1); ZZ_C(11, ntok);
            ZZ_WAVE_SYNC();

            ZZ_T(2);
            // ================= code construction =================================================================
            constexpr bool DIST_PB = PP && ZZ_L2P_DIST_ON_PB && !ZZ_L2P_HELPER_CODES;     // the second parser builds the distance code's lengths meanwhile
            if (DIST_PB) __syncthreads();                                         // (D1) the counts are unpacked
            if (lane == 0) symF[256] += 1;                                       // :470
            ZZ_WAVE_SYNC();
            if (XD) pm_lengths_w(PM, symF, 286, 15, lens); else calc_lengths_w(S, symF, 286, 15, lens);   // ComputeCodes, :171-176
            ZZ_T(6);
            uint32_t bitsum = 0;
            for (int i = lane; i < 286; i += ZZ_WAVE) {                          // CountBits, :178-187
                const uint32_t eb = i < 265 || i == 285 ? 0 : (uint32_t)(i - 261) >> 2;
                bitsum += symF[i] * (lens[i] + eb);
            }
            if (DIST_PB) __syncthreads();                                         // (D2) the distance lengths are in place
            else if (XD) pm_lengths_w(PM, distF, 30, 15, lens + 288); else calc_lengths_w(S, distF, 30, 15, lens + 288);
            if (lane < 30) {
                const uint32_t eb = lane < 4 ? 0 : (uint32_t)(lane - 2) >> 1;
                bitsum += distF[lane] * (lens[288 + lane] + eb);
            }
            ZZ_T(7);
#if ZZ_L2_RLE_W
            {
                ZZ_WAVE_SYNC();
                const int nr = rle_lengths_w<286>(lens, rle, 0, metaF);
                const int nr2 = rle_lengths_w<30>(lens + 288, rle, nr, metaF);
                if (lane == 0) { misc[1] = (uint32_t)nr; misc[2] = (uint32_t)nr2; }
            }
#else
            if (lane == 0) {
                int nr = rle_lengths(lens, 286, rle, 0, metaF);
                misc[1] = (uint32_t)nr;
                misc[2] = (uint32_t)rle_lengths(lens + 288, 30, rle, nr, metaF);
            }
#endif
            ZZ_WAVE_SYNC();
            ZZ_T(8);
            if (XD) pm_lengths_w(PM, metaF, 19, 7, metaLens); else calc_lengths_w(S, metaF, 19, 7, metaLens);   // :263-265
            const uint32_t nrec = misc[2];
            for (uint32_t i = lane; i < nrec; i += ZZ_WAVE) {                    // WriteLengths<LengthCounter>, :20-46
                const uint32_t v = rle[i] & 0xFF;
                bitsum += metaLens[v] + (v == 16 ? 2 : v == 17 ? 3 : v == 18 ? 7 : 0);
            }
            const uint32_t total = 3 + 5 + 5 + 4 + 3 * 19 + wave_sum(bitsum);    // :267
            const uint32_t required = (total + 8) / 8;                           // requiredLength, :271
            ZZ_WAVE_SYNC();
            ZZ_T(3);

            if (required >= n) {
                // ================= UncompressedFallback (encoder.cpp:305-317, 482-502) ==========================
                // n <= 32767 < 65535: one stored block. BFINAL as passed by the caller (:274).
                if (lane == 0) {
                    out[0] = is_final ? 1 : 0;
                    out[1] = (uint8_t)n; out[2] = (uint8_t)(n >> 8);
                    out[3] = (uint8_t)~n; out[4] = (uint8_t)(~n >> 8);
                }
                coop_copy(out + 5, src, n, lane, ZZ_WAVE);
                ZZ_WAVE_DRAIN();
                uint32_t bytes = 5 + n;
                if (!is_final) {
                    if (lane == 0) {
                        uint8_t* t = out + bytes;
                        t[0] = 0; t[1] = 1; t[2] = 0; t[3] = 0xFE; t[4] = 0xFF; t[5] = src[len - 1];
                    }
                    bytes += 6;
                }
                if (lane == 0) { P.sizes[k] = bytes; share[0] = 1; }
                __syncthreads();             // (X)
                return;
            }

            // ================= dynamic block ====================================================================
            generate_codes_w(lens, 286, codes, misc + 16);                        // huffman::generate
            generate_codes_w(lens + 288, 30, dcodes, misc + 16);
            generate_codes_w(metaLens, 19, metaCodes, misc + 16);
            ZZ_WAVE_SYNC();
            ZZ_T(4);
            // StartBlock(UserDefinedHuffman, final) + HLIT=29 HDIST=29 HCLEN=15 (:280-285)
            ring_append_uniform(ring, (is_final ? 1u : 0u) | (2u << 1) | (29u << 3) | (29u << 8) | (15u << 13), 17);
            {   // 19 x 3 bits in `order` (:287-290)
                const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
                uint32_t v = 0;
                for (int i = 0; i < 19; ++i) if (lane == i) v = metaLens[order[i]];
                ring_append(ring, v, lane < 19 ? 3 : 0);
            }
            for (uint32_t r0 = 0; r0 < nrec; r0 += 64) {                          // WriteLengths x2 (:292-293)
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
            // body: both wavefronts emit. This one takes the records [0, r1), the helper the rest and everything behind
            // them; the word the two parts share is put together by the helper at the end.
            {
                // (two parts: a little more than half here, the helper also has a dry run to do; PP: three parts, each later one with
                // a longer dry run in front of it)
                uint32_t r1 = PP ? ((nbody * ZZ_L2P_SPLIT1) >> 6) & ~63u : ((nbody * ZZ_L2_SPLIT64) >> 6) & ~63u;
                uint32_t r2 = ((nbody * ZZ_L2P_SPLIT2) >> 6) & ~63u;
                if (r2 < r1) r2 = r1;
                uint32_t* const cut = misc + 48;     // PP: [0], [1] bits of the records in front of the second / third part, [2], [3] matches among them, [4] 1 = all of this is known
                if (PP && l2p_cut(n, 1) != 0xFFFFFFFFu) {
                    // the body was cut by blocks (l2p_helper_pass): the records in front of a cut are worth sum count x (code length +
                    // extra bits) over the symbols as counted there (CountBits' sum, :178-187) -- no dry run over them
                    uint32_t b1 = 0, b2 = 0;
                    auto cnt = [&](const uint32_t* sp, int j) -> uint32_t { return (sp[j >> 1] >> ((j & 1) << 4)) & 0xFFFFu; };
                    for (int i = lane; i < 286; i += ZZ_WAVE) {
                        const uint32_t w = lens[i] + (i < 265 || i == 285 ? 0u : (uint32_t)(i - 261) >> 2);
                        b1 += cnt(snap, i) * w; b2 += cnt(snap + ZZ_L2_SNAP_WORDS, i) * w;
                    }
                    if (lane < 30) {
                        const uint32_t w = lens[288 + lane] + (lane < 4 ? 0u : (uint32_t)(lane - 2) >> 1);
                        b1 += cnt(snap, 286 + lane) * w; b2 += cnt(snap + ZZ_L2_SNAP_WORDS, 286 + lane) * w;
                    }
                    // ... less the matches that had arrived but start behind the cut (a match may begin up to 258 bytes in front of the block it
                    // was found in: the counters are read when the block in front of the cut is final, five blocks on): a few dozen tokens
                    const uint32_t m1 = snap[158], t1 = snap[160], m2 = snap[ZZ_L2_SNAP_WORDS + 158], t2 = snap[ZZ_L2_SNAP_WORDS + 160];
                    auto tokbits = [&](uint32_t t) -> uint32_t {
                        const uint32_t ls = t >> 23, bucket = (t >> 13) & 31;
                        return lens[257 + ls] + ((ls < 8 || ls == 28) ? 0u : (ls - 4) >> 2) + lens[288 + bucket] + (bucket < 4 ? 0u : (bucket - 2) >> 1);
                    };
                    for (uint32_t j = m1 + (uint32_t)lane; j < t1; j += ZZ_WAVE) b1 -= tokbits(tokens[j]);
                    for (uint32_t j = m2 + (uint32_t)lane; j < t2; j += ZZ_WAVE) b2 -= tokbits(tokens[j]);
                    b1 = wave_sum(b1); b2 = wave_sum(b2);
                    r1 = snap[159]; r2 = snap[ZZ_L2_SNAP_WORDS + 159];
                    if (lane == 0) { cut[0] = b1; cut[1] = b2; cut[2] = m1; cut[3] = m2; cut[4] = 1; }
                } else if (PP && lane == 0) cut[4] = 0;
                if (lane == 0) { share[0] = 2; share[1] = r1; share[2] = ring.bitpos; if (PP) { share[4] = 0; share[5] = r2; share[7] = 0; } }
                __syncthreads();             // (X)
                l2_emit_records(ring, recs, tokens, codes, dcodes, 0, r1, 0);
                ZZ_WAVE_SYNC();
                if (lane == 0) share[3] = (ring.bitpos & 31) ? ring.ring[(ring.bitpos >> 5) & (ZZ_RING_WORDS - 1)] : 0u;
                __syncthreads();             // (Y)
                ZZ_T(5);
                return;
            }
        }
        if (!is_final) {
            // one stored byte = byte alignment (zzflate.cpp:118-120)
            ring_append_uniform(ring, 0, 3);
            ring_pad_to_byte(ring);
            ring_append_uniform(ring, 0xFFFE0001u, 32);
            ring_append_uniform(ring, src[len - 1], 8);
        } else if (n == 0) {
            ring_append_uniform(ring, 1u | (1u << 1), 3);   // empty input: one empty fixed block (D8 divergence)
            ring_append_uniform(ring, 0, 7);
        }
        const uint32_t bytes = ring_finish(ring);
        if (lane == 0) {
            P.sizes[k] = bytes;
            if (bytes > P.slot_stride) atomicOr(P.err, 1u);
        }
        ZZ_T(5);
    };
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);      // the parse is the packet's critical path: ahead of the helpers in the issue arbiter
        for (uint32_t k = Q.k0 + blockIdx.x; k < Q.k1; k = next_packet()) packet(k, std::integral_constant<int, 0>());
    } else if (wave == 1) {
        for (uint32_t k = Q.k0 + blockIdx.x; k < Q.k1; k = next_packet()) packet(k, std::integral_constant<int, 1>());
    } else {
        __builtin_amdgcn_s_setprio(ZZ_L2P_PRIO_P);
        for (uint32_t k = Q.k0 + blockIdx.x; k < Q.k1; k = next_packet()) packet(k, std::integral_constant<int, 2>());
    }
#ifdef ZZ_PROF
    if (threadIdx.x == ((PP && ZZ_L2P_HELPER_CODES) ? 64u : 0u) && P.prof) for (int _i = 0; _i < 16; ++_i) atomicAdd(&P.prof[_i], prof_acc[_i]);     // the coding wavefront's stamps
#endif
}

// xdepth: 0 = levels 2,3; 2 / 4 / 8 = the extended levels 4 / 5 / 6 (chain depth)
static inline uint32_t l2_grid(uint32_t npk, bool xd = false, uint32_t per_cu = 9)
{
    const uint32_t resident = 256 * (xd ? 11 : per_cu);     // what the LDS budget (PP: and the register budget) admits
    return npk < resident ? npk : resident;
}
// the extended levels go through the input in batches of about 1 GiB: k_l6_matches over a batch's packets (one word per input
// byte: 4 GiB), then the encode kernel over the same packets
static inline uint32_t l6_batch_packets(uint32_t npk, uint32_t P)
{
    static const uint64_t bytes = [] { const char* e = getenv("ZZFLATE_L6_BATCH_MIB"); const long v = e ? atol(e) : 0; return (uint64_t)(v >= 1 && v <= 65536 ? v : 1024) << 20; }();
    const uint64_t b = bytes / P ? bytes / P : 1;
    return (uint32_t)(b < npk ? b : npk);
}
static inline uint32_t l6_match_grid()      // one workgroup per CU: it takes the CU's LDS
{
    static const uint32_t g = [] { const char* e = getenv("ZZFLATE_L6_MATCH_WGS"); const int v = e ? atoi(e) : 0; return (uint32_t)(v >= 1 && v <= 1024 ? v : 256); }();
    return g;
}
static inline uint64_t l6_m_bytes(uint32_t npk, uint32_t P) { return ((uint64_t)l6_batch_packets(npk, P) * P * 4u + 255u) & ~255ull; }
static inline uint64_t l2_scratch_bytes(uint32_t npk, int xdepth, uint32_t P)
{
    uint64_t need = (uint64_t)l2_grid(npk, xdepth != 0) * ZZ_L2_SCRATCH_BYTES;
    if (xdepth) need += l6_m_bytes(npk, P) + (uint64_t)l6_match_grid() * 32768u * 2u * (uint32_t)xdepth;   // + the chains of every resident packet
    return need;
}
// two_parser: levels 2,3 with cold packets on k_encode_l2p (the caller decides: not behind ZZFLATE_L2_KERNEL=classic, and -- its insert
// is an ordered LDS exchange -- only where the device's LDS-order verdict is positive)
static inline void launch_level2(const zz_packet_params& pp, uint8_t* scratch, uint32_t* work, hipStream_t st, int xdepth = 0, bool two_parser = true)
{
    zz_l2_params q; q.pk = pp; q.scratch = scratch; q.work = work; q.m = nullptr; q.k0 = 0; q.k1 = pp.npk;
    if (!xdepth) {
        (void)hipMemsetAsync(work, 0, sizeof(uint32_t), st);
        const dim3 g(l2_grid(pp.npk)), b(ZZ_L2_THREADS);
        if (pp.warm) hipLaunchKernelGGL((k_encode_l2_t<32768u, false>), g, b, 0, st, q);
        else if (!two_parser) hipLaunchKernelGGL((k_encode_l2_t<0u, false>), g, b, 0, st, q);
        else hipLaunchKernelGGL((k_encode_l2_t<0u, false, true>), dim3(l2_grid(pp.npk, false, ZZ_L2P_WPE >= 7 ? 9 : 8)), dim3(ZZ_L2P_THREADS), 0, st, q);
        return;
    }
    uint32_t* const m = (uint32_t*)(scratch + (uint64_t)l2_grid(pp.npk, true) * ZZ_L2_SCRATCH_BYTES);
    const uint32_t batch = l6_batch_packets(pp.npk, pp.packet_size);
    const uint32_t mgrid = l6_match_grid();
    uint16_t* const chains = (uint16_t*)((uint8_t*)m + l6_m_bytes(pp.npk, pp.packet_size));
    for (uint32_t k0 = 0; k0 < pp.npk; k0 += batch) {
        const uint32_t k1 = pp.npk - k0 < batch ? pp.npk : k0 + batch;
        (void)hipMemsetAsync(work, 0, 2 * sizeof(uint32_t), st);
        zz_l6m_params qm; qm.pk = pp; qm.m = m; qm.chains = chains; qm.work = work + 1; qm.k0 = k0; qm.k1 = k1;
        const dim3 gm((k1 - k0) < mgrid ? (k1 - k0) : mgrid), bm(ZZ_L6M_THREADS);
        if (xdepth == 2) hipLaunchKernelGGL((k_l6_matches<2>), gm, bm, 0, st, qm);
        else if (xdepth == 4) hipLaunchKernelGGL((k_l6_matches<4>), gm, bm, 0, st, qm);
        else hipLaunchKernelGGL((k_l6_matches<8>), gm, bm, 0, st, qm);
        q.m = m; q.k0 = k0; q.k1 = k1;
        const dim3 g(l2_grid(k1 - k0, true)), b(ZZ_L2_THREADS);
        hipLaunchKernelGGL((k_encode_l2_t<32768u, true>), g, b, 0, st, q);
    }
}

}  // namespace zz
