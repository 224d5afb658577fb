// zz_level6.h -- the extended levels 4, 5, 6: bounded hash chains, one-step lazy matching, code lengths
// by package-merge. NOT in the reference, which has one slot per hash, a greedy parse, a frequency-floor length limiter and
// rejects level > 3 (encoder.h:41-43,76; encoder.cpp:388-424; huffman.cpp:122-154; zzflate.cpp:201,230-234); what this
// file must be bit-exact with is the definition of these levels in the oracle, which it restates for a wavefront:
//
//   chains   every position q of the window in front of the packet and of the packet itself, up to target = n - 16, is
//            entered under a 13-bit hash of its FOUR bytes, ascending; the candidates of q are the nearest DEPTH earlier
//            positions with q's hash, as long as they are less than 32768 back. All positions are entered whatever the
//            parse does, so the chains are parse-independent -- and stored FLATTENED: the positions sorted by (hash,
//            position) in one array (a counting sort: histogram of the hashes in LDS, exclusive scan, then every block of
//            64 positions takes consecutive places in its buckets, in lane order). The DEPTH entries in front of a
//            position's own place are its chain, nearest last: one 2*DEPTH-byte load instead of DEPTH dependent hops.
//            Entries in front of a bucket's first belong to the bucket before (other four bytes: no match of four) or have
//            not been written yet (the array is zeroed per packet: position -32768, out of reach): the chain ends there.
//   match    16 bytes at q against 16 bytes at each candidate; the longest wins, the nearest among equals; >= 4 is a match;
//   lazy     q defers (stays a literal) if its match is shorter than 16 and q+1 has a longer one, unless (q & 63) == 63;
//   parse    greedy over the positions with a match that do not defer; a length of 16 is extended to its true value (<= 258)
//            only when the parse reaches it -- the only thing here that depends on the parse;
//   codes    optimal length-limited code lengths by package-merge in its list form (pm_lengths_w), wave-parallel: it replaces
//            the heap replay of levels 2,3 (which exists only because ties must fall as libstdc++'s heap lets them fall).
// The records, histograms, header and body emission are the level-2 ones (zz_level2.h), as is the two-wavefront split.
#pragma once

namespace zz {

#define ZZ_L6_BIAS 32768u                          // array entries and table positions are position + 32768
#define ZZ_L6_PAD 8u                               // entries in front of the sorted array (a chain read never starts below it)
#define ZZ_L6_SORT_ENTRIES (65536u + 64u)
#define ZZ_L6_SORT_BYTES (ZZ_L6_SORT_ENTRIES * 2u)  // positions sorted by (hash, position)
#define ZZ_L6_IDX_BYTES (32768u * 2u)               // place of every packet position in that array
#define ZZ_L6_SCRATCH_BYTES (ZZ_L6_SORT_BYTES + ZZ_L6_IDX_BYTES)
#define ZZ_L6_TAIL 16u                              // target = n - 16: the 16 bytes compared at a position lie inside the data
#define ZZ_L6_CAP 16u

__device__ __forceinline__ uint32_t l6_hash4(uint32_t four_bytes) { return (four_bytes * 2654435761u) >> (32 - ZZ_HASH_BITS); }
__device__ __forceinline__ uint32_t l6_target(uint32_t n) { return n > ZZ_L6_TAIL ? n - ZZ_L6_TAIL : 0u; }
__device__ __forceinline__ uint32_t l6_trips(uint32_t n) { return (l6_target(n) + 63u) >> 6; }

// ---- the helper wavefront's share of the preparation: the sorted array starts out as zeros --------------------------------
__device__ __forceinline__ void l6_zero_sorted(uint16_t* sorted, uint32_t entries)
{
    uint4* z = (uint4*)sorted;
    const uint32_t n16 = (entries * 2u + 15u) >> 4;
    for (uint32_t i = (uint32_t)lane_id(); i < n16; i += ZZ_WAVE) z[i] = make_uint4(0, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // written before the other wavefront's stores to the same lines are issued
}

// ---- counting sort, part 1: how many positions per hash, then where each bucket starts ------------------------------------
// T: 8192 16-bit counters (the level-2 hash table's space), zero on entry; on exit T[h] = the place of the next position with
// hash h. Positions -W .. target-1 (W + target < 65536: the counts, and the places, fit 16 bits).
__device__ __forceinline__ void l6_histogram_and_scan(uint16_t* T, const uint8_t* src, int32_t W, uint32_t target)
{
    const int lane = lane_id();
    uint32_t* Tw = (uint32_t*)T;
    const int32_t lo = -W, hi = (int32_t)target;
    for (int32_t g = lo; g < hi; g += 8 * ZZ_WAVE) {          // eight loads in flight per trip
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int32_t pos = g + u * ZZ_WAVE + lane;
            v[u] = load32(src + (pos < hi ? pos : lo));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int32_t pos = g + u * ZZ_WAVE + lane;
            const uint32_t h = l6_hash4(v[u]);
            if (pos < hi) atomicAdd(&Tw[h >> 1], 1u << ((h & 1u) << 4));
        }
    }
    ZZ_WAVE_SYNC();
    // exclusive scan over the 8192 counters: a lane owns 128 consecutive ones (64 words)
    uint32_t* mine = Tw + 64 * lane;
    uint32_t s = 0;
    for (int i = 0; i < 64; i += 4) {
        const uint4 w = *(const uint4*)(mine + i);
        s += (w.x & 0xFFFF) + (w.x >> 16) + (w.y & 0xFFFF) + (w.y >> 16) + (w.z & 0xFFFF) + (w.z >> 16) + (w.w & 0xFFFF) + (w.w >> 16);
    }
    uint32_t r = wave_scan_incl(s) - s;
    for (int i = 0; i < 64; ++i) {
        const uint32_t w = mine[i];
        const uint32_t a = r, b = r + (w & 0xFFFF);
        r = b + (w >> 16);
        mine[i] = (a & 0xFFFF) | (b << 16);
    }
    ZZ_WAVE_SYNC();
}

// ---- counting sort, part 2: every position takes its place ----------------------------------------------------------------
// Blocks of 64 positions, ascending; the positions of a block that share a hash take consecutive places in lane order (the
// read-back of a lane tag written to the bucket's counter names one lane per set of equal hashes: six ballots give every lane
// its set, zz_wave.h). The parsing wavefront only works the counters: the places of a block go to the helper wavefront
// through an LDS slot (two slots, one s_barrier per block), and the helper does the stores -- sorted[PAD + place] = position +
// BIAS, idx[q] = place for the packet's own positions. (With the stores on the parser's own memory queue, every wait for its
// next positions' bytes was a wait for all the scattered two-byte stores before them: 27 % of a level-6 packet.)
__device__ __forceinline__ uint32_t l6_place_blocks(int32_t W, uint32_t target)
{
    return ((((uint32_t)W + 63u) & ~63u) + ((target + 63u) & ~63u)) >> 6;
}
__device__ __forceinline__ void l6_place_all(uint16_t* T, uint32_t spare, uint32_t* hb, const uint8_t* src, int32_t W, uint32_t target)
{
    const int lane = lane_id();
    const uint64_t below_me = (1ull << lane) - 1;
    const int32_t lo = -W, hi = (int32_t)target;
    uint32_t slotsel = 0;
    // (blocks are aligned to the packet: the first window block may be partial; four blocks' bytes are requested per trip)
    for (int32_t g0 = -(int32_t)(((uint32_t)W + 63u) & ~63u); g0 < hi; g0 += 4 * ZZ_WAVE) {
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int32_t pos = g0 + u * ZZ_WAVE + lane;
            v[u] = load32(src + ((pos >= lo && pos < hi) ? pos : lo));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int32_t g = g0 + u * ZZ_WAVE;
            if (g >= hi) break;                                             // (uniform)
            const int32_t pos = g + lane;
            const bool act = pos >= lo && pos < hi;
            const uint32_t h = act ? l6_hash4(v[u]) : spare;                // lanes without a position share a spare counter
            const uint32_t old = T[h];
            T[h] = (uint16_t)lane;
            ZZ_WAVE_SYNC();
            const uint32_t rb = T[h];
            ZZ_WAVE_SYNC();
            uint32_t place = old, cnt = 1;
            if (ballot(rb != (uint32_t)lane)) {
                const uint64_t set = wave_match6(rb);
                place = old + (uint32_t)__builtin_popcountll(set & below_me);
                cnt = (uint32_t)__builtin_popcountll(set);
            }
            if (rb == (uint32_t)lane) T[h] = (uint16_t)(old + cnt);          // one lane per set moves the counter on
            (hb + slotsel)[lane] = place;
            slotsel ^= ZZ_L2_HB_WORDS;
            l2_block_barrier();
        }
    }
}
// the helper's side: one barrier per block, then the block's stores
__device__ __forceinline__ void l6_store_places(const uint32_t* hb, int32_t W, uint32_t target, uint16_t* sorted, uint16_t* idx)
{
    const int lane = lane_id();
    const int32_t lo = -W, hi = (int32_t)target;
    uint32_t slotsel = 0;
    for (int32_t g = -(int32_t)(((uint32_t)W + 63u) & ~63u); g < hi; g += ZZ_WAVE) {
        l2_block_barrier();
        const uint32_t place = (hb + slotsel)[lane];
        slotsel ^= ZZ_L2_HB_WORDS;
        const int32_t pos = g + lane;
        if (pos >= lo && pos < hi) {
            sorted[ZZ_L6_PAD + place] = (uint16_t)(pos + (int32_t)ZZ_L6_BIAS);
            if (pos >= 0) idx[pos] = (uint16_t)place;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // in memory before the other wavefront reads them
}

// ---- the token pass: best of the chain, lazy evaluation, greedy parse; hands every block's matches to the helper wavefront
// in the level-2 format (zz_level2.h: ZZ_L2_HB_PACK, one s_barrier per block). ---------------------------------------------
// Everything up to the parse is parse-independent, so the loads run ahead of their use as a three-stage pipeline: while
// block b-1 is compared and parsed, the candidates' bytes of block b, the chain of block b+1 and the place of block b+2 are
// in flight (a block's loads depend on each other: place -> chain -> bytes, three trips to the L2 that would otherwise stand
// in front of every block).
template <int DEPTH, bool SAFE>
__device__ __forceinline__ void l6_match_pass(uint32_t* hb, const uint8_t* src, const uint8_t* end, uint32_t n,
                                              const uint16_t* sorted, const uint16_t* idx)
{
    const int lane = lane_id();
    const uint32_t target = l6_target(n);
    const uint32_t nblk = (target + 63u) >> 6;
    uint32_t nextpos = 0;                   // first position the parse has not decided
    uint32_t slotsel = 0;
    struct chain_t { uint16_t c[DEPTH]; };
    struct bytes_t { uint64_t w, w2, cw[DEPTH], cw2[DEPTH]; uint32_t dist[DEPTH]; };
    // stage A: where block b's positions stand in the sorted array
    auto stageA = [&](uint32_t b) -> uint32_t {
        const uint32_t q = (b << 6) + (uint32_t)lane;
        return idx[q < target ? q : 0u];                                    // (lanes past the target: any valid place)
    };
    // stage B: the chain -- the DEPTH entries in front of the position's place, nearest last
    auto stageB = [&](uint32_t place) -> chain_t {
        chain_t ch;
        __builtin_memcpy(ch.c, sorted + ZZ_L6_PAD + place - DEPTH, 2 * DEPTH);
        return ch;
    };
    // stage C: 16 bytes at the position and at every candidate
    auto stageC = [&](uint32_t b, const chain_t& ch) -> bytes_t {
        bytes_t y;
        const uint32_t q = (b << 6) + (uint32_t)lane;
        const bool act = q < target;
        const uint32_t qa = act ? q : 0u;
        ld128<false>(src + qa, end, y.w, y.w2);                             // q + 16 <= n: inside the data
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {                                   // k = 0: the nearest
            const int32_t c = (int32_t)ch.c[DEPTH - 1 - k] - (int32_t)ZZ_L6_BIAS;
            const uint32_t d = q - (uint32_t)c;                             // 0 < d < 32768: a candidate
            const bool ok = act && (d - 1u) < 32767u;
            y.dist[k] = ok ? d : 0u;
            ld128<false>(src + (ok ? c : (int32_t)qa), end, y.cw[k], y.cw2[k]);
        }
        return y;
    };
    // stage D: best of the chain, lazy evaluation, the parse, hand-over
    auto stageD = [&](uint32_t b, const bytes_t& y) {
        const uint32_t base = b << 6;
        const uint32_t q = base + (uint32_t)lane;
        uint32_t best = 0, bdist = 0;
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            uint32_t len = equal_bits128(y.w ^ y.cw[k], y.w2 ^ y.cw2[k], 128u) >> 3;
            if (!y.dist[k]) len = 0;
            if (len > best) { best = len; bdist = y.dist[k]; }
        }
        if (best < 4) best = 0;
        // lazy: the next position's length (lane 63 never defers)
        const uint32_t nxt = (uint32_t)__shfl_down((int)best, 1);
        const bool defer = best != 0 && best < ZZ_L6_CAP && lane != 63 && nxt > best;
        const uint64_t E = ballot(best != 0 && !defer);
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
                len = wave_extend_match<SAFE>(src, qe, (int32_t)(qe - readlane(bdist, e)), maxlen, end, ZZ_L6_CAP);   // (4-byte loads: may look 3 bytes past n)
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
    };
    if (nblk == 0) return;
    // prologue: fill the pipeline
    uint32_t placeA = stageA(0);
    chain_t chB = stageB(placeA);
    placeA = stageA(nblk > 1 ? 1u : 0u);
    bytes_t yC = stageC(0, chB);
    chB = stageB(placeA);
    placeA = stageA(nblk > 2 ? 2u : 0u);
    for (uint32_t b = 0; b < nblk; ++b) {
        // requests for the blocks behind this one go out first ...
        bytes_t yN;
        const bool more = b + 1 < nblk;
        if (more) yN = stageC(b + 1, chB);
        chB = stageB(placeA);
        placeA = stageA(b + 3 < nblk ? b + 3 : 0u);
        // ... then this block is worked on while they travel
        stageD(b, yC);
        if (more) yC = yN;
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
