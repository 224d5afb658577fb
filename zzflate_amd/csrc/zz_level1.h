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
//   5. tokens are turned into fixed-Huffman fragments and appended through the bit ring (zz_emit.h) -- one
//      iteration later, in the shadow of the next group's candidate loads (software pipelining).
//
// Measured on MI355X (rocprofv3 SQ counters, profiles/): a parsing wave retires about one instruction per 8 cycles
// and only ~9 packets fit a CU (the 16 KiB table), so the loop is bound by the parser's dynamic instruction count and
// by dependent LDS/memory round trips, not by bandwidth. Hence: a second wavefront per workgroup takes everything that
// is not on the parse's dependency chain (Adler-32, Huffman coding, bit packing; one s_barrier per group of 64
// positions hands the tokens over), no barrier inside a wave (LDS ops of one wave execute in order), candidate loads
// issued before the same-hash scan, a branch-light hand-written fast path in the walk, and bounds-checked loads only
// where a 16-byte load could leave the shard.
#pragma once
#include "zz_checksum.h"
#include "zz_emit.h"

namespace zz {

template <bool SAFE> __device__ __forceinline__ uint64_t ld64(const uint8_t* p, const uint8_t* end)
{
    if (SAFE) return load64_safe(p, end);
    return load64(p);
}
// 16 bytes at p as two little-endian words
template <bool SAFE> __device__ __forceinline__ void ld128(const uint8_t* p, const uint8_t* end, uint64_t& lo, uint64_t& hi)
{
    if (!SAFE || p + 16 <= end) {
        uint4 v;
        __builtin_memcpy(&v, p, 16);
        lo = ((uint64_t)v.y << 32) | v.x;
        hi = ((uint64_t)v.w << 32) | v.z;
    } else {
        lo = load64_safe(p, end);
        hi = load64_safe(p + 8, end);
    }
}
template <bool SAFE> __device__ __forceinline__ uint32_t ld32(const uint8_t* p, const uint8_t* end)
{
    if (SAFE) return load32_safe(p, end);
    return load32(p);
}

// compare src[pe+from ..) with src[cand+from ..) with the whole wave, 4 bytes per lane; returns the match
// length (>= from) clamped to maxlen. Only called when the first `from` bytes are known equal.
template <bool SAFE>
__device__ __forceinline__ uint32_t wave_extend_match(const uint8_t* src, uint32_t pe, int32_t cand,
                                                      uint32_t maxlen, const uint8_t* end, uint32_t from = 8)
{
    const uint32_t o = from + 4 * (uint32_t)lane_id();
    uint32_t d = 0;
    const bool act = o < maxlen;
    if (act) d = ld32<SAFE>(src + pe + o, end) ^ ld32<SAFE>(src + (cand + (int32_t)o), end);
    const uint64_t neq = ballot(act && d != 0);
    if (!neq) return maxlen;
    const int k = __builtin_ctzll(neq);
    const uint32_t dk = readlane(d, k);
    const uint32_t len = from + 4 * (uint32_t)k + ((uint32_t)__builtin_ctz(dk) >> 3);
    return len < maxlen ? len : maxlen;
}

// number of equal leading bytes of two 16-byte strings given as the XOR of their two little-endian words, 16 = all.
// Branch-free on purpose: every exec-mask region costs three scalar instructions, and the CU's one scalar unit is the
// scarce resource (tools/ubench_valu.hip).
__device__ __forceinline__ uint32_t equal_bytes16(uint64_t x, uint64_t x2)
{
    const uint32_t c1 = x ? (uint32_t)__builtin_ctzll(x) : 64u;
    const uint32_t c2 = x2 ? (uint32_t)__builtin_ctzll(x2) : 64u;
    return (c1 < 64u ? c1 : 64u + c2) >> 3;
}
// the same in bits, for callers that cap the count themselves: min(cap, 8 x the number of equal leading bytes), and cap if
// all 16 bytes are equal. Ten instructions: v_ffbl_b32 gives -1 for 0, the additions saturate, the minima pick. (Inline
// asm, because the compiler knows that ctz + 32 cannot overflow and turns the saturating additions back into compares
// and selects, each with its wait states.)
__device__ __forceinline__ uint32_t ffbl_or_ones(uint32_t v)
{
    uint32_t r;
    asm("v_ffbl_b32_e32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
// v_ffbh_u32: the number of leading zero bits, and -1 (not 32) for 0
__device__ __forceinline__ uint32_t ffbh_or_ones(uint32_t v)
{
    uint32_t r;
    asm("v_ffbh_u32_e32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
template <int K> __device__ __forceinline__ uint32_t add_sat_k(uint32_t v)
{
    uint32_t r;
    asm("v_add_u32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(v), "n"(K));
    return r;
}
__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t equal_bits128(uint64_t x, uint64_t x2, uint32_t cap)
{
    const uint32_t f0 = ffbl_or_ones((uint32_t)x), f1 = add_sat_k<32>(ffbl_or_ones((uint32_t)(x >> 32)));
    const uint32_t f2 = ffbl_or_ones((uint32_t)x2), f3 = add_sat_k<32>(ffbl_or_ones((uint32_t)(x2 >> 32)));
    const uint32_t hi = add_sat_k<64>(f2 < f3 ? f2 : f3);
    const uint32_t lo = umin3(f0, f1, cap);
    return lo < hi ? lo : hi;
}
__device__ __forceinline__ uint32_t calc_hash3(uint32_t three_bytes)   // encoder.cpp:11-17
{
    return ((three_bytes & 0xFFFFFFu) * 0x00d68664u) >> (32 - ZZ_HASH_BITS);
}

// Loads that may run past the shard's last byte (only a shard's last packet has any, and only in its last blocks) are turned to
// a 128-byte copy of the shard's end that the host keeps behind it: its last 64 bytes, then zeros (zz_packet_params::tail, filled
// by k_fill_tail before the launch). One compare and one select per load, where bounds-checked byte-wise loads (load64_safe)
// cost the kernel its register budget: 65 VGPRs and 100 SGPRs with them, 62 and 62 without.
struct l1p_src {
    const uint8_t* src;        // the packet's first byte
    const uint8_t* tailp;      // tail copy, biased: tailp + position = the copy's byte for that position
    int32_t lim;               // positions above this one (packet-relative) read from the copy: shard end - 16
};
// (positions are packet-relative and may be negative: a candidate in the window in front of the packet, levels >= 2)
template <bool TAIL> __device__ __forceinline__ const uint8_t* l1p_addr(const l1p_src& S, int32_t pos)
{
    if (!TAIL) return S.src + pos;
    return (pos > S.lim ? S.tailp : S.src) + pos;
}
// src: the packet's first byte; end: one past the shard's last readable byte (P.src + P.n); the copy starts at
// max(shard end - 64, shard start) (k_fill_tail)
__device__ __forceinline__ l1p_src l1p_make_src(const zz_packet_params& P, const uint8_t* src, const uint8_t* end)
{
    l1p_src S;
    const int64_t endrel = (int64_t)(end - src);
    const uint64_t tn = P.n < 64 ? P.n : 64;
    S.src = src;
    S.lim = endrel - 16 > 0x7fffffff ? 0x7fffffff : (int32_t)(endrel - 16);
    S.tailp = P.tail - (endrel - (int64_t)tn);
    return S;
}
template <bool TAIL> __device__ __forceinline__ void l1p_ld128(const l1p_src& S, int32_t pos, uint64_t& lo, uint64_t& hi)
{
    uint4 v;
    __builtin_memcpy(&v, l1p_addr<TAIL>(S, pos), 16);
    lo = ((uint64_t)v.y << 32) | v.x;
    hi = ((uint64_t)v.w << 32) | v.z;
}
// the same for a position that is known not to be negative (an unsigned offset from the uniform base is the load's own
// addressing mode; a signed one costs a sign extension and a 64-bit addition in front of it)
template <bool TAIL> __device__ __forceinline__ void l1p_ld128u(const l1p_src& S, uint32_t pos, uint64_t& lo, uint64_t& hi)
{
    uint4 v;
    if (TAIL) __builtin_memcpy(&v, l1p_addr<true>(S, (int32_t)pos), 16);
    else __builtin_memcpy(&v, S.src + pos, 16);
    lo = ((uint64_t)v.y << 32) | v.x;
    hi = ((uint64_t)v.w << 32) | v.z;
}
// wave_extend_match (zz_level1.h) over these loads
__device__ __forceinline__ uint32_t l1p_extend_match(const l1p_src& S, uint32_t pe, int32_t cand, uint32_t maxlen, uint32_t from = 8)
{
    const uint32_t o = from + 4 * (uint32_t)lane_id();
    uint32_t d = 0;
    const bool act = o < maxlen;
    if (act) d = load32(l1p_addr<true>(S, (int32_t)(pe + o))) ^ load32(l1p_addr<true>(S, cand + (int32_t)o));
    const uint64_t neq = ballot(act && d != 0);
    if (!neq) return maxlen;
    const int k = __builtin_ctzll(neq);
    const uint32_t dk = readlane(d, k);
    const uint32_t len = from + 4 * (uint32_t)k + ((uint32_t)__builtin_ctz(dk) >> 3);
    return len < maxlen ? len : maxlen;
}

// per-lane walk info, packed into one VGPR so that the scalar walk needs a single v_readlane per event. The match
// length against the table candidate sits in bits 0..5 (bit 5 always clear) because s_bfm_b64 takes its field size from
// exactly those bits of its operand: the walk never has to extract it.
#define ZZ_WI_LENA(i) ((i) & 31u)            // match length vs the table candidate, 16 = "16 or more"
#define ZZ_WI_LENB(i) (((i) >> 6) & 31u)     // match length vs the nearest earlier same-hash lane
#define ZZ_WI_LENB_SHIFT 6
#define ZZ_WI_QLANE(i) (((i) >> 11) & 63u)   // that lane
#define ZZ_WI_QLANE_SHIFT 11
#define ZZ_WI_DUP 0x20000u                   // lane has an earlier same-hash lane in the group
#define ZZ_WI_HARD 0x40000u                  // its hash occurs more than twice
#define ZZ_WI_EXTA 0x80000u                  // lenA is "16 or more" and more bytes remain: extend
#define ZZ_WI_EXTB 0x100000u
#define ZZ_WI_NEXT_SHIFT 21                  // plain matches: first event lane at or after the match end (0 = none; lane 0 never is one)
#define ZZ_WI_CAP 16u                        // bytes compared up front (two 8-byte words per lane)

// One hop of the walk's inner loop: a plain match at event lane EIN (length lenA < 16 against the table candidate),
// then EOUT = the first event at or after its end. The CU has ONE scalar unit for all its wavefronts (one instruction
// per cycle: tools/ubench_valu.hip), so with nine parsers per CU every scalar instruction of this loop costs ~16
// cycles; hence as few as possible: s_and and s_bfe set SCC themselves (no compare), s_bfm takes the length straight
// from the packed word, the position is only worked out when the loop is left. Unrolled twice with the event lane
// alternating between two registers, so that only every other hop pays for a taken branch.
#define ZZ_L1_HOP(EIN, EOUT, FLAGGED) \
        "v_readlane_b32 %[inf], %[info], " EIN "\n\t" \
        "s_and_b32 %[len], %[inf], 0xe0000\n\t"    /* DUP | HARD | EXTA: not a plain match */ \
        "s_cbranch_scc1 " FLAGGED "\n\t" \
        "s_bitset1_b64 %[mst], " EIN "\n\t"         /* a match starts here (encoder.cpp:356) */ \
        "s_bfm_b64 %[tmp], %[inf], " EIN "\n\t"     /* lanes EIN .. EIN+lenA-1 (encoder.cpp:361-362) */ \
        "s_or_b64 %[cov], %[cov], %[tmp]\n\t" \
        "s_bfe_u32 " EOUT ", %[inf], 0x60015\n\t"   /* hop to the next event at or after the match end; SCC = there is one */

// The inner loop of the walk (encoder.cpp:341-368 replayed over ballot masks), hand-written because scalar code
// is slow on this machine (a dependent SALU op ~8 cycles alone and twice that with nine parsers per CU, a taken branch
// ~40: tools/ubench_scalar.hip, tools/ubench_valu.hip).
//   E    : lanes that may start a match          info : per-lane packed walk info (ZZ_WI_*), VGPR
//   pos  : first lane not yet decided            mst  : match-start lanes so far
//   cov  : lanes inside matches so far           usedB: match starts that matched the in-group candidate
// Handles, without leaving the loop: plain matches (length < 16 against the table candidate) and lanes with an
// earlier same-hash lane q in the group (candidate = q if the parse visited it; else the table's, provided no
// third lane shares the hash). Leaves with pos at an event it cannot decide (q skipped and 3+ lanes share the hash, or
// a length of "16 or more" that needs extension); leaves with pos >= 64, or with no event at or after pos, when the
// group is done.
// `nact`: lanes of the group that hold positions; when no event is left, pos comes back as max(pos, nact): the rest are
// literals and the group is done (the caller need not look at E again).
// FIRST: the group's first entry (pos = 0): the first event comes straight from E. Otherwise: re-entry behind an event
// the C++ path decided; pos may lie at or beyond 64 (then there is nothing to do).
// what the walk needs to decide a lane whose candidate is an arbitrary earlier lane of the group (label 7 of the loop)
struct l1_walk_x {
    uint32_t hash, wlo, whi;         // per lane: the hash, the first eight bytes at the position
    uint32_t candbase;               // candidate lane c as a table entry: candbase + c  (= cur + 1 + BIAS + c)
    uint32_t hardok;                 // 0: leave such lanes to the C++ path (groups at a block's end: lengths need clamping)
    uint32_t ovlen, ovcand1;         // per lane, in/out: the overrides of the tokens' length and candidate ...
    uint64_t ovmL, ovmC;             // ... and the lanes that hold one
};
template <bool FIRST>
__device__ __forceinline__ void l1_fast_walk(uint64_t E, uint32_t info, uint32_t nact, uint32_t& pos, uint64_t& mst, uint64_t& cov,
                                             uint64_t& usedB, l1_walk_x& X)
{
    uint64_t tmp, tmp2;
    int32_t e, e2;
    uint32_t inf, len, q, t0, vt;
#define ZZ_L1_WALK_BODY \
        "1:\n\t"                                    /* (pos < 64 here) */ \
        "s_lshl_b64 %[tmp], -1, %[pos]\n\t" \
        "s_and_b64 %[tmp], %[tmp], %[E]\n\t"        /* events at or after pos; SCC = there is one */ \
        "s_cbranch_scc0 30f\n\t" \
        "s_ff1_i32_b64 %[e], %[tmp]\n"              /* the first of them: straight into the hop loop */ \
        "9:\n\t" \
        ZZ_L1_HOP("%[e]", "%[e2]", "4f") "s_cbranch_scc0 31f\n\t" \
        ZZ_L1_HOP("%[e2]", "%[e]", "41f") "s_cbranch_scc1 9b\n\t" \
        "s_and_b32 %[len], %[inf], 31\n\t"          /* no further event: the position behind the last match */ \
        "s_add_u32 %[pos], %[e2], %[len]\n\t" \
        "s_branch 30f\n" \
        "31:\n\t" \
        "s_and_b32 %[len], %[inf], 31\n\t" \
        "s_add_u32 %[pos], %[e], %[len]\n" \
        "30:\n\t" \
        "s_max_u32 %[pos], %[pos], %[nact]\n\t"     /* the rest of the group are literals */ \
        "s_branch 3f\n" \
        "5:\n\t" \
        "s_bitset1_b64 %[mst], %[e]\n\t"            /* (matches found through the in-group candidate logic) */ \
        "s_bfm_b64 %[tmp], %[len], %[e]\n\t" \
        "s_or_b64 %[cov], %[cov], %[tmp]\n\t" \
        "s_add_u32 %[pos], %[e], %[len]\n\t" \
        "s_cmp_lt_u32 %[pos], 64\n\t" \
        "s_cbranch_scc1 1b\n\t" \
        "s_branch 3f\n" \
        "41:\n\t" \
        "s_mov_b32 %[e], %[e2]\n" \
        "4:\n\t" \
        "s_mov_b32 %[pos], %[e]\n\t"                /* every lane below this event is decided */ \
        "s_bitcmp1_b32 %[inf], 17\n\t"              /* not DUP (so EXTA only): leave */ \
        "s_cbranch_scc0 3f\n\t" \
        "s_bfe_u32 %[q], %[inf], 0x6000b\n\t"       /* the nearest earlier lane with my hash */ \
        "s_bitcmp1_b64 %[mst], %[q]\n\t"            /* visited as a match start? */ \
        "s_cbranch_scc1 6f\n\t" \
        "s_bitcmp0_b64 %[cov], %[q]\n\t"            /* visited as a literal? */ \
        "s_cbranch_scc1 6f\n\t" \
        "s_bitcmp1_b32 %[inf], 18\n\t"              /* skipped, and further lanes share the hash (HARD): see 7 */ \
        "s_cbranch_scc1 7f\n" \
        "71:\n\t" \
        "s_bitcmp1_b32 %[inf], 19\n\t"              /* skipped: the table's candidate; EXTA: leave */ \
        "s_cbranch_scc1 3f\n\t" \
        "s_and_b32 %[len], %[inf], 31\n\t" \
        "s_cmp_lt_u32 %[len], 4\n\t" \
        "s_cbranch_scc0 5b\n\t" \
        "s_branch 8f\n" \
        "6:\n\t" \
        "s_bitcmp1_b32 %[inf], 20\n\t"              /* visited: the candidate is lane q (the most recent); EXTB: leave */ \
        "s_cbranch_scc1 3f\n\t" \
        "s_bfe_u32 %[len], %[inf], 0x50006\n\t" \
        "s_cmp_lt_u32 %[len], 4\n\t" \
        "s_cbranch_scc1 8f\n\t" \
        "s_bitset1_b64 %[usedB], %[e]\n\t" \
        "s_branch 5b\n" \
        "8:\n\t" \
        "s_add_u32 %[pos], %[e], 1\n\t"             /* a literal after all (encoder.cpp:367) */ \
        "s_lshl_b64 %[tmp], -2, %[e]\n\t"           /* events above e (none for e = 63: the shift leaves nothing) */ \
        "s_and_b64 %[tmp], %[tmp], %[E]\n\t" \
        "s_cbranch_scc0 30b\n\t" \
        "s_ff1_i32_b64 %[e], %[tmp]\n\t" \
        "s_branch 9b\n" \
        /* The nearest earlier lane with my hash was skipped and the hash has further members: the candidate is the most   */ \
        /* recent VISITED lane with my hash, else the table's (encoder.cpp:344-346: only probed positions are entered).    */ \
        /* Round 2 left the loop here for ~67 instructions of C++; on data with few distinct trigrams (4-symbol: 59 GB/s) */ \
        /* that is most events. Same-hash lanes: one v_cmp against this lane's hash.                                       */ \
        "7:\n\t" \
        "s_cmp_eq_u32 %[hardok], 0\n\t"             /* (a packet's last groups: lengths need clamping: C++) */ \
        "s_cbranch_scc1 3f\n\t" \
        "v_readlane_b32 %[t0], %[hash], %[e]\n\t" \
        "v_cmp_eq_u32_e64 %[tmp], %[t0], %[hash]\n\t"  /* lanes with my hash */ \
        "s_orn2_b64 %[tmp2], %[mst], %[cov]\n\t"    /* visited lanes: match starts and whatever no match covers */ \
        "s_and_b64 %[tmp], %[tmp], %[tmp2]\n\t" \
        "s_bfm_b64 %[tmp2], %[e], 0\n\t"            /* lanes below e */ \
        "s_and_b64 %[tmp], %[tmp], %[tmp2]\n\t" \
        "s_cbranch_scc0 71b\n\t"                    /* none visited: the table's candidate, as for a lane with one earlier member */ \
        "s_flbit_i32_b64 %[q], %[tmp]\n\t" \
        "s_sub_u32 %[q], 63, %[q]\n\t"              /* the most recent of them */ \
        "v_readlane_b32 %[t0], %[wlo], %[e]\n\t" \
        "v_readlane_b32 %[len], %[wlo], %[q]\n\t" \
        "s_xor_b32 %[t0], %[t0], %[len]\n\t" \
        "s_cbranch_scc1 8b\n\t"                     /* fewer than four bytes agree: a literal (encoder.cpp:356) */ \
        "v_readlane_b32 %[t0], %[whi], %[e]\n\t" \
        "v_readlane_b32 %[len], %[whi], %[q]\n\t" \
        "s_xor_b32 %[t0], %[t0], %[len]\n\t" \
        "s_cbranch_scc0 3f\n\t"                     /* eight or more: extension, C++ */ \
        "s_ff1_i32_b32 %[t0], %[t0]\n\t" \
        "s_lshr_b32 %[t0], %[t0], 3\n\t" \
        "s_add_u32 %[len], %[t0], 4\n\t"            /* 4..7 bytes */ \
        "s_lshl_b64 %[tmp2], 1, %[e]\n\t"           /* the token's length and candidate, into the lane that holds it (a select */ \
        "s_or_b32 %[t0], %[len], 0x8000\n\t"        /* under a one-lane mask: v_writelane would need M0 for the lane number)   */ \
        "v_mov_b32 %[vt], %[t0]\n\t" \
        "v_cndmask_b32_e64 %[ovlen], %[ovlen], %[vt], %[tmp2]\n\t" \
        "s_add_u32 %[t0], %[candbase], %[q]\n\t" \
        "v_mov_b32 %[vt], %[t0]\n\t" \
        "v_cndmask_b32_e64 %[ovcand], %[ovcand], %[vt], %[tmp2]\n\t" \
        "s_bitset1_b64 %[ovmL], %[e]\n\t" \
        "s_bitset1_b64 %[ovmC], %[e]\n\t" \
        "s_branch 5b\n" \
        "3:\n\t"
#define ZZ_L1_WALK_OPERANDS \
        : [pos] "+s"(pos), [mst] "+s"(mst), [cov] "+s"(cov), [usedB] "+s"(usedB), [tmp] "=&s"(tmp), [e] "=&s"(e), [e2] "=&s"(e2), \
          [inf] "=&s"(inf), [len] "=&s"(len), [q] "=&s"(q), [tmp2] "=&s"(tmp2), [t0] "=&s"(t0), [vt] "=&v"(vt), \
          [ovlen] "+v"(X.ovlen), [ovcand] "+v"(X.ovcand1), [ovmL] "+s"(X.ovmL), [ovmC] "+s"(X.ovmC) \
        : [E] "s"(E), [info] "v"(info), [nact] "s"(nact), [hash] "v"(X.hash), [wlo] "v"(X.wlo), [whi] "v"(X.whi), \
          [candbase] "s"(X.candbase), [hardok] "s"(X.hardok) \
        : "scc", "vcc"
    if (FIRST)                                            // (pos = 0: the body starts with "the first event at or after pos")
        asm volatile(
            ZZ_L1_WALK_BODY ZZ_L1_WALK_OPERANDS);
    else
        asm volatile(
            "s_cmp_gt_u32 %[pos], 63\n\t"
            "s_cbranch_scc1 3f\n"
            ZZ_L1_WALK_BODY ZZ_L1_WALK_OPERANDS);
#undef ZZ_L1_WALK_BODY
#undef ZZ_L1_WALK_OPERANDS
}

// a committed token as it waits one iteration for emission: bit31 match (len<<16 | dist), bit30 literal (byte)
#define ZZ_TOK_MATCH 0x80000000u
#define ZZ_TOK_LIT 0x40000000u
#define ZZ_TOK_LAST 0x20000000u   // (packet mode, every lane's word) this was the packet's last group

// Fixed-Huffman fragment of one token, by arithmetic and selects only (RFC 1951 3.2.5/3.2.6; the reference's tables
// lcodes_f / dcodes_f / codes_f, fixedhuffmanluts.cpp:5-55, hold the same values): no lane-mask region, i.e. no scalar
// instruction, in the emitter's inner loop.
__device__ __forceinline__ void l1_token_bits(uint32_t tok, uint32_t& bits, uint32_t& nb)
{
    // literal: 0..143 -> 8 bits 00110000+, 144..255 -> 9 bits 110010000+   (codes_f[*sourcePtr], encoder.cpp:367)
    const uint32_t byte = tok & 0xFF;
    const uint32_t lnb = byte < 144 ? 8u : 9u;
    const uint32_t lbits = __builtin_bitreverse32(byte + (byte < 144 ? 0x30u : 0x100u)) >> (32 - lnb);
    // match: merged length code (symbol code, then its extra bits), 5-bit distance code, distance extra bits
    // (lcodes_f[matchLength], encoder.cpp:358; WriteDistance, encoder.cpp:135-141)
    const uint32_t l = (((tok >> 16) & 0x1FF) - 3u) & 0xFFu;             // 0..255 (masked: a literal's word holds no length)
    const bool is258 = l == 255u;
    uint32_t eb = 29u - (uint32_t)__builtin_clz(l | 4u);                 // 0 for l < 8
    uint32_t sym = 257u + 4u * eb + (l >> eb);
    uint32_t ev = l & ((1u << eb) - 1u);
    sym = is258 ? 285u : sym; eb = is258 ? 0u : eb; ev = is258 ? 0u : ev;
    const uint32_t hi = sym >= 280u ? 1u : 0u;                           // 256..279 -> 7 bits, 280..287 -> 8 bits 11000000+
    const uint32_t snb = 7u + hi;
    const uint32_t sbits = __builtin_bitreverse32(sym - 256u + (hi ? 168u : 0u)) >> (32 - snb);
    const uint32_t ll = snb + eb;
    const uint32_t d = ((tok & 0xFFFF) - 1u) & 0x7FFFu;                  // 0..32767 (masked likewise)
    const uint32_t k = 31u - (uint32_t)__builtin_clz(d | 2u);
    const uint32_t deb = k - 1u;                                         // 0 for d < 4
    const uint32_t bucket = d < 2u ? d : 2u * k + ((d >> deb) & 1u);
    const uint32_t dev = d & ((1u << deb) - 1u);
    const uint32_t mbits = sbits | (ev << snb) | ((__builtin_bitreverse32(bucket) >> 27) << ll) | (dev << (ll + 5u));
    const uint32_t mnb = ll + 5u + deb;
    const bool ism = (tok & ZZ_TOK_MATCH) != 0;
    bits = ism ? mbits : lbits;
    nb = ism ? mnb : lnb;
    if (tok == 0) { bits = 0; nb = 0; }
}
__device__ __forceinline__ void l1_emit_tokens(bitring& ring, const uint32_t* lcodes, uint32_t tok)
{
    (void)lcodes;
    uint32_t bits, nb;
    l1_token_bits(tok, bits, nb);
    ring_append(ring, bits, nb);
}

#ifdef ZZ_L1_PIPE_PROBE
__device__ __forceinline__ uint32_t l1_probe_blocks(uint32_t n) { return n < 6 * ZZ_WAVE ? 0u : ((n - 4 * ZZ_WAVE) / ZZ_WAVE) & ~1u; }
#endif
// TT = uint16_t: packet mode, positions < 32768, every candidate is within reach.
// TT = uint32_t: the sequential whole-buffer stream (threaded=false, one block for the whole input): positions up
//      to 2^32, candidates further than 32768 back are ignored (encoder.cpp:348) but stay in the table.
//
// SPLIT = true: the parse and the emission run on two wavefronts of the workgroup. This one (the parser) hands every
// group's tokens to the emitter (l1_emitter) through a two-slot LDS buffer and meets it at one s_barrier per group.
// The emitter's work per group is a fifth of the parser's, so it is always the one waiting (asleep in the barrier,
// not polling: a polling emitter with an LDS queue was measured 6 % slower) and the parser never stalls.
// `tokbuf` is the buffer; `ring` and `lcodes` are not touched.
#define ZZ_L1_TOKSLOT 64u      // 64 tokens; two slots, 512 bytes, 512-byte aligned: slot = byte address ^ 0x100
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ lds_u32* lds_flip_slot(lds_u32* p) { return (lds_u32*)(size_t)((uint32_t)(size_t)p ^ 0x100u); }
__device__ __forceinline__ void l1_group_barrier()
{
    // the LDS traffic of this wave must have landed; global loads (the next group's prefetch) stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// The block is src[start, n): positions are offsets from `src` (table entries carry over from earlier blocks of the
// same stream, encoder.cpp:320-327,370), match lengths stop at n.
//
// BIAS = 32768 (packet mode with a warm window, TT = uint16_t): table entries are position + 1 + BIAS, so that the
// positions -32768 .. -1 in front of the packet -- hashed into the table before the parse starts -- fit beside the
// packet's own; candidates further than 32768 back are ignored, as in the sequential stream.
template <bool SAFE, typename TT, bool SPLIT = false, uint32_t BIAS = 0>
__device__ __forceinline__ void l1_encode_body(const zz_packet_params& P, TT* T, const uint32_t* lcodes,
                                               bitring& ring, const uint8_t* src, const uint8_t* end, uint32_t n,
                                               uint32_t* tokbuf = nullptr, uint32_t start = 0, uint32_t pw = 0)
{
    const int lane = lane_id();
    const uint64_t below_me = (1ull << lane) - 1, above_me = ~((2ull << lane) - 1);
    ZZ_PROF_DECL
    (void)pw;
#ifdef ZZ_L1_PIPE_PROBE
    // TIMING PROBE ONLY (tools/pipe_probe.sh; the output is NOT a valid stream): two parsing wavefronts per packet take
    // alternate blocks of 64 positions, block g + 1 probed and compared while block g is walked -- the optimistic bound of a
    // two-stage pipeline over one packet: no cross-block same-hash resolution, no exchange, no carried match end.
    start += ZZ_WAVE * pw;
#endif
    uint32_t cur = start;
    uint32_t ptok = 0;                                                    // previous group's tokens
    // Packet mode (SPLIT): lanes past the block's end take part in the table accesses and loads like everyone else --
    // they read clamped addresses, hash whatever they read and scribble over table slots, but only in the packet's
    // last group, after which the table is dead; they can never commit, and `left` = 0 keeps them from matching. A lane
    // mask around each of those accesses would cost three scalar instructions apiece in every group. The sequential
    // stream (table carried from block to block) keeps the masks.
    constexpr bool MASKED = !SPLIT;
    uint64_t w = 0, w2 = 0;                                               // 16 bytes at this lane's position
    if (MASKED) { if (start + (uint32_t)lane < n) ld128<SAFE>(src + start + lane, end, w, w2); }
    else ld128<SAFE>(src + (start + (uint32_t)lane < n ? start + (uint32_t)lane : n - 1), end, w, w2);
    lds_u32* slot = SPLIT ? (lds_u32*)tokbuf + lane : nullptr;            // this lane's word of the hand-over slot in use
    // One group of 64 positions. INTERIOR: every lane holds a position with at least 17 bytes after it (all lanes active,
    // no length can run into the block's end) -- all but the last two groups of a packet; the lane-activity compares,
    // the clamps and their scalar bookkeeping drop out of that copy of the code.
    auto group = [&](auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
        const uint32_t nact = INTERIOR ? ZZ_WAVE : ((n - cur) < ZZ_WAVE ? (n - cur) : ZZ_WAVE);
        const uint32_t p = cur + lane;
        const bool active = INTERIOR ? true : lane < (int)nact;
        const bool tact = MASKED ? active : true;                       // takes part in table accesses

        ZZ_T(0);
        // (1) hash, probe + speculative insert; the candidate's bytes are requested at once
        const uint32_t h = calc_hash3((uint32_t)(w >> 8));              // bytes p+1..p+3 (encoder.cpp:344)
        uint32_t oldraw = 0;                                            // what the slot held (restored if I am skipped)
        if (tact) {
            oldraw = T[h];                                              // encoder.cpp:345
            T[h] = (TT)(p + 1 + BIAS);                                  // encoder.cpp:346
        }
        // the candidate, as pos+1 (+BIAS); 0 = none or out of reach (unsigned(distance) <= 32768, encoder.cpp:348)
        const uint32_t old = ((sizeof(TT) == 4 || BIAS) && p + 1 + BIAS - oldraw > 0x8000u) ? 0 : oldraw;
        // every lane loads (lanes without a candidate: the packet's first bytes, result unused) -- no lane mask to set up
        uint64_t wc, wc2;
        if (SPLIT && BIAS == 0)   // (a saturating subtraction and an unsigned offset from the uniform base: two instructions)
            ld128<SAFE>(src + __builtin_elementwise_sub_sat(old, 1u), end, wc, wc2);
        else
            ld128<SAFE>(src + (old ? (int32_t)(old - 1 - BIAS) : (int32_t)start), end, wc, wc2);  // encoder.cpp:350
        // Interior groups: this lane's bytes for the NEXT group are requested now, a whole group ahead of their use, on the
        // guess that this group ends 0..16 positions past its last lane (the overshoot of its last match): 32 bytes from
        // position cur + 64 + lane. Where the guess holds the next group's 16 bytes are cut out of these registers
        // (v_alignbyte, below) and no load sits between two groups any more -- that wait for the L2 was 15 % of a group's
        // time (phase stamps); where it does not, the load happens as before.
        uint64_t sA = 0, sA2 = 0, sB = 0, sB2 = 0;
        if (INTERIOR) {
#ifdef ZZ_L1_PIPE_PROBE
            ld128<SAFE>(src + cur + 2 * ZZ_WAVE + lane, end, sA, sA2);
#else
            ld128<SAFE>(src + cur + ZZ_WAVE + lane, end, sA, sA2);
            ld128<SAFE>(src + cur + ZZ_WAVE + 16 + lane, end, sB, sB2);
#endif
        }
        ZZ_WAVE_SYNC();
        uint32_t rb = 0;
        if (tact) rb = T[h];                                            // the slot holds whichever lane wrote last

        ZZ_T(1);
        // (1a) the previous group's tokens leave while those loads are in flight
        if (!SPLIT && cur != start) l1_emit_tokens(ring, lcodes, ptok);

        ZZ_T(2);
        // (1b) which lanes share a hash inside the group?
        // The read-back names the lane whose store landed: the same lane for every member of a set of equal
        // hashes, a different one for different sets -- a 6-bit key, where the hash has 13 bits. Six ballots give
        // every lane the mask of its set, whatever the number of sets (no loop over them).
        const uint64_t lostmask = ballot(active && rb != (uint32_t)(TT)(p + 1 + BIAS));
        // (2) lengths against both possible candidates, 16 bytes compared ("16 or more" = 17 where more bytes remain)
        const uint32_t left = active ? n - p : 0;                       // bytes left in the block (D1 clamp)
        // cap17: 8 x min(bytes left, 17)
        const uint32_t cap17 = INTERIOR ? 8u * (ZZ_WI_CAP + 1) : (left < ZZ_WI_CAP + 1 ? left : ZZ_WI_CAP + 1) << 3;   // (0 for lanes past the end)
        uint64_t myset = 0;        // per lane: all lanes of the group sharing my hash (itself included)
        uint32_t info = 0;
        if (lostmask) {            // some lane's store was overwritten: at least one hash occurs twice
            uint32_t W = (uint32_t)lane;
            if (active) W = (rb - 1u - BIAS - cur) & 63u;
            myset = wave_match6(W);
            // (selects, not a lane-mask region: a lane alone with its hash has nothing below it and ends up with info = 0)
            const uint64_t below = myset & below_me;                      // earlier lanes with my hash
            const bool dup = below != 0 && active;                        // (a lane past the end may sit in a set: never an event)
            const uint32_t ql = 63u - (uint32_t)__builtin_clzll(below | 1ull);   // (lane 0 where there is none: result unused)
            // the in-group candidate's bytes come from its lane's registers, while the table candidate's are still in flight
            const int qa = (int)(ql << 2);
            const uint64_t wq = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)(w >> 32)) << 32) | (uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)w);
            const uint64_t wq2 = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)(w2 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)w2);
            const uint32_t lb = equal_bits128(w ^ wq, w2 ^ wq2, cap17) >> 3;
            const uint32_t di = ZZ_WI_DUP | (ql << ZZ_WI_QLANE_SHIFT) | ((uint32_t)__builtin_popcountll(below) > 1u ? ZZ_WI_HARD : 0u)   // two or more earlier lanes share my hash: which of them the parse visited decides
                                | (lb > ZZ_WI_CAP ? ((ZZ_WI_CAP << ZZ_WI_LENB_SHIFT) | ZZ_WI_EXTB) : (lb << ZZ_WI_LENB_SHIFT));
            info = dup ? di : 0u;
        }
        ZZ_T(3);
        ZZ_DRAIN();
        ZZ_T(4);
        const uint64_t x = w ^ wc;                                      // (only looked at where there is a candidate)
        // second half of the previous group's hand-over. The barrier sits behind the wait for the candidate bytes (the operand
        // ties it there): the emitter is released where this wave has just been waiting anyway, +0.4 % over a barrier at the
        // top of the group (profiles/README.md, round 3)
#ifndef ZZ_L1_PIPE_PROBE
        if (SPLIT) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" :: "v"((uint32_t)x) : "memory");
#endif
        uint32_t la = equal_bits128(x, w2 ^ wc2, cap17) >> 3;
        if (!old) la = 0;
        info |= la > ZZ_WI_CAP ? (ZZ_WI_CAP | ZZ_WI_EXTA) : la;
        // events: lanes that can start a match under some parse; "simple" ones need no look at the parse
        // (HARD, or lenA >= 4, or lenB >= 4; one compare, because a ballot of anything else costs two more instructions)
        // (lanes past the end have info = 0: no lengths because nothing is left, no flags by the line above)
        const uint64_t E = ballot((info & (ZZ_WI_HARD | 0x1Cu | (0x1Cu << ZZ_WI_LENB_SHIFT))) != 0);
        // a plain match knows where the walk continues: the first event at or after its end (0 = none). With that
        // in the info word the scalar loop hops from match to match without re-scanning the event mask.
        {
            const uint32_t endl = (uint32_t)lane + la;                  // (la = 17 on EXTA lanes, whose NEXT nobody reads)
            const uint64_t m = E >> (endl & 63u);                       // (endl >= 64: whatever this finds lies at 64 or beyond)
            const uint32_t nx = endl + (m ? (uint32_t)__builtin_ctzll(m) : 64u);
            info |= ((nx < 64u ? nx : 64u) & 63u) << ZZ_WI_NEXT_SHIFT;   // (64 & 63 = 0 = none: no compare-and-select)
        }

        ZZ_T(5); ZZ_C(10, 1);
        ZZ_C(11 + 4, (uint32_t)__builtin_popcountll(ballot((info & ZZ_WI_DUP) != 0) & E));   // (diagnostic) event lanes with an in-group candidate
        ZZ_C(11 + 1, (uint32_t)__builtin_popcountll(E));                                        // (diagnostic) event lanes
        // (3) the walk: replays the reference's decisions in order (encoder.cpp:341-368). Scalar code is slow
        // on this machine (a dependent SALU op ~8 cycles, a taken branch ~40: tools/ubench_scalar.hip), so runs
        // of "simple" matches go through a hand-written 12-instruction loop; everything else drops out to C++.
        uint64_t mst = 0, cov = 0, usedB = 0;    // match-start lanes / lanes covered by matches / matched the in-group candidate
        l1_walk_x X;                             // per-lane overrides written for extended / hard events (length | 0x8000), the lanes that hold one
        X.hash = h; X.wlo = (uint32_t)w; X.whi = (uint32_t)(w >> 32); X.candbase = cur + 1 + BIAS; X.hardok = INTERIOR ? 1u : 0u;
        X.ovlen = 0; X.ovcand1 = 0; X.ovmL = 0; X.ovmC = 0;
        uint32_t& ovlen = X.ovlen; uint32_t& ovcand1 = X.ovcand1;
        uint64_t& ovmL = X.ovmL; uint64_t& ovmC = X.ovmC;
        uint32_t pos = 0;
#ifdef ZZ_L1_PIPE_PROBE
        if (SPLIT) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" :: "v"(info) : "memory");     // the block in front has been walked
#endif
        l1_fast_walk<true>(E, info, nact, pos, mst, cov, usedB, X);
        while (pos < nact) {
            const int e = (int)pos;                                     // the walk stopped AT an event it cannot decide
            ZZ_C(11, 1);
            const uint64_t probed = ~cov | mst;                         // lanes below e the parse has visited
            const uint32_t inf = readlane(info, e);
            ZZ_C(13, (inf & ZZ_WI_HARD) ? 1 : 0); ZZ_C(14, (!(inf & ZZ_WI_DUP) && (inf & ZZ_WI_EXTA)) ? 1 : 0);
            const uint32_t pe = cur + (uint32_t)e;
            const uint32_t maxlen = (n - pe) < ZZ_MAX_LEN ? (n - pe) : ZZ_MAX_LEN;
            uint32_t mlen;
            if (!(inf & ZZ_WI_HARD)) {
                const bool useB = (inf & ZZ_WI_DUP) && ((probed >> ZZ_WI_QLANE(inf)) & 1);
                mlen = useB ? ZZ_WI_LENB(inf) : ZZ_WI_LENA(inf);
                if (mlen >= 4) {
                    if (inf & (useB ? ZZ_WI_EXTB : ZZ_WI_EXTA)) {     // remain(), encoder.cpp:64-90
                        const int32_t cand = useB ? (int32_t)(cur + ZZ_WI_QLANE(inf)) : (int32_t)(readlane(old, e) - 1 - BIAS);
                        mlen = wave_extend_match<SAFE>(src, pe, cand, maxlen, end, ZZ_WI_CAP);
                        if (lane == e) ovlen = mlen | 0x8000u;
                        ovmL |= 1ull << e;
                    }
                    if (useB) usedB |= 1ull << e;
                }
            } else {
                // hash shared by 3+ lanes: candidate = most recent visited lane with my hash, else the table's
                const uint64_t S = readlane64(myset, e) & probed & ((1ull << e) - 1);
                uint32_t cand1 = 0;
                uint64_t xe = ~0ull;
                if (S) {
                    const int c = 63 - __builtin_clzll(S);
                    cand1 = cur + (uint32_t)c + 1 + BIAS;
                    xe = readlane64(w, e) ^ readlane64(w, c);
                } else {
                    cand1 = readlane(old, e);
                    if (cand1) xe = readlane64(x, e);
                }
                mlen = 0;
                if ((uint32_t)xe == 0 && maxlen >= 4) {
                    if (xe != 0) mlen = (uint32_t)__builtin_ctzll(xe) >> 3;
                    else mlen = wave_extend_match<SAFE>(src, pe, (int32_t)(cand1 - 1 - BIAS), maxlen, end);
                    if (mlen > maxlen) mlen = maxlen;
                }
                if (lane == e) { ovlen = mlen | 0x8000u; ovcand1 = cand1; }
                ovmL |= 1ull << e; ovmC |= 1ull << e;
            }
            if (mlen > 3) {                                              // encoder.cpp:356
                mst |= 1ull << e;
                cov |= (mlen >= 64u - (uint32_t)e) ? (~0ull << e) : (((1ull << mlen) - 1) << e);
                pos = (uint32_t)e + mlen;                                // encoder.cpp:361-362
            } else {
                pos = (uint32_t)e + 1;                                   // a literal after all (encoder.cpp:367)
            }
            l1_fast_walk<false>(E, info, nact, pos, mst, cov, usedB, X);
        }
        ZZ_T(6);
#ifdef ZZ_L1_PIPE_PROBE
        if (SPLIT) asm volatile("s_barrier" :: "s"(pos) : "memory");                                   // this block has been walked
#endif
        // visited lanes: every lane in front of `pos` that no match covers, plus the match starts
        // (the walk leaves 1 <= pos, and pos <= nact wherever nact < 64: every lane below pos is an active one)
        uint64_t committed;
        if (SPLIT) {    // five scalar instructions (left to itself the compiler does this 64-bit arithmetic on the VALU)
            uint32_t t;
            asm("s_min_u32 %1, %2, 64\n\ts_sub_u32 %1, 64, %1\n\ts_lshr_b64 %0, -1, %1\n\ts_andn2_b64 %0, %0, %3\n\ts_or_b64 %0, %0, %4"
                : "=&s"(committed), "=&s"(t) : "s"(pos), "s"(cov), "s"(mst) : "scc");
        } else {
            committed = (~cov & (~0ull >> (64u - (pos < 64u ? pos : 64u)))) | mst;
        }
#ifdef ZZ_L1_PIPE_PROBE
        const uint32_t next = SPLIT ? cur + 2 * ZZ_WAVE : cur + pos;
#else
        const uint32_t next = cur + pos;
#endif
        // next group's bytes: in flight while this group is repaired
        uint64_t wnext = 0, wnext2 = 0;
        if (MASKED) { if (next + (uint32_t)lane < n) ld128<SAFE>(src + next + lane, end, wnext, wnext2); }
#ifdef ZZ_L1_PIPE_PROBE
        else if (INTERIOR) { wnext = sA; wnext2 = sA2; }
#endif
        else if (INTERIOR && pos <= ZZ_WAVE + 16) {
            // the guess held: bytes d .. d+15 of the 32 requested at the top of the group (d = pos - 64, uniform)
            const uint32_t d = pos - ZZ_WAVE, sh = d & 3u;
            const uint32_t b0 = (uint32_t)sA, b1 = (uint32_t)(sA >> 32), b2 = (uint32_t)sA2, b3 = (uint32_t)(sA2 >> 32);
            const uint32_t b4 = (uint32_t)sB, b5 = (uint32_t)(sB >> 32), b6 = (uint32_t)sB2, b7 = (uint32_t)(sB2 >> 32);
            uint32_t r0, r1, r2, r3;
            // Dwords d/4 .. d/4+4 of the eight, shifted by d%4 bytes. d is uniform, so this is a jump to one of five blocks
            // of four v_alignbyte_b32 -- written out, because the compiler's lowering of the switch cost about twenty scalar
            // instructions of flow bookkeeping per group; the compare chain is ordered by likelihood (a short overshoot first).
            asm("s_cmp_lt_u32 %[k], 1\n\ts_cbranch_scc1 10f\n\t"
                "s_cmp_lt_u32 %[k], 2\n\ts_cbranch_scc1 11f\n\t"
                "s_cmp_lt_u32 %[k], 3\n\ts_cbranch_scc1 12f\n\t"
                "s_cmp_lt_u32 %[k], 4\n\ts_cbranch_scc1 13f\n\t"
                "v_mov_b32 %[r0], %[b4]\n\tv_mov_b32 %[r1], %[b5]\n\tv_mov_b32 %[r2], %[b6]\n\tv_mov_b32 %[r3], %[b7]\n\t"      // d = 16
                "s_branch 19f\n"
                "13:\n\t"
                "v_alignbyte_b32 %[r0], %[b4], %[b3], %[sh]\n\tv_alignbyte_b32 %[r1], %[b5], %[b4], %[sh]\n\t"
                "v_alignbyte_b32 %[r2], %[b6], %[b5], %[sh]\n\tv_alignbyte_b32 %[r3], %[b7], %[b6], %[sh]\n\t"
                "s_branch 19f\n"
                "12:\n\t"
                "v_alignbyte_b32 %[r0], %[b3], %[b2], %[sh]\n\tv_alignbyte_b32 %[r1], %[b4], %[b3], %[sh]\n\t"
                "v_alignbyte_b32 %[r2], %[b5], %[b4], %[sh]\n\tv_alignbyte_b32 %[r3], %[b6], %[b5], %[sh]\n\t"
                "s_branch 19f\n"
                "11:\n\t"
                "v_alignbyte_b32 %[r0], %[b2], %[b1], %[sh]\n\tv_alignbyte_b32 %[r1], %[b3], %[b2], %[sh]\n\t"
                "v_alignbyte_b32 %[r2], %[b4], %[b3], %[sh]\n\tv_alignbyte_b32 %[r3], %[b5], %[b4], %[sh]\n\t"
                "s_branch 19f\n"
                "10:\n\t"
                "v_alignbyte_b32 %[r0], %[b1], %[b0], %[sh]\n\tv_alignbyte_b32 %[r1], %[b2], %[b1], %[sh]\n\t"
                "v_alignbyte_b32 %[r2], %[b3], %[b2], %[sh]\n\tv_alignbyte_b32 %[r3], %[b4], %[b3], %[sh]\n"
                "19:"
                : [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3)
                : [k] "s"(d >> 2), [sh] "s"(sh), [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [b3] "v"(b3), [b4] "v"(b4), [b5] "v"(b5),
                  [b6] "v"(b6), [b7] "v"(b7)
                : "scc");
            wnext = ((uint64_t)r1 << 32) | r0;
            wnext2 = ((uint64_t)r3 << 32) | r2;
        }
        else if (INTERIOR || next < n)      // (interior copy: no branch; a finished packet reads its last byte once more)
            ld128<SAFE>(src + (next + (uint32_t)lane < n ? next + (uint32_t)lane : n - 1), end, wnext, wnext2);

        // (4) table repair: skipped lanes restore the old entry; among committed lanes sharing a hash the
        // highest position wins -- the state the serial loop leaves behind
        const bool is_committed = (committed >> lane) & 1;
        ZZ_WAVE_SYNC();
        if (SPLIT) {    // the scalar mask is the lane mask of the store: no compare, no branch
            uint64_t saved;
            asm volatile("s_andn1_saveexec_b64 %0, %1\n\tds_write_b16 %2, %3\n\ts_mov_b64 exec, %0"
                         : "=&s"(saved) : "s"(committed), "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) TT*)(T + h)), "v"(oldraw)
                         : "memory", "scc");
        } else if (tact && !is_committed) T[h] = (TT)oldraw;
        if (lostmask) {
            ZZ_WAVE_SYNC();
            if (SPLIT) {
                const uint64_t wm = ballot((myset & committed & above_me) == 0) & committed;
                uint64_t saved;
                asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write_b16 %2, %3\n\ts_mov_b64 exec, %0"
                             : "=&s"(saved) : "s"(wm), "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) TT*)(T + h)), "v"(p + 1 + BIAS)
                             : "memory", "scc");
            } else {
                const bool winner = is_committed && (myset & committed & above_me) == 0;
                if (winner) T[h] = (TT)(p + 1 + BIAS);
            }
        }
        ZZ_WAVE_SYNC();

        // (5) this group's tokens, emitted by the next iteration
        if (SPLIT) {
            // the walk's scalar masks are the lane predicates of the selects (sel_lanes): no shifting by the lane id
            const uint32_t la_ = ZZ_WI_LENA(info) | 0x8000u, lb_ = ZZ_WI_LENB(info) | 0x8000u;   // ZZ_TOK_MATCH >> 16 rides along
            const uint32_t tl = sel_lanes(ovmL, ovlen, sel_lanes(usedB, lb_, la_));
            const uint32_t cn = sel_lanes(ovmC, ovcand1, sel_lanes(usedB, cur + ZZ_WI_QLANE(info) + 1 + BIAS, old));
            const uint32_t tmatch = (tl << 16) | (p + 1 + BIAS - cn);
            const uint32_t tlit = ZZ_TOK_LIT | (uint32_t)(w & 0xFF);
            ptok = keep_lanes(committed, sel_lanes(mst, tmatch, tlit));
            if (!INTERIOR) ptok |= next >= n ? ZZ_TOK_LAST : 0u;          // (a packet that ends in an interior group: see below)
        } else {   // (masks, not branches: the values are cheap and a lane-mask region is three scalar instructions)
            const uint32_t mB = 0u - (uint32_t)((usedB >> lane) & 1);          // all ones where the in-group candidate matched
            const uint32_t la_ = ZZ_WI_LENA(info), ca_ = old;
            const uint32_t tl = la_ ^ ((la_ ^ ZZ_WI_LENB(info)) & mB);
            const uint32_t cn = ca_ ^ ((ca_ ^ (cur + ZZ_WI_QLANE(info) + 1 + BIAS)) & mB);
            const uint32_t mL = 0u - (uint32_t)((ovmL >> lane) & 1), mC = 0u - (uint32_t)((ovmC >> lane) & 1);
            const uint32_t tlen = tl ^ ((tl ^ (ovlen & 0x1FFu)) & mL);
            const uint32_t cand1 = cn ^ ((cn ^ ovcand1) & mC);
            const uint32_t tmatch = ZZ_TOK_MATCH | (tlen << 16) | (p + 1 + BIAS - cand1);
            const uint32_t tlit = ZZ_TOK_LIT | (uint32_t)(w & 0xFF);
            const uint32_t mM = 0u - (uint32_t)((mst >> lane) & 1);
            ptok = (tlit ^ ((tlit ^ tmatch) & mM)) & (0u - (uint32_t)is_committed);
        }
        cur = next;
        if (SPLIT) {
            // hand-over, first half: the tokens go to the slot now; the barrier that releases them to the emitter
            // sits in the next trip, behind the wait for the table read that trip needs anyway
            *slot = ptok;
#ifndef ZZ_L1_PIPE_PROBE
            slot = lds_flip_slot(slot);
#endif
        }
        if (sizeof(TT) == 4 && ring.flushed >= (1u << 24)) {
            // long streams: slide the ring's origin (by a multiple of the ring size, so slots keep their meaning)
            const uint32_t kw = ring.flushed & ~(uint32_t)(ZZ_RING_WORDS - 1);
            ring.out32 += kw; ring.bitpos -= kw * 32; ring.flushed -= kw;
        }
        ZZ_T(7);
        ZZ_DRAIN();
        w = wnext;
        w2 = wnext2;
        ZZ_T(8);
    };
    // (lane 63 has 17 bytes left, and the 32 bytes requested for the next group lie inside the block: cur + 64 + 63 + 32 <= n)
#ifdef ZZ_L1_PIPE_PROBE
    if (SPLIT) {
        // the same number of barriers in every wavefront of the workgroup: nb blocks (even), two barriers per block and one
        // of offset: parser 1 starts a barrier late, parser 0 ends a barrier late; the rest of the packet is dropped
        const uint32_t nb = l1_probe_blocks(n);
        slot = (lds_u32*)tokbuf + lane + pw * ZZ_L1_TOKSLOT;
        if (pw == 1) asm volatile("s_barrier" ::: "memory");
        for (uint32_t b = pw; b < nb; b += 2) group(std::true_type{});
        if (pw == 0) asm volatile("s_barrier" ::: "memory");
        return;
    }
#endif
    if (SPLIT) while (cur + 2 * ZZ_WAVE + 32 <= n) group(std::true_type{});
    const bool flagged = cur < n;                                       // the last group will carry ZZ_TOK_LAST
    while (cur < n) group(std::false_type{});
    if (!SPLIT) l1_emit_tokens(ring, lcodes, ptok);
    else {
        if (!flagged) {         // a long match ended the packet inside an interior group: an empty group carries the flag
            l1_group_barrier();
            *slot = ZZ_TOK_LAST;
        }
        l1_group_barrier();                                             // the last group's hand-over
    }
    ZZ_PROF_FLUSH(P);
}

// The other half of SPLIT: Huffman-codes and appends group after group (fixed codes computed, no table in LDS).
// It reads slot g & 1 into registers right after barrier g; the parser overwrites that slot only after barrier
// g + 1, which this wave reaches after the read.
__device__ __forceinline__ void l1_emitter(bitring& ring, const uint32_t* tokbuf, uint32_t n = 0)
{
    const lds_u32* slot = (const lds_u32*)tokbuf + lane_id();
    (void)n;
#ifdef ZZ_L1_PIPE_PROBE
    for (uint32_t b = 0, nb = l1_probe_blocks(n) + 1; b < nb; ++b) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const uint32_t tok = *slot;
        slot = lds_flip_slot((lds_u32*)slot);
        l1_emit_tokens(ring, nullptr, tok & ~ZZ_TOK_LAST);
    }
    return;
#endif
    l1_group_barrier();                                                  // the parser's barrier in front of its first group
    for (;;) {
        l1_group_barrier();
        const uint32_t tok = *slot;
        slot = lds_flip_slot((lds_u32*)slot);
        l1_emit_tokens(ring, nullptr, tok & ~ZZ_TOK_LAST);
        if (uniform(tok) & ZZ_TOK_LAST) break;                           // (lane 0 of a group always holds a token)
    }
}

// Two wavefronts per packet: wave 0 parses (hash table, match search, the serial walk), wave 1 computes the Adler-32
// and turns the parser's tokens into the bit stream; they meet at one s_barrier per group of 64 positions (see
// l1_encode_body). Both execute exactly one barrier per group, so the counts always match.
#ifdef ZZ_L1_PIPE_PROBE
#define ZZ_L1_THREADS (3 * ZZ_WAVE)
#else
#define ZZ_L1_THREADS (2 * ZZ_WAVE)
#endif
// what a packet is, for both wavefronts
struct l1_pk {
    const uint8_t* src; const uint8_t* end; uint8_t* out;
    uint64_t off; uint32_t len, n; bool is_final;
};
__device__ __forceinline__ l1_pk l1_packet_of(const zz_packet_params& P, uint32_t k)
{
    l1_pk q;
    q.off = (uint64_t)k * P.packet_size;
    q.len = (uint32_t)((P.n - q.off) < P.packet_size ? (P.n - q.off) : P.packet_size);
    q.is_final = P.last_is_final && k == P.npk - 1;
    q.n = q.is_final ? q.len : q.len - 1;          // bytes of the compressing AddData (zzflate.cpp:113,116)
    q.src = P.src + q.off;
    q.end = P.src + P.n;                           // one past the last readable byte
    q.out = P.slots + (uint64_t)k * P.slot_stride;
    return q;
}
// ---- parser: cold table (encoder.cpp:533-536), then the block body ------------------------------------------------
// BIAS = 32768: warm window. The reference's threaded mode starts every range with a cold table and
// so loses the matches that reach back into the previous range (zzflate.cpp:101-125); its single Encoder carries the
// table across blocks instead (FixHashTable, encoder.cpp:320-327). Here the last P.warm bytes in front of the packet are
// hashed into the table before the parse starts -- every position, per hash the highest -- so the packet may refer
// to them (one DEFLATE stream: the window spans packet boundaries). Packets stay independent of each other's parses,
// so nothing is serialised; the stream is not the reference's threaded stream any more (a separate flag), it is
// pinned by the oracle's restatement of this very rule and by inflate.
// Warm window: every position of the W bytes in front of the packet goes into the (empty) table under its key -- the three
// bytes at position + keyoff --, and per hash the HIGHEST position stays. Eight groups of 64 positions per trip, the next trip's
// loads requested before this trip's are used (the window comes from the Infinity Cache, ~2 us).
// MASKFREE: a lane without a position stores to a spare slot instead of sitting out under a lane mask (every masked region is
// three scalar instructions); level 2's kernel has no registers to spare for the address selects and keeps the masks.
// `viol` (in/out, per lane): set where the LDS did NOT keep that order -- checked on the last store of every trip (one read-back
// per 512 positions: the slot must hold a position at or above the lane's own; a sampled run-time monitor behind the per-device
// probe, zz_api.hip lds_order_ok). The caller reports it (ZZ_ERR_LDS_ORDER) and the host refuses the stream.
// Level 2's kernel has no register for `viol` to live in (96 VGPRs, the limit of its five wavefronts per SIMD: the check spilled
// into the block loops), so it checks the same property of the same LDS on a canary instead: all 64 lanes store lane + 1 to
// the spare slot, which must then read 64 (MASKFREE = false; once per packet, behind the last trip).
template <uint32_t BIAS, bool MASKFREE>
__device__ __forceinline__ void warm_prehash(uint16_t* T, const uint8_t* src, int32_t W, const uint8_t* end, int keyoff, uint32_t spare, uint32_t& viol)
{
    const int lane = lane_id();
    auto request = [&](int32_t g, uint32_t (&w4)[8]) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int32_t pos = g + u * ZZ_WAVE + lane;
            w4[u] = (g < 0 && pos < 0) ? load32_safe(src + pos + keyoff, end) : 0u;
        }
    };
    // The LDS serves the lanes of one store that hit one address in ascending lane order, and one wavefront's stores in issue
    // order (zz_debug_lds_atomic_order, under a GPU test): storing the positions block after block, ascending, leaves the
    // highest position of every hash in place -- no read-back, no retry rounds (round 3's earlier form: 104.1 GB/s at 32 KiB).
    auto enter = [&](int32_t g, const uint32_t (&w4)[8]) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int32_t pos = g + u * ZZ_WAVE + lane;
            const uint32_t hh = calc_hash3(w4[u]);
            if (MASKFREE) T[pos < 0 ? hh : spare] = (uint16_t)(pos + 1 + (int32_t)BIAS);
            else if (pos < 0) T[hh] = (uint16_t)(pos + 1 + (int32_t)BIAS);
            if (MASKFREE && u == 7) {
                ZZ_WAVE_SYNC();
                const uint32_t mine = (uint32_t)(uint16_t)(pos + 1 + (int32_t)BIAS);
                if ((uint32_t)T[pos < 0 ? hh : spare] < mine && pos < 0) viol = 1u;
            }
        }
    };
    if (MASKFREE) {
        // (a trip is eight stores now: three trips' loads in flight to cover the window's way from the Infinity Cache)
        uint32_t wa[8], wb[8], wc[8];
        request(-W, wa);
        request(-W + 8 * ZZ_WAVE, wb);
        for (int32_t g = -W; g < 0; g += 24 * ZZ_WAVE) {
            request(g + 16 * ZZ_WAVE, wc);
            enter(g, wa);
            if (g + 8 * ZZ_WAVE >= 0) break;
            request(g + 24 * ZZ_WAVE, wa);
            enter(g + 8 * ZZ_WAVE, wb);
            if (g + 16 * ZZ_WAVE >= 0) break;
            request(g + 32 * ZZ_WAVE, wb);
            enter(g + 16 * ZZ_WAVE, wc);
        }
        return;
    }
    uint32_t wa[8], wb[8];
    request(-W, wa);
    for (int32_t g = -W; g < 0; g += 16 * ZZ_WAVE) {
        request(g + 8 * ZZ_WAVE, wb);
        enter(g, wa);
        if (g + 8 * ZZ_WAVE >= 0) break;
        request(g + 16 * ZZ_WAVE, wa);
        enter(g + 8 * ZZ_WAVE, wb);
    }
    ZZ_WAVE_SYNC();
    T[spare] = (uint16_t)(lane + 1);                     // the canary: 64 lanes, one address
    ZZ_WAVE_SYNC();
    viol = (uint32_t)T[spare] != (uint32_t)ZZ_WAVE ? 1u : 0u;
}

template <uint32_t BIAS>
__device__ __forceinline__ void l1_packet_parser(const zz_packet_params& P, uint32_t k, uint16_t* T, uint32_t* tokbuf, uint32_t pw = 0)
{
    const int lane = lane_id();
    const l1_pk q = l1_packet_of(P, k);
    // the parse is the critical path of the packet, the emitter has slack: the parsing wave goes first in the issue
    // arbiter (+4.8 % on 1 GiB of text)
    __builtin_amdgcn_s_setprio(3);
    uint4* t4 = (uint4*)T;
    for (int i = lane; i < (int)(ZZ_HASH_SIZE * sizeof(uint16_t) / 16); i += ZZ_WAVE) t4[i] = make_uint4(0, 0, 0, 0);
    ZZ_WAVE_SYNC();
    (void)pw;
    if (BIAS) {
        const uint64_t before = P.halo + q.off;                       // input bytes of this stream in front of the packet
        // key of a position: bytes pos+1..pos+3 (encoder.cpp:344); spare slot: the last half word of the hand-over slots, unused so far
        uint32_t viol = 0;
        warm_prehash<BIAS, true>(T, q.src, (int32_t)(before < P.warm ? before : P.warm), q.end, 1, (uint32_t)(((uint16_t*)tokbuf + 255) - T), viol);
        if ((ballot(viol != 0) != 0 || P.dbg_viol) && lane == 0) atomicOr(P.err, ZZ_ERR_LDS_ORDER);
    }
    bitring none;
    none.ring = nullptr; none.out32 = nullptr; none.bitpos = 0; none.flushed = 0;
    if (q.n > 0) {
        // 16-byte loads may run up to 15 bytes past the packet's last byte: bounds-checked loads wherever that
        // would leave the shard (decided by bytes, not by packet index: packets may be as short as one byte)
        if (q.off + q.len + 16 > P.n) l1_encode_body<true, uint16_t, true, BIAS>(P, T, nullptr, none, q.src, q.end, q.n, tokbuf, 0, pw);
        else l1_encode_body<false, uint16_t, true, BIAS>(P, T, nullptr, none, q.src, q.end, q.n, tokbuf, 0, pw);
    }
}
// ---- emitter ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void l1_packet_emitter(const zz_packet_params& P, uint32_t k, uint32_t* ring_words, const uint32_t* tokbuf)
{
    const int lane = lane_id();
    const l1_pk q = l1_packet_of(P, k);
#ifdef ZZ_L1_EMIT_PRIO
    __builtin_amdgcn_s_setprio(ZZ_L1_EMIT_PRIO);
#endif
    bitring ring;
    ring_init(ring, ring_words, q.out);
    if (P.cks_kind == ZZ_CKS_ADLER) {     // while the parser works on its first groups
        zz_cks c = wave_adler(q.src, q.len);
        if (lane == 0) P.cks[k] = c;
    }
    if (q.n > 0) {
        // StartBlock(FixedHuffman, final): encoder.cpp:143-147,338
        ring_append_uniform(ring, (q.is_final ? 1u : 0u) | (1u << 1), 3);
        l1_emitter(ring, tokbuf, q.n);
        // EOB: codes_f[256] = 7 zero bits (encoder.cpp:371)
        ring_append_uniform(ring, 0, 7);
    }
    if (!q.is_final) {
        // SetLevel(0); AddData(e-1, e): one stored byte = byte alignment (zzflate.cpp:118-120,
        // encoder.cpp:482-502): BFINAL=0 BTYPE=00, pad, LEN=1, NLEN=0xFFFE, the byte
        ring_append_uniform(ring, 0, 3);
        ring_pad_to_byte(ring);
        ring_append_uniform(ring, 0xFFFE0001u, 32);
        ring_append_uniform(ring, q.src[q.len - 1], 8);
    } else if (q.n == 0) {
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

// One packet per workgroup, in index order: the hardware dispatcher hands the next packet to whichever CU has room.
// (Measured and rejected, profiles/README.md: persistent workgroups -- values invariant across packets get hoisted out
// of the packet loop and live through the group loop, 95 VGPRs instead of 46 --; a second kind of workgroup with its
// hash table in global memory to fill the wave slots the LDS limit leaves free -- 4x slower per packet: every table
// access moves a 64-byte sector for two bytes.)
__global__ __launch_bounds__(ZZ_L1_THREADS) void k_encode_l1(zz_packet_params P)
{
    // table + ring + token slots = 17,408 bytes <= 17,920 = 35 LDS granules: NINE workgroups share a CU
    __shared__ uint16_t T[ZZ_HASH_SIZE];          // hashtable (encoder.h:76) as pos+1, 0 = empty
    __shared__ uint32_t ring_words[ZZ_RING_WORDS];
    __shared__ __attribute__((aligned(512))) uint32_t tokbuf[2 * ZZ_L1_TOKSLOT];
    const uint32_t k = blockIdx.x;
#ifdef ZZ_L1_PIPE_PROBE
    if (uniform(threadIdx.x >> 6) < 2) l1_packet_parser<0>(P, k, T, tokbuf, uniform(threadIdx.x >> 6));
    else l1_packet_emitter(P, k, ring_words, tokbuf);
    return;
#endif
    if (uniform(threadIdx.x >> 6) == 0) l1_packet_parser<0>(P, k, T, tokbuf);
    else l1_packet_emitter(P, k, ring_words, tokbuf);
}
// the same with a warm window (P.warm > 0)
__global__ __launch_bounds__(ZZ_L1_THREADS) void k_encode_l1w(zz_packet_params P)
{
    __shared__ uint16_t T[ZZ_HASH_SIZE];          // position + 1 + 32768; 0 = empty
    __shared__ uint32_t ring_words[ZZ_RING_WORDS];
    __shared__ __attribute__((aligned(512))) uint32_t tokbuf[2 * ZZ_L1_TOKSLOT];
    const uint32_t k = blockIdx.x;
#ifdef ZZ_L1_PIPE_PROBE
    return;                                        // (the probe build has three wavefronts per workgroup: cold packets only)
#endif
    if (uniform(threadIdx.x >> 6) == 0) l1_packet_parser<32768u>(P, k, T, tokbuf);
    else l1_packet_emitter(P, k, ring_words, tokbuf);
}

// The sequential whole-buffer stream of the reference (threaded=false: zzflate.cpp:84-95, one Encoder over the
// whole input) at level 1: AddData (encoder.cpp:539-552) calls WriteBlockFixedHuff until the input is used up, and every
// call encodes as many bytes as are certain to fit the output buffer at nine bits each (encoder.cpp:331-337). With a
// roomy caller-owned buffer that is ONE block for the whole input; with a tight one (zztest/Test.cpp:206-212,255-258
// pass dest = input size) or the callback API's 1,000,000-byte chunks it is a chain of blocks whose lengths depend on
// how well the earlier ones compressed. The hash table carries over (FixHashTable, encoder.cpp:320-327,370: positions
// here are offsets from the stream's first byte, so nothing needs rebasing), matches stop at the block end. Inherently
// serial: one wavefront, 32-bit table entries. A compatibility mode, not a throughput mode (threaded=true is that).
// Output goes to slot 0; sizes[0] gets its length (< 4 GiB: the host admits inputs below 2 GiB, as the reference's int lengths do).
__global__ __launch_bounds__(ZZ_WAVE) void k_stream_l1(zz_packet_params P, zz_stream_ctl C)
{
    __shared__ uint32_t T[ZZ_HASH_SIZE];          // absolute position + 1, 0 = empty
    __shared__ uint32_t ring_words[ZZ_RING_WORDS];
    const int lane = lane_id();
    {
        uint4* t4 = (uint4*)T;
        for (int i = lane; i < (int)(sizeof(T) / 16); i += ZZ_WAVE) t4[i] = make_uint4(0, 0, 0, 0);
    }
    bitring ring;
    ring_init(ring, ring_words, P.slots);
    uint32_t* const out0 = ring.out32;
    zz_chunker ck = { 0, 0 };
    uint32_t nlog = 0;
    uint64_t done = 0;                                                  // input bytes encoded so far
    bool truncated = false;
    while (done < P.n) {                                                // AddData, encoder.cpp:539-552
        const int64_t byteCount = (int64_t)(P.n - done);
        // EnsureOutputLength(byteCount) - 1 (encoder.cpp:331): the packer stores whole 64-bit words
        // (outputbitstream.h:83-98), so "bytes stored" is the bit count rounded down to 64
        const uint64_t bits = (uint64_t)(ring.out32 - out0) * 32 + ring.bitpos;
        const uint64_t stored = (bits >> 6) << 3;
        int64_t avail;
        if (C.chunked) {
            bool opened;
            avail = zz_chunk_ensure(ck, stored, byteCount, &opened);
            zz_log_ensure(C, nlog, stored, (uint64_t)byteCount);
        } else {
            avail = (int64_t)C.cap - (int64_t)stored;
        }
        avail -= 1;
        int64_t nb = (avail * 8) / 9 - 8;                               // encoder.cpp:332-333
        bool final = true;
        if (nb < byteCount) final = false;                              // encoder.cpp:334-337
        else nb = byteCount;
        ring_append_uniform(ring, (final ? 1u : 0u) | (1u << 1), 3);    // StartBlock(FixedHuffman, final)
        if (nb <= 0) {
            // no room for even one byte: the reference writes this empty block and AddData gives up (encoder.cpp:546-
            // 548), leaving a stream that does not decode. Reported to the caller.
            ring_append_uniform(ring, 0, 7);
            truncated = true;
            break;
        }
        l1_encode_body<true, uint32_t>(P, T, nullptr, ring, P.src, P.src + P.n, (uint32_t)(done + (uint64_t)nb), nullptr, (uint32_t)done);
        ring_append_uniform(ring, 0, 7);                                // codes_f[256], encoder.cpp:371
        done += (uint64_t)nb;
    }
    const uint64_t bytes = (uint64_t)(ring.out32 - out0) * 4 + ring_finish(ring);
    if (lane == 0) {
        P.sizes[0] = (uint32_t)bytes;
        if (bytes > (uint64_t)P.slot_stride * P.npk) atomicOr(P.err, 1u);
        if (C.log_n) *C.log_n = nlog;
        if (truncated && C.truncated) *C.truncated = 1u;
    }
}

}  // namespace zz
