// zz_level2p.h -- levels 2 and 3: the token pass on TWO parsing wavefronts (k_encode_l2_t<0, false, true>, "k_encode_l2p").
//
// The same token pass as l2_token_pass (zz_level2.h), i.e. FirstPass + AddHashEntries (encoder.cpp:375-440, 474-480) over a
// packet, bit for bit; what changes is who does what when -- level 1's remedy (zz_level1p.h), and simpler here: at this level
// EVERY position is entered whatever the parse does (encoder.cpp:418, 474-480), so a probe's candidate -- the nearest earlier
// position with its hash -- and both quick lengths are parse-independent. Only the greedy walk ("first probe whose forward +
// backward match is >= 4", backRefEnd, the next probe position) is a chain through the packet. So the packet's blocks of 64
// positions alternate between two parsing wavefronts:
//
//      parser 0:  [P 0] B0 [W 0] B1 [P 2] B2 [W 2] B3 [P 4] ...          P g: enter block g's positions, candidates, quick lengths
//      parser 1:        B0 [P 1] B1 [W 1] B2 [P 3] B3 [W 3] ...          W g: walk block g, its matches to symbols and to the helper
//
// with one s_barrier per block (Bg), which the helper wavefront joins. Between two barriers one parser enters and compares while
// the other walks; nothing speculative, no cross lanes, no repair: the table order is the barrier order (P g in front of Bg,
// P g+1 behind it), and the walk's two scalars -- backRefEnd and the next probe position -- travel through two LDS words (xb).
// What the one-parser form left to the helper and this one moves to the walker's side: the matches' starts and lengths, their
// length / distance symbols (GetFrequencies' lookups, encoder.cpp:455-463) and the covered / match-start bits, so that the
// helper -- which an instrumented build showed busy for 1.66 M of the token pass's 1.83 M cycles per packet -- keeps the token
// stores, the symbol counts, the finished blocks' records and the Adler-32 sums.
//
// The one place where an entry depends on the parse is the batch switch (encoder.cpp:228-230, 435-438): a batch's first byte is
// entered only if the batch before ran over it. A packet has at most one switch (16,384 + 16,125 >= its search region), at
// block 256, which parser 0 owns: that block's P waits for the walk in front of it, and one extra barrier (Bx) keeps parser 1's
// P 257 behind it -- a bubble once per packet.
#pragma once

namespace zz {

#define ZZ_L2P_THREADS (3 * ZZ_WAVE)
// wavefronts per SIMD the kernel is compiled for: 7 = nine workgroups per CU (27 wavefronts), at most 72 VGPRs; 6 = eight, 84
#ifndef ZZ_L2P_WPE
#define ZZ_L2P_WPE 7
#endif
// issue priorities of a parsing wavefront while it walks (the packet's chain) and while it enters and compares
#ifndef ZZ_L2P_PRIO_W
#define ZZ_L2P_PRIO_W 3
#endif
#ifndef ZZ_L2P_PRIO_P
#define ZZ_L2P_PRIO_P 1
#endif
#ifndef ZZ_L2P_PRIO_H
#define ZZ_L2P_PRIO_H 0         // ... and of the helper during the token pass
#endif
// who counts a match's length and distance symbols: 0 = the helper (from the match word), 1 = the walker (it has them at hand)
#ifndef ZZ_L2P_HIST_W
#define ZZ_L2P_HIST_W 0
#endif
// where the walk's packed word per lane (hop pointers, flags) is put together: 0 = in front of the barrier (the prober's side), 1 = behind it
#ifndef ZZ_L2P_WINFO_W
#define ZZ_L2P_WINFO_W 0
#endif
// the walk's out-of-line token: 1 = forward and backward extension in one memory round trip (l2p_extend_both), 0 = one after the other
// (measured: 1 is 0.3 % slower on text and 0.7 % on the mix -- profiles/r05_ab_l2p_extension_and_priorities_*.txt; kept for the record)
#ifndef ZZ_L2P_EXT_BOTH
#define ZZ_L2P_EXT_BOTH 0
#endif
// bytes a probe compares forward on the prober's side: 16 (zz_level2.h's quick length) or 32 -- where some lane of a block agrees with its
// candidate in all sixteen bytes, every lane loads the next sixteen at its position and at its candidate, and only a length of "32 or
// more" is left to the walk's out-of-line path (a memory round trip on the packet's chain: profiles/r05_family_rates_and_free_extension_probe.txt)
// ZZ_L2P_FWD32 = k: up to k further rounds of sixteen bytes (0: the quick length as it was; 1: 32 bytes; 2: 48)
#ifndef ZZ_L2P_FWD32
#define ZZ_L2P_FWD32 1
#endif
#define ZZ_L2P_FCAP (16u + 16u * ZZ_L2P_FWD32)
#if ZZ_L2P_FWD32 && ZZ_L2P_EXT_BOTH
#error "l2p_extend_both starts at byte 16: build it with -DZZ_L2P_FWD32=0"
#endif
// who builds the codes and emits the first part behind the token pass: 1 = the helper (the parsers' token pass keeps its registers), 0 = wavefront 0
// (measured, profiles/r05_ab_l2p_roles_and_forward_32_*.txt: with the helper coding the spills of the parsers' out-of-line path go --
// parser 0's walk on the mix 2,541 -> 1,261 cycles per block -- and level 2 text is 1.1 % slower, the mix the same, log lines +1.2 %: 0)
#ifndef ZZ_L2P_HELPER_CODES
#define ZZ_L2P_HELPER_CODES 0
#endif
// How a block's positions get into the table: 0 = read / write / read-back, six-ballot same-hash sets, the last member rewrites the
// slot, in-block candidates' bytes through ds_bpermute (zz_level2.h's way: asks the LDS for nothing undocumented); 1 = ONE ordered
// exchange per lane (ds_mskor_rtn_b32 on the word that holds the 16-bit slot): the LDS serves the lanes of one instruction that hit
// one address in ascending lane order, so what a lane gets back IS its candidate -- the table's entry, or the nearest lower lane of
// the block with its hash -- and the slot ends up with the last one; no read-back, no sets, no rewrite, and a candidate inside the
// block is loaded like any other (its bytes are in the cache). About 65 of the prober's 170 instructions per block. The property is
// the one level 1's kernel stands on: probed per device (k_lds_order_probe checks this very instruction), checked here in the blocks
// at a packet's edges (what comes back must lie below the lane's own position), ZZ_ERR_LDS_ORDER reruns the call on the
// two-wavefront kernel.
#ifndef ZZ_L2P_XCHG
#define ZZ_L2P_XCHG 1
#endif
// the distance code's lengths built by the second parser beside wavefront 0's literal / length code (it idles there otherwise)
#ifndef ZZ_L2P_DIST_ON_PB
#define ZZ_L2P_DIST_ON_PB 1
#endif
// the body's three parts in 64ths of the records: wavefront 0 takes [0, SPLIT1), the helper [SPLIT1, SPLIT2), the second parser the
// rest; a dry run over a record costs about 0.4 of emitting it, so equal finishing times want 0.51 / 0.31 / 0.18
#ifndef ZZ_L2P_SPLIT1
#define ZZ_L2P_SPLIT1 33u
#endif
#ifndef ZZ_L2P_SPLIT2
#define ZZ_L2P_SPLIT2 52u
#endif
#define ZZ_L2P_SWITCH_BLOCK (ZZ_BATCH_LEN / ZZ_WAVE)      // block 256: where the second batch starts (if the search region reaches it)

// ZZ_L2P_FLAGS = 1: the three wavefronts meet through counters in LDS instead of one s_barrier per block. With the barrier a block's
// interval is walk + symbols on one side and enter + compare on the other, and the next walk starts behind BOTH; with the counters
// the next walk starts as soon as backRefEnd and the probe position are out, and the symbols, the bits and the hand-over run
// beside it. A wait polls its counter (one LDS read, s_sleep between reads) and is bounded: a wait that runs out reports
// ZZ_ERR_SYNC_TIMEOUT and lets everybody pass (the stream is then refused by the host), so no wavefront can spin for good.
// Measured (profiles/r05_ab_l2p_counters_instead_of_barriers_*.txt): bit-exact, level 2 text 89.5 -> 84.1 GB/s, level 3 mix 75.5 -> 75.9:
// a wavefront asleep in s_barrier costs nothing, a polling one takes issue slots from the eight packets beside it (round 1 found
// the same for a polling emitter at level 1). Off; kept as the measured alternative.
#ifndef ZZ_L2P_FLAGS
#define ZZ_L2P_FLAGS 0
#endif
#define ZZ_ERR_SYNC_TIMEOUT 8u
struct l2p_sync {
    uint32_t B, nextProbe;      // backRefEnd and the next probe position as the walk of block `walked - 1` left them
    uint32_t walked;            // blocks walked
    uint32_t entered;           // blocks entered into the table
    uint32_t tok[2];            // per hand-over slot: 1 + the block whose match words (and covered / start bits) are in place
    uint32_t cons;              // blocks the helper has picked up
    uint32_t pad;
};
__device__ __forceinline__ void l2p_wait_ge(uint32_t* f, uint32_t need, uint32_t* err)
{
    volatile lds_u32* vf = (volatile lds_u32*)f;
    if (uniform(*vf) >= need) return;
    for (uint32_t spins = 0;; ++spins) {
        __builtin_amdgcn_s_sleep(1);
        if (uniform(*vf) >= need) return;
        if (spins > (1u << 16)) {                      // ~10 ms: nothing in a packet takes that long
            *vf = 0x7FFFFFF0u;
            if (lane_id() == 0) atomicOr(err, ZZ_ERR_SYNC_TIMEOUT);
            return;
        }
    }
}

// the switch block of a packet whose search region is [0, target), or ~0: none
__device__ __forceinline__ uint32_t l2p_switch_block(uint32_t target) { return target > ZZ_BATCH_LEN ? ZZ_L2P_SWITCH_BLOCK : 0xFFFFFFFFu; }

// The walk's out-of-line token: a forward length of "16 or more" (remain(), encoder.cpp:64-90) and / or a backward length of "8 or
// more" (countMatchBackward, :92-102) measured by the whole wavefront, four bytes per lane -- l1p_extend_match and wave_extend_back
// (zz_level1.h, zz_level2.h) in ONE memory round trip: both pairs of loads go out before either result is looked at. On the mix
// such a token comes 0.35 times per block and was 4,400 cycles of the walker's time.
__device__ __forceinline__ void l2p_extend_both(const l1p_src& TS, const uint8_t* src, uint32_t qe, int32_t ce, bool need_f, bool need_b,
                                                uint32_t blim, uint32_t& fwd, uint32_t& bw)
{
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t of = 16 + 4 * lane, ob = 8 + 4 * lane;
    const bool actf = need_f && of < ZZ_MAX_LEN, actb = need_b && ob < blim;
    uint32_t df = 0, db = 0;
    if (actf) df = load32(l1p_addr<true>(TS, (int32_t)(qe + of))) ^ load32(l1p_addr<true>(TS, ce + (int32_t)of));
    if (actb) {
        // (the caller guarantees ce - blim >= the start of readable memory; a 4-byte load may reach up to 3 bytes below that: byte-wise there)
        if (ob + 4 <= blim) db = load32(src + (int64_t)qe - ob - 4) ^ load32(src + (int64_t)ce - ob - 4);
        else {
            for (uint32_t i = 0; i < 4; ++i)
                if (ob + i < blim) db |= (uint32_t)(src[(int64_t)qe - ob - 1 - i] ^ src[(int64_t)ce - ob - 1 - i]) << (8 * (3 - i));
        }
    }
    if (need_f) {
        const uint64_t neq = ballot(actf && df != 0);
        uint32_t len = ZZ_MAX_LEN;
        if (neq) {
            const int k = __builtin_ctzll(neq);
            const uint32_t dk = readlane(df, k);
            len = 16 + 4 * (uint32_t)k + ((uint32_t)__builtin_ctz(dk) >> 3);
            if (len > ZZ_MAX_LEN) len = ZZ_MAX_LEN;
        }
        fwd = len;
    }
    if (need_b) {
        const uint64_t neq = ballot(actb && db != 0);
        uint32_t len = blim;
        if (neq) {
            const int k = __builtin_ctzll(neq);
            const uint32_t dk = readlane(db, k);
            len = 8 + 4 * (uint32_t)k + ((uint32_t)__builtin_clz(dk) >> 3);   // top byte = nearest
            if (len > blim) len = blim;
        }
        bw = len;
    }
}

// One parsing wavefront (pw = 0: even blocks, 1: odd blocks). xb: [0] backRefEnd, [1] the next probe position, as the walk
// of the block walked last left them. Barriers: see above; every wavefront of the workgroup executes B0 .. B_NB (+ Bx).
__device__ __forceinline__ void l2p_token_pass(uint16_t* T, uint32_t* hb, uint32_t* xb, uint64_t* covw, uint64_t* mstw, uint32_t* histP, const uint8_t* src,
                                               const uint8_t* end, const l1p_src& TS, const uint32_t n, const uint64_t before, const uint32_t pw,
                                               uint32_t* err, const uint32_t dbg_viol, unsigned long long* prof = nullptr)
{
    l2p_sync* const S = (l2p_sync*)xb;     // (xb[0], xb[1] are S->B, S->nextProbe)
    (void)S; (void)err;
    const int lane = lane_id();
    ZZ_PROF_DECL
    const uint64_t below_me = (1ull << lane) - 1, above_me = ~((2ull << lane) - 1);
    const uint32_t target = n > ZZ_MAX_LEN ? n - ZZ_MAX_LEN : 0;     // :222 last 258 bytes never searched
    const uint32_t NB = (target + 63) >> 6;
    const uint32_t batch1 = target < ZZ_BATCH_LEN ? target : ZZ_BATCH_LEN;
    const uint32_t gs = l2p_switch_block(target);
    const uint32_t bcap = (uint32_t)(before < ZZ_MAX_LEN ? before : ZZ_MAX_LEN);
    const uint32_t lo8 = before >= 8 ? 0u : 8u - (uint32_t)before;
    const uint8_t* const srcm8 = src - 8;
    // this lane's own bytes of this wavefront's first block: 16 from its position and the 8 in front
    uint64_t wa = 0, wa2 = 0, wb = 0;
    {
        const uint32_t q0 = pw * ZZ_WAVE + (uint32_t)lane;
        if (q0 < n) {
            l1p_ld128<true>(TS, (int32_t)q0, wa, wa2);
            if (before + q0 >= 8) wb = load64(src + (int64_t)q0 - 8);
        }
    }
    uint32_t myNext = 1;            // the next probe position as this wavefront knew it last: a lower bound of the true one
    uint32_t viol = 0;              // ZZ_L2P_XCHG: the exchange handed a lane something that is not below its own position
#if !ZZ_L2P_FLAGS
    if (pw == 0 && lane == 0) { xb[0] = 1; xb[1] = 1; }               // backRefEnd (:380), j (:383)
    if (pw == 1) l2_block_barrier();                                  // B0
#endif
    // Loop state the block's code reads: set by whoever calls it. Most blocks are INTERIOR (below); they run in a loop of their own
    // with nothing between two blocks but the counter (every scalar instruction between two blocks is the prober's: 500 cycles of
    // loop top per block in the first form, profiles/r05_phases_l2p_text_first.txt). Everything else -- the packet's first blocks,
    // the switch block and the one behind it, the last ones -- goes through `general`.
    uint32_t g = pw, base = 0, skipPos = 0xFFFFFFFFu, batchEnd = batch1;
    bool sw = false;
    {
        // the walk's word per lane: fwd8 [4:0], need [7:5], "8 or more backward possible" bit 8, "16 or more forward" bit 9, plain bit 10,
        // strong bit 11, next candidate [21:16] (0 = none), lane + fwd8 [29:23] (zz_level2.h, ZZ_L2_HOP)
        auto make_winfo = [&](uint32_t fwd8, uint32_t broom, uint64_t Amask) -> uint32_t {
            // ("FCAP or more" forward -> bit 9, "8 or more" backward -> bit 8; the scalar loop never reads bits 0..4: no length there)
            const uint32_t inexact = (fwd8 == ZZ_L2P_FCAP ? 16u : 0u) | (broom & 8u);
            uint32_t winfo = (sub_from4_sat(fwd8) << 5) | (inexact << 5) | (fwd8 >= 4 ? 0x800u : 0u) |
                             ((fwd8 >= 4 && inexact == 0) ? 0x400u : 0u) | (((uint32_t)lane + fwd8) << 23);
            const uint32_t endl = (uint32_t)lane + fwd8 + 1;                  // first lane probed after a match here
            const uint64_t m = Amask >> (endl & 63u);
            const uint32_t nx = endl + (m ? (uint32_t)__builtin_ctzll(m) : 64u);
            return winfo | (((nx < 64u ? nx : 64u) & 63u) << 16);             // (64 & 63 = 0 = none)
        };
        auto block = [&](auto interior_tag) {
            // INTERIOR: not the packet's first block, not the switch block, every position inside the current batch
            constexpr bool INTERIOR = decltype(interior_tag)::value;
            ZZ_T(5); ZZ_C(10, 1);
#if ZZ_L2P_FLAGS
            l2p_wait_ge(&S->entered, g, err);                             // the block in front is in the table
#endif
            const uint32_t q = base + (uint32_t)lane;
            const bool ins = INTERIOR ? true : (q != skipPos && q != 0);
            // (a block that a match found two blocks ago covers entirely is entered but not compared: myNext is a lower bound)
            const bool cmp = myNext < base + 64;
            const uint32_t h = calc_hash3((uint32_t)wa);                  // CalcHash(source + j), :388
            const uint32_t hs = ins ? h : (uint32_t)(ZZ_L2_LDS_BYTES / 2);
#if ZZ_L2P_XCHG
            // this wavefront's next block (g + 2): its own bytes, in flight during the rest of this one (inside the packet: q + 143 < n)
            uint64_t wan, wan2, wbn;
            ld128<false>(src + q + 2 * ZZ_WAVE, end, wan, wan2);
            wbn = load64(src + q + 2 * ZZ_WAVE - 8);
            uint32_t old;                                                 // :389-390 in one: what the slot held when this lane's turn came
            {
                const uint32_t baddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)(T + hs);
                const uint32_t sh = (baddr & 2u) << 3;
                uint32_t ret;
                asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)"
                             : "=v"(ret) : "v"(baddr & ~3u), "v"(0xFFFFu << sh), "v"((q + 1) << sh) : "memory");
                old = (ret >> sh) & 0xFFFFu;
            }
            if (!INTERIOR && ins && old > q) viol = 1u;                   // (an entry is a position + 1 below mine: anything else came out of order)
            if (!ins) old = 0;
            uint64_t ca = 0, ca2 = 0, cpre = 0;
            if (cmp) {
                const uint32_t c0 = __builtin_elementwise_sub_sat(old, 1u);
                ld128<false>(src + c0, end, ca, ca2);
                cpre = load64(srcm8 + (c0 > lo8 ? c0 : lo8));          // (too close to the stream's start: fixed up below)
            }
            const uint32_t cand1 = old;                                   // candidate as pos+1, 0 = none
            const uint64_t lost = 0;
            const bool inl = false;
            const uint32_t il = 0;
            (void)lost; (void)inl; (void)il; (void)below_me; (void)above_me;
#else
            uint32_t old = T[hs];                                         // :389
            T[hs] = (uint16_t)(q + 1);                                    // :390 / :474-480
            if (!ins) old = 0;
            uint64_t ca = 0, ca2 = 0, cpre = 0;
            if (cmp) {
                const uint32_t c0 = __builtin_elementwise_sub_sat(old, 1u);
                ld128<false>(src + c0, end, ca, ca2);
                cpre = load64(srcm8 + (c0 > lo8 ? c0 : lo8));          // (too close to the stream's start: fixed up below)
            }
            // this wavefront's next block (g + 2): its own bytes, in flight during the rest of this one (inside the packet: q + 143 < n)
            uint64_t wan, wan2, wbn;
            ld128<false>(src + q + 2 * ZZ_WAVE, end, wan, wan2);
            wbn = load64(src + q + 2 * ZZ_WAVE - 8);
            ZZ_WAVE_SYNC();
            const uint32_t rb = T[hs];
            uint32_t cand1 = old;                                         // candidate as pos+1, 0 = none
            bool inl = false;                                             // the candidate is lane `il` of this block
            uint32_t il = 0;
            const uint64_t lost = INTERIOR ? ballot(rb != q + 1) : ballot(ins && rb != q + 1);
            if (lost) {
                const uint32_t W = ins ? (rb - 1u - base) & 63u : (uint32_t)lane;
                const uint64_t set = wave_match6(W);
                const uint64_t below = set & below_me;
                inl = ins && below != 0;
                il = 63u - (uint32_t)__builtin_clzll(below | 1ull);       // nearest earlier member (lane 0 where there is none: unused)
                cand1 = inl ? base + il + 1 : old;
                ZZ_WAVE_SYNC();
                // last member wins: the highest lane of a set rewrites the slot unless its own store was the one that landed
                if (INTERIOR) {
                    const uint64_t fix = ballot(W != (uint32_t)lane) & ballot((set & above_me) == 0);
                    uint64_t saved;
                    asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write_b16 %2, %3\n\ts_mov_b64 exec, %0"
                                 : "=&s"(saved) : "s"(fix), "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)(T + h)), "v"(q + 1)
                                 : "memory", "scc");
                } else if (ins && W != (uint32_t)lane && (set & above_me) == 0) T[h] = (uint16_t)(q + 1);
            }
#endif
            ZZ_WAVE_SYNC();
#if ZZ_L2P_FLAGS
            S->entered = g + 1;                                           // (behind the stores above: the LDS takes a wavefront's instructions in order)
#endif

            // ---- quick compare info for all 64 probes of this block (zz_level2.h, l2_token_pass) ---------------------------
            uint32_t fwd8 = 0, broom = 0, room = 0, winfo = 0;
            uint64_t Amask = 0;
            const int32_t c = (int32_t)(cand1 - 1);
            if (cmp) {
                const bool has = INTERIOR ? cand1 != 0 : (ins && cand1 != 0 && q < batchEnd);
                if (lost) {
                    const int qa = (int)(il << 2);
                    const uint64_t sa = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)(wa >> 32)) << 32) | (uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)wa);
                    const uint64_t sa2 = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)(wa2 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)wa2);
                    const uint64_t sp = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)(wb >> 32)) << 32) | (uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)wb);
                    if (inl) { ca = sa; ca2 = sa2; cpre = sp; }
                }
                uint32_t bwd8;
                {
                    const uint32_t r = bcap + (uint32_t)c;                 // bytes in front of the candidate, as far as they count
                    room = r < ZZ_MAX_LEN ? r : ZZ_MAX_LEN;                // D4 + D11 caps
                    const uint64_t y = wb ^ cpre;
                    bwd8 = umin3(ffbh_or_ones((uint32_t)(y >> 32)), add_sat_k<32>(ffbh_or_ones((uint32_t)y)), 64u) >> 3;
                    if (before < 8 && ballot(has && room < 8)) {           // the first bytes of a stream: byte by byte
                        if (has && room < 8) {
                            bwd8 = 0;
                            while (bwd8 < room && src[(int64_t)q - 1 - bwd8] == src[(int64_t)c - 1 - bwd8]) bwd8++;
                        }
                    }
                    fwd8 = equal_bits128(wa ^ ca, wa2 ^ ca2, 128u) >> 3;   // 16 = "16 or more" (:399)
                    if (!has) fwd8 = 0;
#if ZZ_L2P_FWD32
#pragma unroll
                    for (uint32_t at = 16u; at < ZZ_L2P_FCAP; at += 16u) {
                        if (!ballot(fwd8 == at)) break;
                        // the next sixteen bytes (every lane loads: no lane mask; a lane that is not that far reads its own bytes
                        // against its own: inside the packet either way, q + 16 + FCAP < n and the candidate lies below q)
                        const bool go = fwd8 == at;
                        uint64_t ya, ya2, yc, yc2;
                        ld128<false>(src + q + (go ? at : 0u), end, ya, ya2);
                        ld128<false>(src + (go ? c + (int32_t)at : (int32_t)q), end, yc, yc2);
                        const uint32_t f2 = equal_bits128(ya ^ yc, ya2 ^ yc2, 128u) >> 3;
                        if (go) fwd8 = at + f2;                           // FCAP = "FCAP or more"
                    }
#endif
                }
                broom = bwd8 < room ? bwd8 : room;
                if (!has) broom = 0;
                Amask = ballot(fwd8 + broom >= 4);                                // strong or weak
                if (!ZZ_L2P_WINFO_W) winfo = make_winfo(fwd8, broom, Amask);
            }
            ZZ_T(0);
#if ZZ_L2P_FLAGS
            l2p_wait_ge(&S->walked, g, err);                              // the block in front has been walked
#else
            l2_block_barrier();                                           // Bg (the switch block: Bx) -- the block in front has been walked
#endif
            ZZ_T(1);
            if (ZZ_L2P_PRIO_W != ZZ_L2P_PRIO_P) __builtin_amdgcn_s_setprio(ZZ_L2P_PRIO_W);

            // ---- W: the greedy walk (zz_level2.h: the same scalar loop) ----------------------------------------------------------
            uint32_t B = uniform(xb[0]), nextProbe = uniform(xb[1]);
            if (ZZ_L2P_WINFO_W) winfo = make_winfo(fwd8, broom, Amask);
            const bool doProbe = base + 64 > nextProbe && nextProbe < batchEnd;   // some position of this block is probed
            uint32_t* slot = hb + (g & 1u) * ZZ_L2_HB_WORDS;
            uint64_t evmask = 0, slowmask = 0;
            uint32_t tk = 0;
            const uint32_t Bentry = B;
            if (doProbe) {
                uint32_t np = (int32_t)(nextProbe - base) > 0 ? nextProbe - base : 0u;
                int32_t Brel = (int32_t)(B - base);
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
                        const uint32_t re = readlane(room, e);
                        const uint32_t blim = re < pe ? re : pe;
#ifdef ZZ_L2P_X_NOEXT
                        (void)ce; (void)blim;                            // TIMING EXPERIMENT (valid but WRONG streams): no extension loads
#elif ZZ_L2P_EXT_BOTH
                        l2p_extend_both(TS, src, qe, ce, fwd == ZZ_L2P_FCAP, bw == 8 && blim > 8, blim, fwd, bw);
#else
                        if (fwd == ZZ_L2P_FCAP) fwd = l1p_extend_match(TS, qe, ce, ZZ_MAX_LEN, ZZ_L2P_FCAP);    // remain(), :64-90
                        if (bw == 8 && blim > 8) bw = wave_extend_back(src, qe, ce, blim);            // :92-102
#endif
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
                // the parser behind is waiting for these two and nothing else: out first
#if ZZ_L2P_FLAGS
                xb[0] = B; xb[1] = nextProbe;                             // (the same words from all lanes)
#else
                if (lane == 0) { xb[0] = B; xb[1] = nextProbe; }
#endif
            }
#if ZZ_L2P_FLAGS
            ZZ_WAVE_SYNC();
            S->walked = g + 1;                                            // the walk of block g + 1 may start
            if (ZZ_L2P_PRIO_W != ZZ_L2P_PRIO_P) __builtin_amdgcn_s_setprio(ZZ_L2P_PRIO_P);     // what follows runs beside it
            if (g >= 2) l2p_wait_ge(&S->cons, g - 1, err);                // the helper has taken block g - 2 out of this slot
#endif
            myNext = nextProbe;
            ZZ_T(2); ZZ_C(11, (uint32_t)__builtin_popcountll(evmask)); ZZ_C(12, (uint32_t)__builtin_popcountll(slowmask));
            // ---- this block's matches: starts and lengths, their symbols, the covered / match-start bits; one word per match
            // to the helper (length symbol - 257 [27:23], length extra value [22:18], distance code [17:13], distance extra value [12:0])
            if (evmask) {
                const bool slowl = (slowmask >> lane) & 1;
                uint32_t ms = tk & 0xFFFFu, mlen = tk >> 16;               // as the C++ path left them
                // matches the scalar loop only marked: backRefEnd after a match is probe + forward length (:416,422); the backward
                // part is limited by the literals pending since the previous match of this block (or since the block was entered)
                const uint32_t endp = slowl ? ms + mlen : q + fwd8;
                const uint64_t prev = evmask & below_me;
                const int pl = prev ? 63 - __builtin_clzll(prev) : lane;
                const uint32_t pend_end = (uint32_t)__shfl((int)endp, pl);            // every lane takes part
                if (!slowl) {
                    const uint32_t pe = q - (prev ? pend_end : Bentry);
                    const uint32_t bq = broom < pe ? broom : pe;
                    ms = q - bq; mlen = fwd8 + bq;
                }
                if ((evmask >> lane) & 1) {
                    const uint32_t dist = (uint32_t)((int32_t)q - c);
                    uint32_t sym, leb, lev, bucket, deb, dev;            // GetFrequencies, :455-463
                    length_symbol(mlen, sym, leb, lev);
                    dist_symbol(dist, bucket, deb, dev);
                    slot[lane] = ((sym - 257) << 23) | (lev << 18) | (bucket << 13) | dev;
                    if (ZZ_L2P_HIST_W) { hist_add(histP, sym); hist_add(histP, 286 + bucket); }
                    // covered / start bits: words (base>>6)-5 .. (base>>6)+5 of the LDS window
                    const uint32_t last = ms + mlen - 1;
                    const uint32_t w0 = ms >> 6, w1 = last >> 6;
                    const uint64_t from_lo = ~0ull << (ms & 63), to_hi = ~0ull >> (63u - (last & 63));
                    atomicOr((unsigned long long*)&covw[w0 & (ZZ_L2_WIN - 1)], (unsigned long long)(w1 == w0 ? (from_lo & to_hi) : from_lo));
                    if (w1 != w0) {
                        atomicOr((unsigned long long*)&covw[w1 & (ZZ_L2_WIN - 1)], (unsigned long long)to_hi);
                        for (uint32_t wi = w0 + 1; wi < w1; ++wi) atomicOr((unsigned long long*)&covw[wi & (ZZ_L2_WIN - 1)], ~0ull);
                    }
                    atomicOr((unsigned long long*)&mstw[(ms >> 6) & (ZZ_L2_WIN - 1)], 1ull << (ms & 63));
                }
            }
            slot[64] = (uint32_t)evmask; slot[65] = (uint32_t)(evmask >> 32);     // (the same words from all lanes: no lane mask to set up)
            wa = wan; wa2 = wan2; wb = wbn;
            ZZ_T(3);
#if ZZ_L2P_FLAGS
            ZZ_WAVE_SYNC();
            S->tok[g & 1u] = g + 1;                                       // the helper may take this block's match words
#else
            l2_block_barrier();                                           // B_g+1: this block has been walked
            ZZ_T(4);
            if (ZZ_L2P_PRIO_W != ZZ_L2P_PRIO_P) __builtin_amdgcn_s_setprio(ZZ_L2P_PRIO_P);
#endif
        };
        // the blocks [g, hi) that are INTERIOR (not block 0 or 1, not the switch block or the one behind it, every position inside the batch)
        auto interior_end = [&](uint32_t g0) -> uint32_t {
            if (g0 < 2 || g0 == gs || (gs == ZZ_L2P_SWITCH_BLOCK && g0 == ZZ_L2P_SWITCH_BLOCK + 1)) return g0;
            const uint32_t hi = g0 < gs ? ((batch1 >> 6) < gs ? (batch1 >> 6) : gs) : (target >> 6);
            return hi < NB ? hi : NB;
        };
        while (g < NB) {
            const uint32_t hi = interior_end(g);
            if (g < hi) {
                sw = false; skipPos = 0xFFFFFFFFu; batchEnd = g < gs ? batch1 : target;
                for (; g < hi; g += 2) { base = g << 6; block(std::true_type{}); }
                continue;
            }
            // ---- general: one block with everything that can be special about it
            base = g << 6;
            sw = g == gs;
#if !ZZ_L2P_FLAGS
            if (g == ZZ_L2P_SWITCH_BLOCK + 1 && gs == ZZ_L2P_SWITCH_BLOCK) l2_block_barrier();   // Bx: the switch block's entries are in the table
#endif
            skipPos = 0xFFFFFFFFu;
            // blocks behind the switch belong to the second batch, which ends where the search region ends: a packet's rest is shorter
            // than a batch (s2 >= 16384, target <= 32509); a match that overruns batch AND region leaves nothing to probe (nextProbe > target)
            batchEnd = g < gs ? batch1 : target;
            if (sw) {
                // batch switch (:228-230, :435-438): the next batch starts at max(backRefEnd, end); its first byte is entered only if
                // the last match covered it. Needs the walk of block g - 1: this block's P runs behind Bg, and Bx follows it.
#if ZZ_L2P_FLAGS
                l2p_wait_ge(&S->walked, g, err);
#else
                l2_block_barrier();                                       // Bg
#endif
                const uint32_t Bw = uniform(xb[0]);
                const uint32_t s2 = Bw > batch1 ? Bw : batch1;
                skipPos = Bw >= batch1 ? 0xFFFFFFFFu : batch1;
                ZZ_WAVE_SYNC();
                if (lane == 0) { xb[0] = s2 + 1; xb[1] = s2 + 1; }
                myNext = s2 + 1;
            }
            block(std::false_type{});
            g += 2;
        }
    }
#if !ZZ_L2P_FLAGS
    if (gs < NB && gs + 1 >= NB && pw == 1) l2_block_barrier();          // Bx, where the switch block is the packet's last
    if (((NB - 1) & 1u) != pw) l2_block_barrier();                       // B_NB: the other parser's last walk (NB = 0: B0)
#endif
    if (ZZ_L2P_XCHG && (viol | dbg_viol)) atomicOr(err, ZZ_ERR_LDS_ORDER);             // (per lane, no ballot: rare, and this kernel has no scalar register to spare)
#ifdef ZZ_PROF
    if (lane == 0 && prof) for (int _i = 0; _i < 16; ++_i) atomicAdd(&prof[16 * (1 + pw) + _i], prof_acc[_i]);
#endif
    (void)prof;
}

// The helper wavefront's side: per block the parsers' match words go to the packet's scratch and into the symbol counts, and the
// block that can no longer change is turned into records (zz_level2.h, l2_helper_pass: everything else is the walker's now).
#ifndef ZZ_L2P_SNAP
#define ZZ_L2P_SNAP 1            // the body cut for its three emitters by BLOCKS, the bits in front of a part from the symbol counters as they stood there (0: by records, a dry run each)
#endif
static_assert(!(ZZ_L2P_SNAP && ZZ_L2P_HIST_W), "the snapshots are taken where the helper counts the match symbols");
// the first block of the body's second (which = 1) and third (2) part; no such cut (0xFFFFFFFF) for packets of fewer than 16 probed blocks
__device__ __forceinline__ uint32_t l2p_cut(uint32_t n, uint32_t which)
{
    const uint32_t target = n > ZZ_MAX_LEN ? n - ZZ_MAX_LEN : 0, trips = (target + 63) >> 6;      // (l2_probe_blocks)
    if (!ZZ_L2P_SNAP || trips < 16u) return 0xFFFFFFFFu;
    return (which * ((n + 63u) >> 6)) / 3u;
}
__device__ __forceinline__ uint32_t l2p_helper_pass(const uint32_t* hb, uint64_t* covw, uint64_t* mstw, uint32_t* histP,
                                                    uint32_t* tokens, uint16_t* recs, const uint8_t* src, uint32_t n,
                                                    uint32_t& nrec_out, uint32_t& adA, uint64_t& adC, uint32_t* xb, uint32_t* err,
                                                    uint32_t* snap, unsigned long long* prof = nullptr)
{
    l2p_sync* const S = (l2p_sync*)xb;
    (void)S; (void)err;
    const int lane = lane_id();
    ZZ_PROF_DECL
    const uint32_t target = n > ZZ_MAX_LEN ? n - ZZ_MAX_LEN : 0;
    const uint32_t trips = (target + 63) >> 6;
    const uint32_t gs = l2p_switch_block(target);
    uint32_t nrec = 0, Fnext = 0, ntok = 0;
    adA = 0; adC = 0;
    // Where the body is cut for its three emitters (ZZ_L2P_SNAP): in front of the blocks F1 and F2, a third and two thirds of the way.
    // When the block in front of a cut is final the symbol counters go to scratch as they stand, with the number of records, of
    // matches among them and of matches that have arrived: with the code lengths they give the bits in front of the cut without a
    // dry run over its records -- less the few matches that have arrived but begin behind it (zz_level2.h).
    const uint32_t F1 = l2p_cut(n, 1), F2 = l2p_cut(n, 2);
    uint32_t nmrec = 0;                                                  // matches among the records so far
    auto snap_counters = [&](uint32_t finished) {
        if (finished == F1 || finished == F2) {
            uint32_t* sp = snap + (finished == F1 ? 0u : ZZ_L2_SNAP_WORDS);
            sp[lane] = histP[lane]; sp[64 + lane] = histP[64 + lane];             // symbols 0 .. 255, two to a word
            if (lane < 30) sp[128 + lane] = histP[128 + lane];                    // 256 .. 315: lengths and distances of every match that has ARRIVED
            if (lane == 0) { sp[158] = nmrec; sp[159] = nrec; sp[160] = ntok; }
        }
    };
    __builtin_amdgcn_s_setprio(ZZ_L2P_PRIO_H);                           // (down from the code construction's 3 of the packet before)
#if !ZZ_L2P_FLAGS
    l2_block_barrier();                                                  // B0
#endif
    for (uint32_t i = 0; i < trips; ++i) {
        const uint32_t base = i << 6;
        uint32_t fbyte = 0;
        if (i >= ZZ_L2_LAG) {
            const uint32_t p = base + lane - 64 * ZZ_L2_LAG;        // < n: the probe front is at least 258 bytes from the end
            fbyte = src[p];
            adA += fbyte; adC += (uint64_t)p * fbyte;
        }
        ZZ_T(1); ZZ_C(10, 1);
#if ZZ_L2P_FLAGS
        l2p_wait_ge(&S->tok[i & 1u], i + 1, err);                        // block i's match words and bits are in place
#else
        if (i == gs) l2_block_barrier();                                 // Bx
        l2_block_barrier();                                              // B_i+1: block i has been walked
#endif
        ZZ_T(0);
        const uint32_t* slot = hb + (i & 1) * ZZ_L2_HB_WORDS;
        const uint32_t tok = slot[lane];
        const uint64_t evmask = ((uint64_t)uniform(slot[65]) << 32) | uniform(slot[64]);
#if ZZ_L2P_FLAGS
        ZZ_WAVE_SYNC();
        S->cons = i + 1;                                                 // the slot is free for block i + 2
#endif
        if (evmask) {
            if ((evmask >> lane) & 1) {
                tokens[ntok + mbcnt(evmask)] = tok;
                if (!ZZ_L2P_HIST_W) {
                    hist_add(histP, 257 + (tok >> 23));
                    hist_add(histP, 286 + ((tok >> 13) & 31));
                }
            }
            ntok += (uint32_t)__builtin_popcountll(evmask);
        }
        // later matches start at >= base + 64 - 258: block (base>>6) - 5 cannot change any more
        if (i >= ZZ_L2_LAG) {
            ZZ_WAVE_SYNC();
            nrec = l2_finish_block(covw, mstw, histP, recs, nrec, Fnext, n, fbyte, &nmrec);
            Fnext++;
            snap_counters(Fnext);
        }
    }
    ZZ_WAVE_SYNC();
    for (const uint32_t nblk = (n + 63) >> 6; Fnext < nblk; ++Fnext) {       // the tail nobody probes (:222) + the lag
        const uint32_t p = (Fnext << 6) + (uint32_t)lane;
        const uint32_t d = p < n ? src[p] : 0u;
        adA += d; adC += (uint64_t)p * d;
        nrec = l2_finish_block(covw, mstw, histP, recs, nrec, Fnext, n, d, &nmrec);
        snap_counters(Fnext + 1);
    }
    nrec_out = nrec;
#ifdef ZZ_PROF
    if (lane == 0 && prof) for (int _i = 0; _i < 16; ++_i) atomicAdd(&prof[48 + _i], prof_acc[_i]);
#endif
    (void)prof;
    return ntok;
}

}  // namespace zz
