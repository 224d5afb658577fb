// zz_level1p.h -- level 1, packet mode: TWO parsing wavefronts per packet, pipelined over blocks of 64 positions.
//
// The same stream as k_encode_l1 (zz_level1.h), i.e. WriteBlockFixedHuff (encoder.cpp:329-373) over a packet
// (zzflate.cpp:101-125), bit for bit. What changes is who does what when. In k_encode_l1 ONE wavefront runs a packet's
// whole dependency chain: probe / compare (about 200 instructions per 64 positions), then the serial walk (about 100), then
// repair and tokens, then the next group -- and only nine such chains fit a CU (the 16 KiB table), which leaves the CU's
// issue slots two-thirds empty (profiles/README.md, round 4: tools/ceiling_probe.sh, tools/pipe_probe.sh). Here the packet
// is cut into fixed blocks of 64 positions and two wavefronts take alternate blocks:
//
//      wave A:  [probe 0] B0 [walk 0] B1 [repair 0, probe 2] B2 [walk 2] B3 [repair 2, probe 4] ...
//      wave B:            B0 [probe 1] B1 [walk 1] B2 [repair 1, probe 3] B3 [walk 3] ...
//
// so that the only thing left on the packet's critical path is walk, hand-over, walk, hand-over. Block g + 1 is probed
// while block g is still being walked, i.e. against a table that holds block g's SPECULATIVE entries (every position of
// g entered; the walk has not said yet which of them the reference would have entered). That is resolved exactly:
//   * a lane of g + 1 whose table read returns a position of block g ("cross lane", about a quarter of the lanes of text)
//     has two likely candidates -- that position q if the walk of g visits it, else what the table held before block g
//     under that hash (told_g[q], which block g's owner publishes, tagged with the block, as soon as it has settled its
//     own cross lanes) -- and asks for the bytes of both before the walk of g has finished. When it has, g's owner publishes
//     per lane the highest VISITED lane of the block with that lane's hash (`win`, 64 = none), and one gathered byte per
//     cross lane picks: q itself, nobody (the older entry), or -- 0.2 times per block of text -- a lower lane of g, whose
//     bytes are then fetched from the cache; ONE comparison per lane follows, against the candidate that is the right one
//     (`R` below; comparing against both in front of the barrier shortened R and lost 6 %: both parsers' intervals are
//     critical, and so is every instruction in either).
//   * repair of block g (skipped lanes restore the old entry, the highest visited lane per hash wins: the state the
//     serial loop leaves) runs after block g + 1 has entered its positions, so it only touches slots that still hold
//     a position of block g; a slot block g + 1 has overwritten is that block's to repair, from its resolved old entry.
//   * a match that runs past its block's end is carried: block g + 1's walk starts behind it (`cin`).
// This relies on the LDS serving the lanes of one store that hit one address in ascending lane order (the slot ends up
// with the HIGHEST position of the block under that hash): zz_ctx probes that once per device and falls back to
// k_encode_l1 where it does not hold (zz_api.hip, lds_order_ok).
//
// Hand-over between the two parsers goes through 468 bytes of LDS (l1p_xch) and one s_barrier per block, which the
// emitter wavefront (the third of the workgroup: Adler-32, fixed-Huffman coding, bit packing -- as in k_encode_l1) joins.
// LDS: 16,384 table + 512 ring + 512 token slots + 468 = 17,876 bytes <= 17,920: nine workgroups per CU, 27 wavefronts
// (61 VGPRs, 62 SGPRs: the 27 fit whatever SIMDs the workgroups' wavefronts land on).
// Measured (profiles/README.md, round 4; DESIGN.md 4 has the steps): 1 GiB text 119 -> 140 GB/s, log lines in gzip 96 -> 116,
// the twelve-family mix 102.5 -> 116; the optimistic bound of this shape (tools/pipe_probe.sh: no cross lanes, no
// exchange) was 159 before the kernel existed, and the first exact form ran at 124: most of the way from there to 140 was
// taking instructions out between the barriers, one A/B at a time.
#pragma once
#include "zz_level1.h"

namespace zz {

#define ZZ_L1P_THREADS (3 * ZZ_WAVE)
// where the work that may sit on either side of a barrier sits (tools/sweep_define.sh): the two wavefronts alternate
// between "walk" and "probe" intervals, and an interval lasts as long as the longer of the two
#ifndef ZZ_L1P_PRIO_W
#define ZZ_L1P_PRIO_W 3         // issue priority while a wavefront resolves and walks (the packet's critical path) ...
#endif
#ifndef ZZ_L1P_PRIO_F
#define ZZ_L1P_PRIO_F 1         // ... while it repairs and probes ...
#endif
#ifndef ZZ_L1P_PRIO_E
#define ZZ_L1P_PRIO_E 0         // ... and of the emitter
#endif
// The LDS-order invariant (l1p_parse, wmin) is checked inside the kernel: 0 (default) = in the last blocks of EVERY packet (the
// copy of the block's code that clamps at the packet's end: three blocks of a 32 KiB packet's 512, no instruction in the other
// 509), 1 = in every block (one v_min_u32 per block on the prober's side). Measured with tools/abn.sh against the kernel without
// any check, 1 GiB text / mix (profiles/r05_ab_l1p_order_check.txt): sampled -0.2 % / -0.3 %, every block -0.8 .. -1.1 % / -0.6 ..
// -1.2 % in four forms of the one instruction (inside the branch, in front of it, as asm, as C++) -- the instruction itself
// replaces a mask that was redundant, what costs is the register that lives through the block loop. VERDICT r04 set the bar at
// 0.5 %: the sampled form is the default, -DZZ_L1P_ORDER_CHECK=1 builds the other. Either way a violation takes the device's
// verdict away and the call runs again on k_encode_l1 (zz_api.hip encode_finish).
#ifndef ZZ_L1P_ORDER_CHECK
#define ZZ_L1P_ORDER_CHECK 0
#endif
// the prober compares 32 bytes where the walk has just needed more than 16 (l1p_parse, hot_until). Measured (bit-exact;
// profiles/r05_ab_l1p_ext32.txt): text +0.4 %, mix -2.3 %, log lines in gzip -9 %, database dump 102 -> 81 GB/s -- at level 1 the
// prober's interval is as critical as the walker's, and a second memory round trip in it costs more than the one it saves the
// walk (at level 2, where the prober has slack on such data, the same idea is worth +4 % on the mix and +9 % on log lines). Off.
#ifndef ZZ_L1P_EXT32
#define ZZ_L1P_EXT32 0
#endif
// the emitter's completed words leave the bit ring in batches of 32..64 (zz_emit.h ring_append_lazy) instead of after every block
#ifndef ZZ_L1P_ADLER3
#define ZZ_L1P_ADLER3 1          // cold packets: the Adler-32 sums in three parts, one per wavefront, in front of the first barrier (0: the emitter alone, behind it)
#endif
#ifndef ZZ_L1P_ADLER_U
#define ZZ_L1P_ADLER_U 6        // ... with this many 16-byte loads in flight per lane
#endif
#ifndef ZZ_L1P_LAZY_FLUSH
#define ZZ_L1P_LAZY_FLUSH 1
#endif
typedef __attribute__((address_space(3))) uint16_t lds_u16;

#define ZZ_L1P_NONE 64u          // win[]: no lane of the block with that hash was visited
#define ZZ_L1P_SELF 65u          // the slot lanes that are NOT cross lanes read: win[65] = 65, "my candidate is the one I read"
struct l1p_xch {
    uint32_t told[ZZ_WAVE + 4];    // per lane of block t - 1, t = the word's high half: the table's entry under the lane's hash BEFORE that block
    uint8_t win[ZZ_WAVE + 4];      // per lane of the block walked last: the highest VISITED lane of the block with the lane's hash, 64 = none; [65] = 65
    uint16_t scal[ZZ_WAVE];        // [0]: the positions by which the block walked last runs into the next one. Every lane stores
                                   // (lane l to [l]: no lane mask to set up for "lane 0 only"), only [0] is read
};   // 468 bytes



// One parsing wavefront (pw = 0: even blocks, 1: odd blocks). Barriers: B_g closes the walk of block g - 1. Per block g its
// owner runs  [P1 P2](g)  B_g  [R W](g)  B_g+1  [P4](g)  and then block g + 2; the other wavefront is one barrier out of step.
//
// BIAS = 32768 (k_encode_l1pw): the warm window (zz_level1.h, l1_packet_parser). Table entries are position + 1 +
// BIAS, the positions -32768 .. -1 in front of the packet are in the table before block 0 is probed (warm_prehash), and an entry
// further back than 32768 from the position that reads it is no candidate (encoder.cpp:348) -- decided by every READER for its own
// position: the entry itself stays what it is (it is what `told` hands to the block behind and what the repair restores).
template <uint32_t BIAS>
__device__ __forceinline__ void l1p_parse(const zz_packet_params& P, const l1_pk& pk, uint16_t* T, uint32_t* tokbuf, l1p_xch* X, const uint32_t pw)
{
    const int lane = lane_id();
    const uint64_t below_me = (1ull << lane) - 1, above_me = ~((2ull << lane) - 1), self_bit = 1ull << lane;
    const uint32_t n = pk.n;
    const l1p_src SRC = l1p_make_src(P, pk.src, pk.end);
    const uint32_t NB = (n + ZZ_WAVE - 1) >> 6;                          // blocks of the packet, n > 0
    lds_u32* const slot = (lds_u32*)tokbuf + lane + pw * ZZ_L1_TOKSLOT;   // block g's tokens go to slot g & 1
    uint64_t w = 0, w2 = 0;                                               // 16 bytes at this lane's position of the block at hand
    {
        const uint32_t p0 = pw * ZZ_WAVE + (uint32_t)lane;
        l1p_ld128<true>(SRC, p0 < n ? p0 : n - 1, w, w2);
    }
    ZZ_PROF_DECL
    // The invariant this kernel stands on, checked as it runs: where lanes of one block share a hash, the slot must end up with
    // the HIGHEST of their positions, i.e. the lane the read-back names (rb - 1 - base, computed for the same-hash sets anyway)
    // is never below the reading lane. wmin = the lowest such lane number this lane has seen over the blocks that check
    // (ZZ_L1P_ORDER_CHECK); reported at the packet's end where wmin < lane (ZZ_ERR_LDS_ORDER), and the host runs the call again
    // on k_encode_l1 (zz_api.hip encode_finish).
    uint32_t wmin = ZZ_WAVE;
    uint32_t mycout = 0;                                                  // positions by which this wavefront's last block ran into the next one
    // ZZ_L1P_EXT32: a match of "16 or more" bytes is extended by the whole wavefront when the walk gets to it -- a dependent memory
    // round trip on the packet's chain, 0.9 times per block of C source, and what log lines, XML and HTML lose 8-14 % to
    // (profiles/r05_family_rates_and_free_extension_probe.txt). Where a wavefront has just had such an event (hot_until: its next
    // blocks), the PROBER's side compares sixteen more bytes at the table candidate as read for the lanes that agree in all sixteen
    // (`ext`); the walk takes the length from there unless the lane's candidate moved or sits inside the block, and only "32 or more"
    // still asks memory. One scalar compare per block where the gate is shut (text: nearly always).
    uint32_t hot_until = 0;
    if (pw == 1) l1_group_barrier();                                      // B_0: block 0 has entered its positions
    uint32_t xlo_next = pw ? 1u + BIAS : 0xFFFF0000u;                     // block 1: base - 63 = 1; block 0: nothing can be a cross lane
    for (uint32_t g = pw; g < NB; g += 2) {
        const uint32_t tag = (g + 1) << 16;                               // told[]'s tag: "these are block g's"
        const uint32_t xlo = xlo_next;
        xlo_next = (g << 6) + (2 * ZZ_WAVE - (ZZ_WAVE - 1)) + BIAS;       // block g + 2's: its base - 63
        if (mycout >= 2 * ZZ_WAVE) {
            // A match found two blocks ago covers this block entirely (and the one between, which the other wavefront had probed by
            // then): nothing is probed, entered or walked -- the barriers, the carried match end and an empty token slot are all there
            // is. What keeps long runs (zeros: 258 bytes per match, four blocks) from costing four blocks' work each.
            const uint32_t p = (g << 6) + (uint32_t)lane;
            uint64_t wn = 0, wn2 = 0;
            if (g + 2 < NB) {
                const uint32_t pn = p + 2 * ZZ_WAVE;
                l1p_ld128<true>(SRC, pn < n ? pn : n - 1, wn, wn2);
            }
            l1_group_barrier();                                          // B_g
            const uint32_t cin = uniform((uint32_t)X->scal[0]);          // (>= 64: the block in front was covered too)
            mycout = cin > ZZ_WAVE ? cin - ZZ_WAVE : 0u;
            X->win[lane] = (uint8_t)ZZ_L1P_NONE;
            X->told[lane] = tag;
            X->scal[lane] = (uint16_t)mycout;
            l1_group_barrier();                                          // B_g+1
            *slot = 0;
            w = wn;
            w2 = wn2;
            continue;
        }
        auto block = [&](auto interior_tag) {
            // INTERIOR: every lane holds a position with 17+ bytes after it, and the look-ahead load lies inside the packet
            constexpr bool INT = decltype(interior_tag)::value;
            const uint32_t base = g << 6;
            const uint32_t p = base + (uint32_t)lane;
            const uint32_t nact = INT ? ZZ_WAVE : ((n - base) < ZZ_WAVE ? (n - base) : ZZ_WAVE);
            const bool active = INT ? true : lane < (int)nact;

            ZZ_T(6); ZZ_C(10, 1);
            // ---- P1: hash, probe + speculative insert (encoder.cpp:344-346); the candidate's bytes are requested at once
            const uint32_t h = calc_hash3((uint32_t)(w >> 8));
            const uint32_t oldraw = T[h];
            T[h] = (uint16_t)(p + 1 + BIAS);
            uint64_t wc, wc2;
            if (BIAS == 0) l1p_ld128<!INT>(SRC, __builtin_elementwise_sub_sat(oldraw, 1u), wc, wc2);     // (no candidate: the packet's first bytes, unused)
            else l1p_ld128<!INT>(SRC, (p + 1 + BIAS - oldraw > 0x8000u) ? 0 : (int32_t)(oldraw - 1u - BIAS), wc, wc2);   // (none, or out of reach: likewise)
            uint64_t wn, wn2;                                            // this lane's bytes two blocks on: blocks are fixed, so the address is known
            if (!INT) { wn = 0; wn2 = 0; }
            if (INT || g + 2 < NB) {                                     // (an interior block has two whole blocks behind it: no test)
                const uint32_t pn = p + 2 * ZZ_WAVE;
                l1p_ld128u<!INT>(SRC, INT ? pn : (pn < n ? pn : n - 1), wn, wn2);
            }
            ZZ_WAVE_SYNC();
            const uint32_t rb = T[h];                                    // the slot holds whichever lane wrote last
            // cross lanes: the entry read is a position of block g - 1, entered speculatively while that block is being walked
            // (xlo: base - 63, and for the packet's first block a value no entry can reach -- carried, one addition per block)
            const uint32_t qx = oldraw - xlo;                            // its lane there
            const bool xd = active && qx < ZZ_WAVE;
#ifdef ZZ_L1P_X_NOCROSS
            const uint64_t XD = 0;                                       // TIMING EXPERIMENT: no cross lanes at all (wrong streams)
#else
            const uint64_t XD = ballot(xd);
#endif
            // (what the lanes that are not cross lanes hold in these is never looked at: no zeroing)
            uint32_t talt;
            uint64_t wa, wa2;
            asm volatile("" : "=v"(talt), "=v"(wa), "=v"(wa2));
            // (lanes that are not cross lanes point at the sentinel slot: what they read there says "the candidate as read")
            const uint32_t qa = xd ? qx : ZZ_L1P_SELF;
            if (XD) {
                // what the table held under this hash BEFORE block g - 1: its owner publishes that, tagged with the block, as soon as
                // it has settled its own cross lanes (right after B_g-1; this wavefront has repaired block g - 2 since)
                uint32_t v;
                do {
                    ZZ_C(13, 1);
                    v = *(volatile lds_u32*)&X->told[qa];
                } while (ballot((v >> 16) != g) & XD);                   // (the scalar AND of two masks: no select, no second compare)
                talt = v & 0xFFFFu;
                if (xd) {                                                // (a gather costs the address path per lane)
                    if (BIAS == 0) l1p_ld128<!INT>(SRC, __builtin_elementwise_sub_sat(talt, 1u), wa, wa2);
                    else l1p_ld128<!INT>(SRC, (p + 1 + BIAS - talt > 0x8000u) ? 0 : (int32_t)(talt - 1u - BIAS), wa, wa2);
                }
            }

            // ---- P2: same-hash sets inside the block, lengths against every possible candidate (16 bytes compared)
            const uint64_t lostmask = ballot(active && rb != (uint32_t)(uint16_t)(p + 1 + BIAS));
            const uint32_t left = active ? n - p : 0;
            const uint32_t cap17 = INT ? 8u * (ZZ_WI_CAP + 1) : (left < ZZ_WI_CAP + 1 ? left : ZZ_WI_CAP + 1) << 3;
            uint64_t myset = self_bit;                  // the lanes of the block with my hash, myself included
            uint32_t infoB = 0;
            // (the slot was written by lanes of THIS instruction with my hash, me among them, and reads back one of them: wraw is a lane
            // number, 0..63, whatever order the LDS kept -- and wave_match6 looks at six bits only: no mask. The order check takes the
            // instruction slot the mask had; in front of the branch, so that wmin is updated in place, no copy behind a merge.)
            const uint32_t wraw = rb - 1u - BIAS - base;               // the lane whose store the slot kept
            if (INT && ZZ_L1P_ORDER_CHECK) wmin = wraw < wmin ? wraw : wmin;
            if (lostmask) {
                uint32_t W = wraw;
                if (!INT && !active) W = (uint32_t)lane;
                if (!INT && active && wraw < wmin) wmin = wraw;
                myset = wave_match6(W);
                const uint64_t below = myset & below_me;
                const bool dup = below != 0 && active;
                const uint32_t ql = 63u - (uint32_t)__builtin_clzll(below | 1ull);
                const int qa = (int)(ql << 2);
                const uint64_t wq = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)(w >> 32)) << 32) | (uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)w);
                const uint64_t wq2 = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)(w2 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)w2);
                const bool hard = (uint32_t)__builtin_popcountll(below) > 1u;
                uint32_t lbf;                                            // the LENB field with its "16 or more" flag
                if (INT) {
                    // (interior blocks: "all sixteen bytes equal" is the only way to the cap, so the cap itself can BE the field's
                    // value for that case -- length 16 and the flag -- and no compare-and-select follows the count)
                    lbf = (equal_bits128(w ^ wq, w2 ^ wq2, (((ZZ_WI_CAP << ZZ_WI_LENB_SHIFT) | ZZ_WI_EXTB) >> ZZ_WI_LENB_SHIFT) << 3) >> 3) << ZZ_WI_LENB_SHIFT;
                } else {
                    const uint32_t lb = equal_bits128(w ^ wq, w2 ^ wq2, cap17) >> 3;
                    lbf = lb > ZZ_WI_CAP ? ((ZZ_WI_CAP << ZZ_WI_LENB_SHIFT) | ZZ_WI_EXTB) : (lb << ZZ_WI_LENB_SHIFT);
                }
                const uint32_t di = ZZ_WI_DUP | (ql << ZZ_WI_QLANE_SHIFT) | (hard ? ZZ_WI_HARD : 0u) | lbf;
                infoB = dup ? di : 0u;
            }
            uint32_t ext;                                                // equal bytes 16..31 at the table candidate as read (16: all of them); see hot_until
            asm volatile("" : "=v"(ext));
            const bool gate = ZZ_L1P_EXT32 && BIAS == 0 && INT && g < hot_until;
            if (gate) {
                const bool all16 = ((w ^ wc) | (w2 ^ wc2)) == 0 && oldraw != 0;
                if (ballot(all16)) {
                    // (every lane loads: no lane mask; a lane that does not agree reads its own bytes against its own. Interior
                    // block: p + 32 <= n, and the candidate lies below p)
                    uint64_t ya, ya2, yc, yc2;
                    l1p_ld128u<false>(SRC, p + (all16 ? 16u : 0u), ya, ya2);
                    l1p_ld128u<false>(SRC, all16 ? oldraw + 15u : p, yc, yc2);           // (entry - 1 + 16)
                    ext = equal_bits128(ya ^ yc, ya2 ^ yc2, 128u) >> 3;
                }
            }
            ZZ_T(0);
            l1_group_barrier();                                          // B_g: block g - 1 has been walked
            if (ZZ_L1P_PRIO_W != ZZ_L1P_PRIO_F) __builtin_amdgcn_s_setprio(ZZ_L1P_PRIO_W);
            ZZ_T(1);

            // ---- R: which of its lanes the walk of block g - 1 visited settles the cross lanes; then ONE comparison per lane,
            // against the candidate that is the right one (this side of the barrier on purpose: the interval in which the other
            // wavefront repairs, probes and builds its same-hash sets is the longer of the two)
            uint32_t cin;
            uint32_t told, info;
            uint64_t x;                                                  // the candidate's first eight bytes XOR mine
            uint64_t MV;                                                 // lanes whose candidate is not the entry they read
            {
                // Everything between the barrier and the walk is on the packet's critical path, instruction by instruction. Both
                // reads go out at once and unconditionally: a lane that is not a cross lane reads the sentinel (win[65] = 65 = its
                // qa), and "what I read is my qa" means "the candidate is the table's entry as read" -- for it as for a cross lane
                // whose q the walk visited (q is the highest lane of its set: the LDS leaves the highest lane's store in the slot).
                // Otherwise: 64 = nobody with my hash was visited, the entry from before the block; a lower lane r: its bytes are not here.
                const uint32_t sc = X->scal[0];
                const uint32_t r = X->win[qa];
                const bool moved = r != qa;
                const bool use3 = r == ZZ_L1P_NONE;
                const uint32_t toldh = (base - ZZ_WAVE + 1u + BIAS) + r;
                const uint32_t t1 = use3 ? talt : toldh;
                told = moved ? t1 : oldraw;
                ZZ_T(7);                                                   // (diagnostic builds: the two LDS reads have come back)
                // the block behind is waiting for this (its cross lanes' second candidate): out first
                X->told[lane] = told | tag;
                MV = ballot(moved);
                const uint64_t LDM = MV & ~ballot(use3);
                // (interior blocks: the cap IS "16 | the flag", see LENB above: three instructions less between the barrier and the walk)
                constexpr uint32_t CAPA = (ZZ_WI_CAP | ZZ_WI_EXTA) << 3;
                auto compare = [&]() {
                    __builtin_amdgcn_sched_barrier(0);                   // (nothing that waits for the candidates' bytes may move in front of this)
                    const uint64_t c = use3 ? wa : wc, c2 = use3 ? wa2 : wc2;
                    x = w ^ c;
                    uint32_t la = equal_bits128(x, w2 ^ c2, INT ? CAPA : cap17) >> 3;
                    if (BIAS == 0 ? !told : (p + 1 + BIAS - told > 0x8000u)) la = 0;      // none (0: further back than anything), or out of reach
                    info = infoB | (INT ? la : (la > ZZ_WI_CAP ? (ZZ_WI_CAP | ZZ_WI_EXTA) : la));
                };
                // (ONE test of LDM, with the comparison in both arms: the test is scalar code on the critical path)
                if (__builtin_expect(LDM == 0, 1)) {
                    compare();
                } else {
                    ZZ_C(14, 1);
                    // a lower lane of block g - 1: its bytes come from the cache (that block's owner has just read them); asked for
                    // before anything else so that the comparison runs under the load (every lane loads: no lane mask to set
                    // up; the others read their own candidate's line again) -- 0.2 times per block of text, 0.35 of the mix
                    uint64_t l0, l1;
                    if (BIAS == 0) l1p_ld128u<!INT>(SRC, __builtin_elementwise_sub_sat(told, 1u), l0, l1);
                    else l1p_ld128<!INT>(SRC, (p + 1 + BIAS - told > 0x8000u) ? 0 : (int32_t)(told - 1u - BIAS), l0, l1);
                    compare();
                    const uint64_t x3 = w ^ l0;
                    const uint32_t la3 = equal_bits128(x3, w2 ^ l1, INT ? CAPA : cap17) >> 3;
                    const uint32_t info3 = infoB | (INT ? la3 : (la3 > ZZ_WI_CAP ? (ZZ_WI_CAP | ZZ_WI_EXTA) : la3));
                    x = ((uint64_t)sel_lanes(LDM, (uint32_t)(x3 >> 32), (uint32_t)(x >> 32)) << 32) | sel_lanes(LDM, (uint32_t)x3, (uint32_t)x);
                    info = sel_lanes(LDM, info3, info);
                }
                cin = uniform(sc);
            }
            ZZ_T(8);                                                       // (the comparison)
            const uint64_t E = ballot((info & (ZZ_WI_HARD | 0x1Cu | (0x1Cu << ZZ_WI_LENB_SHIFT))) != 0);
            {
                const uint32_t endl = (uint32_t)lane + (info & 31u);
                const uint64_t m = E >> (endl & 63u);
                const uint32_t nx = endl + (m ? (uint32_t)__builtin_ctzll(m) : 64u);
                info |= ((nx < 64u ? nx : 64u) & 63u) << ZZ_WI_NEXT_SHIFT;
            }

            ZZ_T(2); ZZ_C(12, (uint32_t)__builtin_popcountll(E)); ZZ_C(15, cin < 64 ? cin : 64);
            // ---- W: the walk (encoder.cpp:341-368 replayed over the event mask), starting behind the match carried in
            uint64_t mst = 0, usedB = 0;
            uint64_t cov;                                                // the lanes the match carried in covers: cin ones (all of them from 64 on)
            l1_walk_x Xw;
            Xw.hash = h; Xw.wlo = (uint32_t)w; Xw.whi = (uint32_t)(w >> 32); Xw.candbase = base + 1 + BIAS; Xw.hardok = INT ? 1u : 0u;
            // (the overrides' lanes are named by ovmL / ovmC; what the other lanes hold is never looked at: no zeroing)
            asm volatile("" : "=v"(Xw.ovlen), "=v"(Xw.ovcand1));
            Xw.ovmL = 0; Xw.ovmC = 0;
            uint32_t& ovlen = Xw.ovlen; uint32_t& ovcand1 = Xw.ovcand1;
            uint64_t& ovmL = Xw.ovmL; uint64_t& ovmC = Xw.ovmC;
            uint32_t pos = cin;
            // (ONE test of "the carried match covers the whole block" for both the cover mask and the walk's entry)
            if (cin < ZZ_WAVE) {
                asm("s_bfm_b64 %0, %1, 0" : "=s"(cov) : "s"(cin));
                l1_fast_walk<true>(E, info, nact, pos, mst, cov, usedB, Xw);     // (<true>: pos < 64 is known, no test of its own)
            } else {
                cov = ~0ull;
            }
            while (pos < nact) {
                const int e = (int)pos;
                ZZ_C(11, 1);
                const uint64_t probed = ~cov | mst;
                const uint32_t inf = readlane(info, e);
                const uint32_t pe = base + (uint32_t)e;
                const uint32_t maxlen = (n - pe) < ZZ_MAX_LEN ? (n - pe) : ZZ_MAX_LEN;
                uint32_t mlen;
                if (!(inf & ZZ_WI_HARD)) {
                    const bool useB = (inf & ZZ_WI_DUP) && ((probed >> ZZ_WI_QLANE(inf)) & 1);
                    mlen = useB ? ZZ_WI_LENB(inf) : ZZ_WI_LENA(inf);
                    if (mlen >= 4) {
                        if (inf & (useB ? ZZ_WI_EXTB : ZZ_WI_EXTA)) {     // remain(), encoder.cpp:64-90
                            const int32_t cand = useB ? (int32_t)(base + ZZ_WI_QLANE(inf)) : (int32_t)(readlane(told, e) - 1u - BIAS);
#ifdef ZZ_L1P_X_NOEXT
                            (void)cand;                                  // TIMING EXPERIMENT (valid but WRONG streams): "16 or more" is 16, no extension loads
                            mlen = maxlen < ZZ_WI_CAP ? maxlen : ZZ_WI_CAP;
#else
                            // (the prober's sixteen further bytes hold for the entry as this lane read it: not for a candidate inside
                            // the block, not where the walk of the block in front moved it)
                            const uint32_t e16 = (gate && !useB && !((MV >> e) & 1)) ? readlane(ext, e) : 0xFFu;
                            if (e16 < 16u) mlen = 16u + e16;             // (an interior block: 32 bytes lie inside the packet)
                            else mlen = l1p_extend_match(SRC, pe, cand, maxlen, e16 == 16u ? 32u : ZZ_WI_CAP);
                            if (ZZ_L1P_EXT32) hot_until = g + 8u;
#endif
                            if (lane == e) ovlen = mlen | 0x8000u;
                            ovmL |= 1ull << e;
                        }
                        if (useB) usedB |= 1ull << e;
                    }
                } else {
                    // candidate = most recent visited lane of this block with my hash, else the table's
                    const uint64_t S = readlane64(myset, e) & probed & ((1ull << e) - 1);
                    uint32_t cand1 = 0;
                    uint64_t xe = ~0ull;
                    if (S) {
                        const int c = 63 - __builtin_clzll(S);
                        cand1 = base + (uint32_t)c + 1 + BIAS;
                        xe = readlane64(w, e) ^ readlane64(w, c);
                    } else {
                        cand1 = readlane(told, e);
                        if (BIAS && pe + 1 + BIAS - cand1 > 0x8000u) cand1 = 0;      // out of reach (encoder.cpp:348)
                        if (cand1) {
                            xe = readlane64(x, e);
                        }
                    }
                    mlen = 0;
                    if ((uint32_t)xe == 0 && maxlen >= 4) {
                        if (xe != 0) mlen = (uint32_t)__builtin_ctzll(xe) >> 3;
#ifdef ZZ_L1P_X_NOEXT
                        else mlen = 8;
#else
                        else mlen = l1p_extend_match(SRC, pe, (int32_t)(cand1 - 1u - BIAS), maxlen);
#endif
                        if (mlen > maxlen) mlen = maxlen;
                    }
                    if (lane == e) { ovlen = mlen | 0x8000u; ovcand1 = cand1; }
                    ovmL |= 1ull << e; ovmC |= 1ull << e;
                }
                if (mlen > 3) {                                          // encoder.cpp:356
                    mst |= 1ull << e;
                    cov |= (mlen >= 64u - (uint32_t)e) ? (~0ull << e) : (((1ull << mlen) - 1) << e);
                    pos = (uint32_t)e + mlen;                            // encoder.cpp:361-362
                } else {
                    pos = (uint32_t)e + 1;                               // a literal after all (encoder.cpp:367)
                }
                l1_fast_walk<false>(E, info, nact, pos, mst, cov, usedB, Xw);
            }
            ZZ_T(3);
            // visited lanes: every lane in front of `pos` that no match covers, plus the match starts
            uint64_t committed;
            if (INT) {
                // (an interior block's walk ends at or behind lane 63: "in front of pos" is every lane -- one instruction, not five)
                asm("s_orn2_b64 %0, %2, %1" : "=s"(committed) : "s"(cov), "s"(mst) : "scc");
            } else {
                uint32_t t;
                asm("s_min_u32 %1, %2, 64\n\ts_sub_u32 %1, 64, %1\n\ts_lshr_b64 %0, -1, %1\n\ts_andn2_b64 %0, %0, %3\n\ts_or_b64 %0, %0, %4"
                    : "=&s"(committed), "=&s"(t) : "s"(pos), "s"(cov), "s"(mst) : "scc");
            }
            {
                // for the block behind: per lane the highest visited lane with its hash; the match end carried over
                // (v_ffbh gives -1 for 0 and the addition saturates: 63 - min is 64 = "none" by itself, no compare, no select)
                const uint64_t sv = myset & committed;
                const uint32_t fh = ffbh_or_ones((uint32_t)(sv >> 32)), fl = add_sat_k<32>(ffbh_or_ones((uint32_t)sv));
                X->win[lane] = (uint8_t)(63u - (fh < fl ? fh : fl));
                mycout = pos > ZZ_WAVE ? pos - ZZ_WAVE : 0u;
                X->scal[lane] = (uint16_t)mycout;
            }
            ZZ_T(4);
            l1_group_barrier();                                          // B_g+1: block g has been walked
            if (ZZ_L1P_PRIO_W != ZZ_L1P_PRIO_F) __builtin_amdgcn_s_setprio(ZZ_L1P_PRIO_F);
            ZZ_T(5);

            // ---- P4: table repair where the slot still holds a position of this block (block g + 1 has entered its own since)
            // (the tokens go to the slot only now: the emitter reads block g - 2's from it after B_g, with nothing but its own
            // pace between that barrier and the read)
            const uint32_t rb2 = T[h];
            {
                const uint32_t la_ = ZZ_WI_LENA(info) | 0x8000u, lb_ = ZZ_WI_LENB(info) | 0x8000u;     // ZZ_TOK_MATCH >> 16 rides along
                const uint32_t tl = sel_lanes(ovmL, ovlen, sel_lanes(usedB, lb_, la_));
                const uint32_t cn = sel_lanes(ovmC, ovcand1, sel_lanes(usedB, base + ZZ_WI_QLANE(info) + 1 + BIAS, told));
                const uint32_t tmatch = (tl << 16) | (p + 1 + BIAS - cn);
                const uint32_t tlit = ZZ_TOK_LIT | (uint32_t)(w & 0xFF);
                *slot = keep_lanes(committed, sel_lanes(mst, tmatch, tlit));
            }
            const uint64_t INB = ballot((rb2 - (base + 1 + BIAS)) < ZZ_WAVE);
            const uint32_t taddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)(T + h);
            {
                const uint64_t rm = INB & ~committed;                    // skipped lanes restore the old entry
                uint64_t saved;
                asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write_b16 %2, %3\n\ts_mov_b64 exec, %0"
                             : "=&s"(saved) : "s"(rm), "v"(taddr), "v"(told) : "memory", "scc");
            }
            if (lostmask) {                                              // among visited lanes sharing a hash the highest position wins
                ZZ_WAVE_SYNC();
                const uint64_t wm = ballot((myset & committed & above_me) == 0) & committed & INB;
                uint64_t saved;
                asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write_b16 %2, %3\n\ts_mov_b64 exec, %0"
                             : "=&s"(saved) : "s"(wm), "v"(taddr), "v"(p + 1 + BIAS) : "memory", "scc");
            }
            ZZ_WAVE_SYNC();
            w = wn;
            w2 = wn2;
        };
        if ((g << 6) + 3 * ZZ_WAVE + 15 <= n) block(std::true_type{});
        else block(std::false_type{});
    }
    if (((NB - 1) & 1u) != pw) l1_group_barrier();                       // the other wavefront's last walk
    l1_group_barrier();                                                  // hand-over of the last block's tokens
    if ((ballot(wmin < (uint32_t)lane) != 0 || P.dbg_viol) && lane == 0) atomicOr(P.err, ZZ_ERR_LDS_ORDER);
    ZZ_PROF_FLUSH_W(P, pw);
}

template <uint32_t BIAS>
__device__ __forceinline__ void l1p_packet_parser(const zz_packet_params& P, uint32_t k, uint16_t* T, uint32_t* tokbuf, l1p_xch* X, uint32_t pw)
{
    const int lane = lane_id();
    const l1_pk q = l1_packet_of(P, k);
    __builtin_amdgcn_s_setprio(ZZ_L1P_PRIO_F);
    // cold table (encoder.cpp:533-536): each parser clears its half
    uint4* t4 = (uint4*)T + pw * (ZZ_HASH_SIZE * sizeof(uint16_t) / 32);
    for (int i = lane; i < (int)(ZZ_HASH_SIZE * sizeof(uint16_t) / 32); i += ZZ_WAVE) t4[i] = make_uint4(0, 0, 0, 0);
    if (pw == 0) X->scal[lane] = 0;                                      // block 0: nothing carried in
    if (pw == 0) X->told[lane] = 0;                                      // (tag 0: no block's)
    if (pw == 1 && lane == 0) X->win[ZZ_L1P_SELF] = (uint8_t)ZZ_L1P_SELF; // the sentinel (zz_level1p.h, R)
    if (!BIAS && ZZ_L1P_ADLER3 && P.cks_kind == ZZ_CKS_ADLER) {
        // a third of the packet's Adler-32 sums (the emitter takes the last third and puts them together behind the barrier): summed by
        // the emitter alone, the packet began with both parsers waiting for it -- 48 K cycles of a packet's 1.4 M (instrumented)
        uint32_t A = 0;
        uint64_t C = 0;
        wave_adler_part<ZZ_L1P_ADLER_U>(q.src, q.len, pw, 3, A, C);
        const uint64_t At = wave_sum64(A), Ct = wave_sum64(C);
        if (lane == 0) { uint64_t* part = (uint64_t*)(tokbuf + ZZ_L1_TOKSLOT) + 2 * pw; part[0] = At; part[1] = Ct; }   // (the second hand-over slot: free until block 1's tokens)
    }
    if (BIAS) {
        // the warm window: the first parser enters the last P.warm bytes in front of the packet (zz_level1.h warm_prehash: ascending,
        // the LDS's own order leaves the highest position per hash -- checked as it goes) between two barriers, while the emitter
        // sums the packet's Adler-32; the second parser's half of the table must be clear before the first entry lands
        l1_group_barrier();                                              // B_c: the table is clear
        if (pw == 0) {
            const uint64_t before = P.halo + q.off;                      // input bytes of this stream in front of the packet
            uint32_t viol = 0;
            // key of a position: bytes pos+1..pos+3 (encoder.cpp:344); spare slot: the last half word of the hand-over slots, unused so far
            warm_prehash<BIAS, true>(T, q.src, (int32_t)(before < P.warm ? before : P.warm), q.end, 1, (uint32_t)(((uint16_t*)tokbuf + 255) - T), viol);
            if ((ballot(viol != 0) != 0 || P.dbg_viol) && lane == 0) atomicOr(P.err, ZZ_ERR_LDS_ORDER);
        }
    }
    l1_group_barrier();                                                  // B_z
    if (q.n > 0) {
        l1p_parse<BIAS>(P, q, T, tokbuf, X, pw);
    }
}

template <uint32_t BIAS>
__device__ __forceinline__ void l1p_packet_emitter(const zz_packet_params& P, uint32_t k, uint32_t* ring_words, const uint32_t* tokbuf)
{
    const int lane = lane_id();
    const l1_pk q = l1_packet_of(P, k);
    bitring ring;
    // (every append of this wavefront goes the same way: the batched flush must not meet the unbatched one's "at most 64 words wait")
    auto append_uniform = [&](uint32_t bits, uint32_t nbits) {
        if (ZZ_L1P_LAZY_FLUSH) ring_append_lazy(ring, lane == 0 ? bits : 0u, lane == 0 ? nbits : 0u);
        else ring_append_uniform(ring, bits, nbits);
    };
    if (ZZ_L1P_PRIO_E) __builtin_amdgcn_s_setprio(ZZ_L1P_PRIO_E);
    ZZ_PROF_DECL                                                         // (diagnostic builds: [2..6] per packet, [0], [1] per block, tools/prof_l1p.py)
    ring_init(ring, ring_words, q.out);
    const bool adler3 = !BIAS && ZZ_L1P_ADLER3 && P.cks_kind == ZZ_CKS_ADLER;
    uint64_t At = 0, Ct = 0;
    if (adler3) {
        uint32_t A = 0;
        uint64_t C = 0;
        wave_adler_part<ZZ_L1P_ADLER_U>(q.src, q.len, 2, 3, A, C);
        At = wave_sum64(A); Ct = wave_sum64(C);
    }
    if (BIAS) l1_group_barrier();                                        // B_c
    else l1_group_barrier();                                             // B_z
    ZZ_T(2);
    if (adler3) {
        if (lane == 0) {
            const uint64_t* part = (const uint64_t*)(tokbuf + ZZ_L1_TOKSLOT);
            At += part[0] + part[2]; Ct += part[1] + part[3];
            zz_cks c;
            c.a = (uint32_t)(At % ZZ_ADLER_MOD);
            c.b = (uint32_t)(((uint64_t)q.len * At - Ct) % ZZ_ADLER_MOD);
            P.cks[k] = c;
        }
    } else if (P.cks_kind == ZZ_CKS_ADLER) {                             // BIAS: while the first parser enters the window
        zz_cks c = wave_adler(q.src, q.len);
        if (lane == 0) P.cks[k] = c;
    }
    ZZ_T(3);
    if (BIAS) l1_group_barrier();                                        // B_z
    if (q.n > 0) {
        append_uniform((q.is_final ? 1u : 0u) | (1u << 1), 3);           // StartBlock(FixedHuffman, final): encoder.cpp:143-147,338
        const uint32_t NB = (q.n + ZZ_WAVE - 1) >> 6;
        const lds_u32* slot = (const lds_u32*)tokbuf + lane;
        l1_group_barrier();                                              // B_0
        ZZ_T(4);
        l1_group_barrier();                                              // B_1
        ZZ_T(5); ZZ_C(11, 1);
        for (uint32_t g = 0; g < NB; ++g) {
            ZZ_T(1); ZZ_C(10, 1);                                        // (diagnostic builds: [1] = emitting, [0] = asleep in the barrier)
            l1_group_barrier();                                          // B_g+2: block g's tokens are in slot g & 1
            ZZ_T(0);
            const uint32_t tok = *slot;
            slot = lds_flip_slot((lds_u32*)slot);
#ifdef ZZ_L1P_X_NOEMIT
            if (g == 0)                                                  // TIMING EXPERIMENT: the emitter only keeps the barriers (wrong streams)
#endif
#if ZZ_L1P_LAZY_FLUSH
            { uint32_t bits, nb; l1_token_bits(tok, bits, nb); ring_append_lazy(ring, bits, nb); }
#else
            l1_emit_tokens(ring, nullptr, tok);
#endif
        }
        append_uniform(0, 7);                                 // EOB: codes_f[256] (encoder.cpp:371)
    }
    if (!q.is_final) {
        // SetLevel(0); AddData(e-1, e): one stored byte = byte alignment (zzflate.cpp:118-120, encoder.cpp:482-502)
        append_uniform(0, 3);
        ring_pad_to_byte(ring);
        append_uniform(0xFFFE0001u, 32);
        append_uniform(q.src[q.len - 1], 8);
    } else if (q.n == 0) {
        append_uniform(1u | (1u << 1), 3);                    // empty final packet: one empty fixed block (D8)
        append_uniform(0, 7);
    }
    const uint32_t bytes = ZZ_L1P_LAZY_FLUSH ? ring_finish_lazy(ring) : ring_finish(ring);
    if (lane == 0) {
        P.sizes[k] = bytes;
        if (bytes > P.slot_stride) atomicOr(P.err, ZZ_ERR_SLOT_OVERFLOW);
    }
    ZZ_T(6);
    ZZ_PROF_FLUSH_W(P, 2);
}

__global__ __launch_bounds__(ZZ_L1P_THREADS) void k_encode_l1p(zz_packet_params P)
{
    __shared__ uint16_t T[ZZ_HASH_SIZE];          // hashtable (encoder.h:76) as pos+1, 0 = empty
    __shared__ uint32_t ring_words[ZZ_RING_WORDS];
    __shared__ __attribute__((aligned(512))) uint32_t tokbuf[2 * ZZ_L1_TOKSLOT];
    __shared__ l1p_xch X;
    const uint32_t k = blockIdx.x;
    const uint32_t wv = uniform(threadIdx.x >> 6);
    if (wv < 2) l1p_packet_parser<0u>(P, k, T, tokbuf, &X, wv);
    else l1p_packet_emitter<0u>(P, k, ring_words, tokbuf);
}
// the same with a warm window (P.warm > 0): table entries position + 1 + 32768, the window entered before block 0 is probed
__global__ __launch_bounds__(ZZ_L1P_THREADS) void k_encode_l1pw(zz_packet_params P)
{
    __shared__ uint16_t T[ZZ_HASH_SIZE];
    __shared__ uint32_t ring_words[ZZ_RING_WORDS];
    __shared__ __attribute__((aligned(512))) uint32_t tokbuf[2 * ZZ_L1_TOKSLOT];
    __shared__ l1p_xch X;
    const uint32_t k = blockIdx.x;
    const uint32_t wv = uniform(threadIdx.x >> 6);
    if (wv < 2) l1p_packet_parser<32768u>(P, k, T, tokbuf, &X, wv);
    else l1p_packet_emitter<32768u>(P, k, ring_words, tokbuf);
}

}  // namespace zz
