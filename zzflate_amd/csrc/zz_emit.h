// zz_emit.h -- LSB-first bit packing for one wavefront.
//
// Replaces outputbitstream.h:83-124 (a serial 64-bit accumulator) with a wave-parallel packer: every lane
// holds one variable-length fragment (<= 32 bits), a DPP prefix scan turns fragment lengths into bit
// offsets, fragments are OR-ed into a small LDS ring of 32-bit words with ds_or_b32, and completed words
// leave the ring as coalesced dword stores. The bit order is the reference's (A.1 in SURVEY.md): codes
// pre-reversed, extra bits little-endian, byte i of the stream = bits 8i..8i+7.
#pragma once
#include "zz_wave.h"

namespace zz {

#define ZZ_RING_WORDS 128   // power of two; one append may add at most 64*32 bits = 64 words

struct bitring {
    uint32_t* ring;      // LDS, ZZ_RING_WORDS words, all-zero outside the live window
    uint32_t* out32;     // destination (4-byte aligned)
    uint32_t bitpos;     // bits appended so far
    uint32_t flushed;    // whole words already stored to out32
    uint32_t hold;       // level 2, second emitter: this word goes to *holdp instead of out32 (~0u: none)
    uint32_t* holdp;
};

__device__ __forceinline__ void ring_init(bitring& r, uint32_t* lds_ring, uint8_t* out)
{
    r.ring = lds_ring;
    r.out32 = (uint32_t*)out;
    r.bitpos = 0;
    r.flushed = 0;
    r.hold = ~0u;
    r.holdp = nullptr;
    for (int i = lane_id(); i < ZZ_RING_WORDS; i += ZZ_WAVE) lds_ring[i] = 0;
    ZZ_WAVE_SYNC();
}

// store the words completed so far; caller guarantees at most 64 are pending
__device__ __forceinline__ void ring_flush_full(bitring& r)
{
    ZZ_WAVE_SYNC();
    const uint32_t full = r.bitpos >> 5;
    const uint32_t w = r.flushed + lane_id();
    if (w < full) {
        uint32_t v = r.ring[w & (ZZ_RING_WORDS - 1)];
        r.ring[w & (ZZ_RING_WORDS - 1)] = 0;
        r.out32[w] = v;
    }
    r.flushed = full;
    ZZ_WAVE_SYNC();
}

// every lane appends `nb` bits (0 = nothing; nb <= 32, value already masked) in lane order
__device__ __forceinline__ void ring_append(bitring& r, uint32_t bits, uint32_t nb)
{
    const uint32_t incl = wave_scan_incl(nb);
    const uint32_t total = readlane(incl, 63);
    const uint32_t o = r.bitpos + incl - nb;
    const uint32_t sh = o & 31;
    const uint32_t w = o >> 5;
    if (nb) {
        atomicOr(&r.ring[w & (ZZ_RING_WORDS - 1)], bits << sh);
        if (sh + nb > 32) atomicOr(&r.ring[(w + 1) & (ZZ_RING_WORDS - 1)], bits >> (32 - sh));
    }
    r.bitpos += total;
    ring_flush_full(r);
}

__device__ __forceinline__ void ring_pad_to_byte(bitring& r);
// The same with the stores batched: completed words leave the ring only once 32 of them are waiting (one coalesced store of up to
// 64 words instead of one of ~6 per append: level 1's emitter appends ~200 bits per block of text). At most 31 words wait in front
// of an append and one append adds at most 63 (64 lanes x 31 bits), so one pass of 64 lanes empties the ring far enough and the
// 128-word ring never wraps onto live words. ring_finish_lazy drains whatever is left.
__device__ __forceinline__ void ring_append_lazy(bitring& r, uint32_t bits, uint32_t nb)
{
    const uint32_t incl = wave_scan_incl(nb);
    const uint32_t total = readlane(incl, 63);
    const uint32_t o = r.bitpos + incl - nb;
    const uint32_t sh = o & 31;
    const uint32_t w = o >> 5;
    if (nb) {
        atomicOr(&r.ring[w & (ZZ_RING_WORDS - 1)], bits << sh);
        if (sh + nb > 32) atomicOr(&r.ring[(w + 1) & (ZZ_RING_WORDS - 1)], bits >> (32 - sh));
    }
    r.bitpos += total;
    const uint32_t full = r.bitpos >> 5;
    if (full - r.flushed >= 32u) {
        ZZ_WAVE_SYNC();
        const uint32_t upto = full - r.flushed > 64u ? r.flushed + 64u : full;
        const uint32_t wi = r.flushed + lane_id();
        if (wi < upto) {
            const uint32_t v = r.ring[wi & (ZZ_RING_WORDS - 1)];
            r.ring[wi & (ZZ_RING_WORDS - 1)] = 0;
            r.out32[wi] = v;
        }
        r.flushed = upto;
        ZZ_WAVE_SYNC();
    }
}
__device__ __forceinline__ uint32_t ring_finish_lazy(bitring& r)
{
    ring_pad_to_byte(r);
    const uint32_t bytes = r.bitpos >> 3;
    const uint32_t words = (bytes + 3) >> 2;
    ZZ_WAVE_SYNC();
    while (r.flushed < words) {                          // (at most two passes: fewer than 96 words wait)
        const uint32_t w = r.flushed + lane_id();
        if (w < words) r.out32[w] = r.ring[w & (ZZ_RING_WORDS - 1)];
        r.flushed += ZZ_WAVE;
    }
    return bytes;
}

// wave-uniform append of up to 32 bits (all lanes pass the same values)
__device__ __forceinline__ void ring_append_uniform(bitring& r, uint32_t bits, uint32_t nb)
{
    ring_append(r, lane_id() == 0 ? bits : 0u, lane_id() == 0 ? nb : 0u);
}

__device__ __forceinline__ void ring_pad_to_byte(bitring& r)   // outputbitstream.h:100-103
{
    r.bitpos = (r.bitpos + 7) & ~7u;
}

// outputbitstream.h:105-124 Flush: pad to a byte and drain. Returns the stream length in bytes. The
// final partial word is stored whole (slots are padded so this never leaves the slot).
__device__ __forceinline__ uint32_t ring_finish(bitring& r)
{
    ring_pad_to_byte(r);
    const uint32_t bytes = r.bitpos >> 3;
    const uint32_t words = (bytes + 3) >> 2;
    ZZ_WAVE_SYNC();
    const uint32_t w = r.flushed + lane_id();
    if (w < words) r.out32[w] = r.ring[w & (ZZ_RING_WORDS - 1)];
    return bytes;
}

// ---- fixed-Huffman and length/distance symbol arithmetic ---------------------------------------------
// Tables of the reference (luts.cpp, fixedhuffmanluts.cpp) are RFC 1951 3.2.5/3.2.6 data; on the GPU the
// symbol, extra-bit count and extra value of a length or distance are computed with clz instead of a
// 32 KiB LUT.

// length 3..258 -> symbol 257..285, extra bit count, extra value   (luts.cpp:5-58)
__host__ __device__ inline void length_symbol(uint32_t len, uint32_t& sym, uint32_t& eb, uint32_t& ev)
{
    uint32_t l = len - 3;
    if (len == 258) { sym = 285; eb = 0; ev = 0; return; }
    if (l < 8) { sym = 257 + l; eb = 0; ev = 0; return; }
    uint32_t k = 31 - __builtin_clz(l);
    eb = k - 2;
    sym = 257 + 4 * eb + (l >> eb);
    ev = l & ((1u << eb) - 1);
}
// distance 1..32768 -> bucket 0..29, extra bit count, extra value   (luts.cpp:64,79-1160)
__host__ __device__ inline void dist_symbol(uint32_t dist, uint32_t& bucket, uint32_t& eb, uint32_t& ev)
{
    uint32_t d = dist - 1;
    if (d < 4) { bucket = d; eb = 0; ev = 0; return; }
    uint32_t k = 31 - __builtin_clz(d);
    eb = k - 1;
    bucket = 2 * k + ((d >> eb) & 1);
    ev = d & ((1u << eb) - 1);
}
__host__ __device__ inline uint32_t bitrev(uint32_t v, uint32_t n)   // huffman.cpp:11-33
{
#ifdef __HIP_DEVICE_COMPILE__
    return __builtin_bitreverse32(v) >> (32 - n);
#else
    uint32_t r = 0;
    for (uint32_t i = 0; i < n; ++i) r |= ((v >> i) & 1u) << (n - 1 - i);
    return r;
#endif
}
// fixed code of literal/length symbol 0..287 (RFC 1951 3.2.6; fixedhuffmanluts.cpp:5), pre-reversed
__host__ __device__ inline void fixed_code(uint32_t sym, uint32_t& bits, uint32_t& nb)
{
    if (sym < 144) { bits = bitrev(0x30 + sym, 8); nb = 8; }
    else if (sym < 256) { bits = bitrev(0x190 + (sym - 144), 9); nb = 9; }
    else if (sym < 280) { bits = bitrev(sym - 256, 7); nb = 7; }
    else { bits = bitrev(0xC0 + (sym - 280), 8); nb = 8; }
}
// merged length code (symbol code, then extra bits above it): fixedhuffmanluts.cpp:8-46 / encoder.cpp:121-133
__host__ __device__ inline uint32_t fixed_lcode_packed(uint32_t len)   // (nbits << 16) | bits
{
    uint32_t sym, eb, ev, bits, nb;
    length_symbol(len, sym, eb, ev);
    fixed_code(sym, bits, nb);
    return ((nb + eb) << 16) | (ev << nb) | bits;
}

}  // namespace zz
