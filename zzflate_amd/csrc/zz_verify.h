// zz_verify.h -- self-verification at scale: inflate every packet of a finished stream on the
// device and compare with the input it was made from. The reference has no decoder (decoder.h is an empty stub) and
// host zlib manages ~0.4 GB/s per core, so this is what checks multi-GiB outputs exhaustively.
//
// Packets are independent, byte-aligned runs of complete DEFLATE blocks (zzflate.cpp:101-125), so one LANE decodes one
// packet: a plain serial RFC 1951 decoder (stored, fixed and dynamic blocks; canonical codes decoded bit by bit from
// per-length counts). No output is produced: a literal is compared with the input byte, a match (length, distance)
// is accepted iff the input repeats itself accordingly -- the bytes in front were verified already, so that is
// equivalent to comparing the decoded bytes. Test infrastructure for the product path, not part of it.
#pragma once
#include "zz_common.h"

namespace zz {

struct zz_verify_params {
    const uint8_t* src; uint64_t n;           // the input of the encode call
    uint64_t halo;                            // input bytes of the same stream in front of src (shards)
    uint32_t packet_size, npk; int last_is_final;
    const uint8_t* stream;                    // compacted DEFLATE bytes (behind the container header)
    const uint64_t* offsets; const uint32_t* sizes;   // per packet; null at level 0 (sizes follow from the level)
    uint32_t l0_stride;                       // level 0: bytes per full packet
    uint64_t stream_bytes;
    unsigned long long* out;                  // [0] packets that failed, [1] lowest failing packet
};

struct vbits {
    const uint8_t* p; uint32_t nbytes, bitpos; bool err;
    __device__ uint32_t get(uint32_t n)       // n <= 16, LSB-first
    {
        if (n == 0) return 0;
        if (bitpos + n > nbytes * 8) { err = true; return 0; }
        const uint32_t b = bitpos >> 3, s = bitpos & 7;
        uint32_t w = p[b];
        if (b + 1 < nbytes) w |= (uint32_t)p[b + 1] << 8;
        if (b + 2 < nbytes) w |= (uint32_t)p[b + 2] << 16;
        bitpos += n;
        return (w >> s) & ((1u << n) - 1);
    }
};
struct vhuff { uint16_t count[16]; uint16_t symbol[288]; };

// canonical code from code lengths (RFC 1951 3.2.2): symbols sorted by (length, symbol); false if over-subscribed
__device__ inline bool vbuild(vhuff& h, const uint8_t* lens, int n)
{
    for (int i = 0; i < 16; ++i) h.count[i] = 0;
    for (int i = 0; i < n; ++i) h.count[lens[i]]++;
    int left = 1;
    for (int l = 1; l < 16; ++l) { left = (left << 1) - h.count[l]; if (left < 0) return false; }
    uint16_t offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + h.count[l];
    for (int i = 0; i < n; ++i) if (lens[i]) h.symbol[offs[lens[i]]++] = (uint16_t)i;
    return true;
}
// one symbol: codes of a length are consecutive integers, so walk the lengths with the first code of each
__device__ inline int vdecode(vbits& b, const vhuff& h)
{
    int code = 0, first = 0, index = 0;
    for (int l = 1; l < 16; ++l) {
        code |= (int)b.get(1);
        if (b.err) return -1;
        const int cnt = h.count[l];
        if (code - cnt < first) return h.symbol[index + (code - first)];
        index += cnt; first += cnt;
        first <<= 1; code <<= 1;
    }
    return -1;
}

__device__ inline bool vcheck_packet(const uint8_t* in, uint32_t in_len, const uint8_t* ref, uint32_t ref_len, uint64_t before,
                                     bool expect_final)
{
    vbits b; b.p = in; b.nbytes = in_len; b.bitpos = 0; b.err = false;
    uint32_t out = 0;
    bool saw_final = false;
    vhuff lit, dst;
    uint8_t lens[320];
    while (((b.bitpos + 7) >> 3) < in_len) {
        if (saw_final) return false;                                   // bytes behind the final block
        const uint32_t bfinal = b.get(1), type = b.get(2);
        if (b.err || type == 3) return false;
        saw_final = bfinal != 0;
        if (type == 0) {
            b.bitpos = (b.bitpos + 7) & ~7u;
            const uint32_t ln = b.get(16), nl = b.get(16);
            if (b.err || (ln ^ nl) != 0xFFFF) return false;
            const uint32_t at = b.bitpos >> 3;
            if (at + ln > in_len || out + ln > ref_len) return false;
            for (uint32_t i = 0; i < ln; ++i) if (in[at + i] != ref[out + i]) return false;
            b.bitpos += ln * 8; out += ln;
            continue;
        }
        if (type == 1) {
            for (int i = 0; i < 288; ++i) lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
            vbuild(lit, lens, 288);
            for (int i = 0; i < 30; ++i) lens[i] = 5;
            vbuild(dst, lens, 30);
        } else {
            const uint32_t hlit = b.get(5) + 257, hdist = b.get(5) + 1, hclen = b.get(4) + 4;
            if (b.err || hlit > 286 || hdist > 30) return false;
            const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
            for (int i = 0; i < 19; ++i) lens[i] = 0;
            for (uint32_t i = 0; i < hclen; ++i) lens[order[i]] = (uint8_t)b.get(3);
            if (b.err || !vbuild(lit, lens, 19)) return false;          // `lit` holds the code-length code for a moment
            uint32_t idx = 0;
            while (idx < hlit + hdist) {
                const int sym = vdecode(b, lit);
                if (sym < 0) return false;
                if (sym < 16) { lens[idx++] = (uint8_t)sym; continue; }
                uint32_t rep, val = 0;
                if (sym == 16) { if (idx == 0) return false; val = lens[idx - 1]; rep = 3 + b.get(2); }
                else if (sym == 17) rep = 3 + b.get(3);
                else rep = 11 + b.get(7);
                if (b.err || idx + rep > hlit + hdist) return false;
                while (rep--) lens[idx++] = (uint8_t)val;
            }
            uint8_t dl[30];
            for (uint32_t i = 0; i < hdist; ++i) dl[i] = lens[hlit + i];
            if (!vbuild(lit, lens, (int)hlit) || !vbuild(dst, dl, (int)hdist)) return false;
        }
        for (;;) {
            const int sym = vdecode(b, lit);
            if (sym < 0) return false;
            if (sym < 256) {
                if (out >= ref_len || ref[out] != (uint8_t)sym) return false;
                out++;
                continue;
            }
            if (sym == 256) break;
            if (sym > 285) return false;
            uint32_t len;                                                // RFC 1951 3.2.5
            if (sym < 265) len = (uint32_t)sym - 254;
            else if (sym == 285) len = 258;
            else { const uint32_t eb = ((uint32_t)sym - 261) >> 2; len = 3 + ((4 | (((uint32_t)sym - 265) & 3)) << eb) + b.get(eb); }
            const int ds = vdecode(b, dst);
            if (ds < 0 || ds > 29) return false;
            uint32_t dist;
            if (ds < 4) dist = (uint32_t)ds + 1;
            else { const uint32_t eb = ((uint32_t)ds >> 1) - 1; dist = 1 + ((2 | ((uint32_t)ds & 1)) << eb) + b.get(eb); }
            // level >= 2 may extend a match backward into the previous packet's bytes (encoder.cpp:404-416): the
            // distance then reaches in front of the packet, which is fine inside the stream
            if (b.err || dist > out + before || out + len > ref_len) return false;
            for (uint32_t i = 0; i < len; ++i) if (ref[out + i] != ref[(int64_t)out - (int64_t)dist + i]) return false;
            out += len;
        }
    }
    return !b.err && out == ref_len && saw_final == expect_final;
}

__global__ __launch_bounds__(64) void k_verify_packets(zz_verify_params V)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= V.npk) return;
    const uint64_t off = k * V.packet_size;
    const uint32_t len = (uint32_t)((V.n - off) < V.packet_size ? (V.n - off) : V.packet_size);
    uint64_t at; uint32_t sz;
    if (V.offsets) { at = V.offsets[k]; sz = V.sizes[k]; }
    else { at = k * V.l0_stride; sz = (uint32_t)((V.stream_bytes - at) < V.l0_stride ? (V.stream_bytes - at) : V.l0_stride); }
    bool ok = at + sz <= V.stream_bytes;
    if (ok) ok = vcheck_packet(V.stream + at, sz, V.src + off, len, V.halo + off, V.last_is_final && k == V.npk - 1);
    if (!ok) { atomicAdd(&V.out[0], 1ull); atomicMin(&V.out[1], (unsigned long long)k); }
}

}  // namespace zz
