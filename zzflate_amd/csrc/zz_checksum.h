// zz_checksum.h -- Adler-32 and CRC-32 as per-packet partials plus order-preserving combines.
//
// Replaces zzflate.cpp:170-192 (AppendChecksum), adler.cpp:5-43 and crc.cpp:5-33: the reference makes a
// second scalar pass over the whole input; here every packet's partial is produced by the wave that
// encodes the packet (Adler, fused into the encode kernels) or by a sibling kernel (CRC), and partials
// are folded in packet order with `combine` semantics (adler.cpp:5-15) / GF(2) polynomial shifts.
#pragma once
#include "zz_wave.h"

#define ZZ_ADLER_MOD 65521u
#define ZZ_CRC_POLY 0xEDB88320u

namespace zz {

// ---- GF(2) helpers (host + device) ------------------------------------------------------------------
// a(x)*b(x) mod P(x), bit-reflected representation (bit 31 = x^0)
__host__ __device__ inline uint32_t gf2_mulmod(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
    for (uint32_t m = 0x80000000u; m; m >>= 1) {
        if (a & m) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ ZZ_CRC_POLY : b >> 1;
    }
    return p;
}
// x^(8*nbytes) mod P
__host__ __device__ inline uint32_t gf2_xpow8(uint64_t nbytes)
{
    uint32_t r = 0x80000000u;   // x^0
    uint32_t sq = 0x00800000u;  // x^8
    for (; nbytes; nbytes >>= 1) {
        if (nbytes & 1) r = gf2_mulmod(r, sq);
        sq = gf2_mulmod(sq, sq);
    }
    return r;
}
// crc(A||B) from the finished CRC-32s of A and B and |B|
__host__ __device__ inline uint32_t crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2)
{
    return gf2_mulmod(crc1, gf2_xpow8(len2)) ^ crc2;
}
// adler.cpp:5-15: `second` computed with start value 0
__host__ __device__ inline uint32_t adler_combine(uint32_t first, uint32_t second, uint64_t len2)
{
    uint64_t a = (uint64_t)(first & 0xFFFF) + (second & 0xFFFF);
    uint64_t b = (uint64_t)(first >> 16) + (second >> 16) + (len2 % ZZ_ADLER_MOD) * (first & 0xFFFF);
    return (uint32_t)(((b % ZZ_ADLER_MOD) << 16) | (a % ZZ_ADLER_MOD));
}

// ---- Adler-32 partial of one packet, one wavefront ----------------------------------------------------
// Returns (a, b) for start value 0: a = sum d_i, b = sum (len - i) d_i, both mod 65521. 16 bytes per lane
// per step, coalesced; len <= 32768 so 64-bit accumulators cannot overflow.
#ifndef ZZ_ADLER_INFLIGHT
#define ZZ_ADLER_INFLIGHT 4      // 16-byte loads a lane has in flight (the packet kernels sum a packet on ONE wavefront in front of its first block)
#endif
__device__ __forceinline__ zz_cks wave_adler(const uint8_t* p, uint32_t len)
{
    const int lane = lane_id();
    uint32_t A = 0;
    uint64_t C = 0;  // sum i * d_i
    const uint32_t nchunks = len >> 4;
#if ZZ_ADLER_INFLIGHT > 1
    // U loads in flight per lane (one at a time left the wavefront waiting out 32 memory latencies per 32 KiB packet), and the sums by
    // v_dot4_u32_u8: a chunk's byte sum is four dot products with ones, its sum k b_k (k = 0..15) four with the weights 0..15
    for (uint32_t c0 = 0; c0 < nchunks; c0 += ZZ_WAVE * ZZ_ADLER_INFLIGHT) {
        uint4 v[ZZ_ADLER_INFLIGHT];
#pragma unroll
        for (int u = 0; u < ZZ_ADLER_INFLIGHT; ++u) {
            const uint32_t c = c0 + (uint32_t)u * ZZ_WAVE + (uint32_t)lane;
            v[u] = make_uint4(0, 0, 0, 0);
            if (c < nchunks) __builtin_memcpy(&v[u], p + ((uint64_t)c << 4), 16);
        }
#pragma unroll
        for (int u = 0; u < ZZ_ADLER_INFLIGHT; ++u) {
            const uint32_t c = c0 + (uint32_t)u * ZZ_WAVE + (uint32_t)lane;      // (a chunk beyond the end is all zeros: adds nothing)
            const uint32_t s = __builtin_amdgcn_udot4(v[u].x, 0x01010101u, __builtin_amdgcn_udot4(v[u].y, 0x01010101u,
                               __builtin_amdgcn_udot4(v[u].z, 0x01010101u, __builtin_amdgcn_udot4(v[u].w, 0x01010101u, 0u, false), false), false), false);
            const uint32_t t = __builtin_amdgcn_udot4(v[u].x, 0x03020100u, __builtin_amdgcn_udot4(v[u].y, 0x07060504u,
                               __builtin_amdgcn_udot4(v[u].z, 0x0B0A0908u, __builtin_amdgcn_udot4(v[u].w, 0x0F0E0D0Cu, 0u, false), false), false), false);
            A += s;
            C += (uint64_t)(c << 4) * s + t;
        }
    }
#else
    for (uint32_t c = lane; c < nchunks; c += ZZ_WAVE) {
        uint4 v;
        __builtin_memcpy(&v, p + ((uint64_t)c << 4), 16);
        uint32_t s0 = __builtin_amdgcn_sad_u8(v.x, 0u, 0u), s1 = __builtin_amdgcn_sad_u8(v.y, 0u, 0u);
        uint32_t s2 = __builtin_amdgcn_sad_u8(v.z, 0u, 0u), s3 = __builtin_amdgcn_sad_u8(v.w, 0u, 0u);
        // sum k*b_k inside each dword (k = 0..3)
        auto w3 = [](uint32_t x) { return ((x >> 8) & 0xFF) + 2 * ((x >> 16) & 0xFF) + 3 * (x >> 24); };
        uint32_t t = w3(v.x) + (w3(v.y) + 4 * s1) + (w3(v.z) + 8 * s2) + (w3(v.w) + 12 * s3);
        uint32_t s = s0 + s1 + s2 + s3;
        A += s;
        C += (uint64_t)(c << 4) * s + t;
    }
#endif
    uint32_t i = (nchunks << 4) + lane;
    if (i < len) {
        uint32_t d = p[i];
        A += d;
        C += (uint64_t)i * d;
    }
    uint64_t At = wave_sum64(A);
    uint64_t Ct = wave_sum64(C);
    zz_cks r;
    r.a = (uint32_t)(At % ZZ_ADLER_MOD);
    r.b = (uint32_t)(((uint64_t)len * At - Ct) % ZZ_ADLER_MOD);
    return r;
}

// One wavefront's share of a packet's sums when `nparts` wavefronts split it: the 1 KiB rows r * nparts + part (64 chunks of 16 bytes, one
// per lane), U loads in flight; the bytes behind the last whole chunk go to the last part. Raw sums: A = sum d, C = sum i d_i.
template <int U>
__device__ __forceinline__ void wave_adler_part(const uint8_t* p, uint32_t len, uint32_t part, uint32_t nparts, uint32_t& A, uint64_t& C)
{
    const int lane = lane_id();
    const uint32_t nchunks = len >> 4;
    for (uint32_t r0 = 0; (r0 * nparts + part) * ZZ_WAVE < nchunks; r0 += U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t c = ((r0 + (uint32_t)u) * nparts + part) * ZZ_WAVE + (uint32_t)lane;
            v[u] = make_uint4(0, 0, 0, 0);
            if (c < nchunks) __builtin_memcpy(&v[u], p + ((uint64_t)c << 4), 16);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t c = ((r0 + (uint32_t)u) * nparts + part) * ZZ_WAVE + (uint32_t)lane;
            const uint32_t s = __builtin_amdgcn_udot4(v[u].x, 0x01010101u, __builtin_amdgcn_udot4(v[u].y, 0x01010101u,
                               __builtin_amdgcn_udot4(v[u].z, 0x01010101u, __builtin_amdgcn_udot4(v[u].w, 0x01010101u, 0u, false), false), false), false);
            const uint32_t t = __builtin_amdgcn_udot4(v[u].x, 0x03020100u, __builtin_amdgcn_udot4(v[u].y, 0x07060504u,
                               __builtin_amdgcn_udot4(v[u].z, 0x0B0A0908u, __builtin_amdgcn_udot4(v[u].w, 0x0F0E0D0Cu, 0u, false), false), false), false);
            A += s;
            C += (uint64_t)(c << 4) * s + t;
        }
    }
    if (part == nparts - 1) {
        const uint32_t i = (nchunks << 4) + (uint32_t)lane;
        if (i < len) { const uint32_t d = p[i]; A += d; C += (uint64_t)i * d; }
    }
}

// ---- Adler-32 per chunk as its own kernel (only the sequential-stream mode needs it: the packet kernels fuse it)
__global__ __launch_bounds__(ZZ_WAVE) void k_adler_packets(zz_packet_params P)
{
    for (uint32_t k = blockIdx.x; k < P.npk; k += gridDim.x) {
        const uint64_t off = (uint64_t)k * P.packet_size;
        const uint32_t len = (uint32_t)((P.n - off) < P.packet_size ? (P.n - off) : P.packet_size);
        zz_cks c = wave_adler(P.src + off, len);
        if (lane_id() == 0) P.cks[k] = c;
    }
}

// ---- CRC-32 per packet (sibling kernel; gzip container only) ------------------------------------------
// 256 threads per packet, four wavefronts with a quarter of the packet each. Inside a quarter lane j takes the
// 32-bit words j, j+64, j+128, ... so that every load is one coalesced 256-byte row. A lane's words sit 256 bytes
// apart, so its Horner step is "times x^2048" instead of the usual "times x^32": the same four table lookups per
// word as slicing-by-4 (crc.cpp:5-33 is the byte-wise form), with tables scaled by x^(8*252). The last word and
// everything behind it (rest of the row, the later quarters) go into one GF(2) multiplication per lane; the
// packet CRC is the XOR over lanes. CRC's initial value is folded into the packet's first word.
// Packets that are not full (the last one) or not a multiple of 1024 bytes take the plain slicing-by-4 path.
#define ZZ_CRC_THREADS 256
__global__ __launch_bounds__(ZZ_CRC_THREADS) void k_crc32_packets(zz_packet_params P)
{
    __shared__ uint32_t tab[4][256];      // slicing-by-4: (byte at register position k) * x^32
    __shared__ uint32_t tabB[4][256];     // the same * x^(8*252): one step of 256 bytes
    __shared__ uint32_t red[ZZ_CRC_THREADS / ZZ_WAVE];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    {
        uint32_t c = tid;
        for (int j = 0; j < 8; ++j) c = (c >> 1) ^ ((c & 1u) * ZZ_CRC_POLY);   // crc.cpp:5-20
        tab[0][tid] = c;
    }
    __syncthreads();
    for (int t = 1; t < 4; ++t) {
        uint32_t c = tab[t - 1][tid];
        tab[t][tid] = (c >> 8) ^ tab[0][c & 0xFF];
    }
    __syncthreads();
    {
        const uint32_t x252 = gf2_xpow8(252);
        for (int t = 0; t < 4; ++t) tabB[t][tid] = gf2_mulmod(tab[t][tid], x252);
    }
    const bool fast = P.packet_size % 1024 == 0 && P.packet_size >= 1024;
    const uint32_t quarter = P.packet_size / 4, rows = quarter / 256;
    // fast path: x^(8 * bytes from my last word (inclusive of its own x^32) to the end of the packet)
    const uint32_t ktail = gf2_xpow8(256 - 4 * lane + quarter * (3 - wv));
    // slow path: x^(8 * bytes after my slice) for a full packet
    uint32_t shift_full;
    {
        const uint32_t len = P.packet_size;
        const uint32_t slice = ((len + ZZ_CRC_THREADS - 1) / ZZ_CRC_THREADS + 3) & ~3u;
        uint32_t b1 = tid * slice + slice;
        if (b1 > len) b1 = len;
        shift_full = gf2_xpow8(len - b1);
    }
    __syncthreads();
    for (uint32_t k = blockIdx.x; k < P.npk; k += gridDim.x) {
        const uint64_t off = (uint64_t)k * P.packet_size;
        const uint32_t len = (uint32_t)((P.n - off) < P.packet_size ? (P.n - off) : P.packet_size);
        const uint8_t* p = P.src + off;
        uint32_t c = 0;
        if (fast && len == P.packet_size) {
            const uint8_t* q = p + (uint64_t)wv * quarter + 4 * lane;
            uint32_t s = (wv == 0 && lane == 0) ? ~0u : 0u;        // the initial value meets the first word
            uint32_t r = 0;
            for (; r + 8 < rows; r += 8) {
                uint32_t w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) w[u] = load32(q + (uint64_t)(r + u) * 256);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t v = s ^ w[u];
                    s = tabB[3][v & 0xFF] ^ tabB[2][(v >> 8) & 0xFF] ^ tabB[1][(v >> 16) & 0xFF] ^ tabB[0][v >> 24];
                }
            }
            for (; r + 1 < rows; ++r) {
                const uint32_t v = s ^ load32(q + (uint64_t)r * 256);
                s = tabB[3][v & 0xFF] ^ tabB[2][(v >> 8) & 0xFF] ^ tabB[1][(v >> 16) & 0xFF] ^ tabB[0][v >> 24];
            }
            c = gf2_mulmod(s ^ load32(q + (uint64_t)(rows - 1) * 256), ktail);
        } else {
            const uint32_t slice = ((len + ZZ_CRC_THREADS - 1) / ZZ_CRC_THREADS + 3) & ~3u;
            uint32_t b0 = tid * slice, b1 = b0 + slice;
            if (b0 > len) b0 = len;
            if (b1 > len) b1 = len;
            if (b1 > b0) {
                c = ~0u;
                uint32_t i = b0;
                for (; i + 4 <= b1; i += 4) {
                    c ^= load32(p + i);
                    c = tab[3][c & 0xFF] ^ tab[2][(c >> 8) & 0xFF] ^ tab[1][(c >> 16) & 0xFF] ^ tab[0][c >> 24];
                }
                for (; i < b1; ++i) c = (c >> 8) ^ tab[0][(c & 0xFF) ^ p[i]];   // crc.cpp:28-31
                c = ~c;
                c = gf2_mulmod(c, len == P.packet_size ? shift_full : gf2_xpow8(len - b1));
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) c ^= __shfl_xor(c, o);
        if (lane == 0) red[wv] = c;
        __syncthreads();
        if (tid == 0) {
            uint32_t r = 0;
            for (int w = 0; w < ZZ_CRC_THREADS / ZZ_WAVE; ++w) r ^= red[w];
            P.cks[k].a = (fast && len == P.packet_size) ? ~r : r;
            P.cks[k].b = 0;
        }
        __syncthreads();
    }
}

// ---- fold the per-packet partials of a shard into one (in packet order) --------------------------------
// One workgroup; thread t folds a contiguous run of packets, then thread 0 folds the 1024 runs.
struct zz_cks_total { uint32_t a, b; uint64_t len; };
#define ZZ_RED_THREADS 1024
__global__ __launch_bounds__(ZZ_RED_THREADS) void k_cks_reduce(const zz_cks* cks, uint32_t npk, uint32_t packet_size,
                                                               uint64_t n, int kind, zz_cks_total* out)
{
    __shared__ uint32_t sa[ZZ_RED_THREADS], sb[ZZ_RED_THREADS];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (npk + ZZ_RED_THREADS - 1) / ZZ_RED_THREADS;
    uint32_t k0 = t * per, k1 = k0 + per;
    if (k0 > npk) k0 = npk;
    if (k1 > npk) k1 = npk;
    if (kind == ZZ_CKS_ADLER) {
        // adler.cpp:5-15 (combine), unrolled over all packets: with start-value-0 partials (a_k, b_k),
        //   A = sum a_k,  B = sum b_k + a_k * (bytes after packet k)      (mod 65521)
        // every term stands alone, so the fold is a plain parallel sum; a_k and the reduced byte count are below
        // 2^16, the products below 2^32.
        uint64_t sumA = 0, sumB = 0;
        if (k0 < k1) {
            const uint64_t e0 = (uint64_t)(k0 + 1) * packet_size;
            uint32_t r = (uint32_t)((n - (e0 < n ? e0 : n)) % ZZ_ADLER_MOD);      // bytes after packet k0
            for (uint32_t k = k0; k < k1; ++k) {
                const uint32_t ak = cks[k].a, bk = cks[k].b;
                sumA += ak;
                sumB += bk + (uint64_t)ak * r;
                if (k + 1 < k1) {
                    const uint64_t off = (uint64_t)(k + 1) * packet_size;
                    const uint32_t l = (uint32_t)((n - off) < packet_size ? (n - off) : packet_size);
                    r = (r + ZZ_ADLER_MOD - l % ZZ_ADLER_MOD) % ZZ_ADLER_MOD;
                }
            }
        }
        sa[t] = (uint32_t)(sumA % ZZ_ADLER_MOD); sb[t] = (uint32_t)(sumB % ZZ_ADLER_MOD);
        __syncthreads();
        for (uint32_t d = ZZ_RED_THREADS / 2; d >= 1; d >>= 1) {       // 1024 values below 2^16: no overflow
            if (t < d) { sa[t] += sa[t + d]; sb[t] += sb[t + d]; }
            __syncthreads();
        }
        if (t == 0) { out->a = sa[0] % ZZ_ADLER_MOD; out->b = sb[0] % ZZ_ADLER_MOD; out->len = n; }
        return;
    }
    uint32_t a = 0;
    uint64_t len = 0;
    const uint32_t xp = kind == ZZ_CKS_CRC ? gf2_xpow8(packet_size) : 0;
    for (uint32_t k = k0; k < k1; ++k) {
        uint64_t off = (uint64_t)k * packet_size;
        uint64_t l = (n - off) < packet_size ? (n - off) : packet_size;
        uint32_t sh = l == packet_size ? xp : gf2_xpow8(l);
        a = gf2_mulmod(a, sh) ^ cks[k].a;
        len += l;
    }
    if (kind == ZZ_CKS_CRC) {
        // CRC folds linearly: total = XOR over runs of crc_run * x^(8 * bytes after the run), all runs in parallel
        const uint64_t after = n - ((uint64_t)k1 * packet_size < n ? (uint64_t)k1 * packet_size : n);
        a = len ? gf2_mulmod(a, gf2_xpow8(after)) : 0;
    }
    sa[t] = a;
    __syncthreads();
    for (uint32_t d = ZZ_RED_THREADS / 2; d >= 1; d >>= 1) {
        if (t < d) sa[t] ^= sa[t + d];
        __syncthreads();
    }
    if (t == 0) { out->a = kind == ZZ_CKS_CRC ? sa[0] : 0; out->b = 0; out->len = n; }
}

}  // namespace zz
