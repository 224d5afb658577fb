// zz_level0.h -- level 0: stored blocks (encoder.cpp:482-502), one packet per wavefront, written straight
// to their final place: a stored packet's size is known in advance, so no slots and no compaction.
//
// Packet layout (zzflate.cpp:101-125 with level 0):
//   non-final, len L > 1 : [00][L-1 as LEN,~LEN][L-1 bytes] [00][01 00][FE FF][last byte]  = L + 10 bytes
//   non-final, L == 1    : [00][01 00][FE FF][byte]                                       = 6 bytes
//   final                : [01][L,~L][L bytes]                                             = L + 5 bytes
// (packet_size <= 32768 < 65535, so a packet is never split into several stored blocks)
#pragma once
#include "zz_checksum.h"

namespace zz {

__host__ __device__ inline uint64_t l0_packet_bytes(uint32_t len, bool is_final)
{
    if (is_final) return (uint64_t)len + 5;
    return len > 1 ? (uint64_t)len + 10 : 6;
}

struct zz_l0_params {
    zz_packet_params pk;
    uint8_t* dst;          // final destination of the first packet of this shard
    int stream_mode;       // 1: the reference's sequential stream (threaded=false): stored blocks of pk.packet_size
                           //    (= 65535, encoder.cpp:484) with no alignment blocks; only the last one is final
};

__device__ __forceinline__ void put_stored_header(uint8_t* d, uint32_t final, uint32_t n)
{
    d[0] = (uint8_t)final;                 // BFINAL, BTYPE=00, padded to a byte (encoder.cpp:495-496)
    d[1] = (uint8_t)(n & 0xFF);            // LEN  (encoder.cpp:497)
    d[2] = (uint8_t)(n >> 8);
    d[3] = (uint8_t)(~n & 0xFF);           // NLEN (encoder.cpp:498)
    d[4] = (uint8_t)((~n >> 8) & 0xFF);
}

// coop_copy (zz_wave.h) with the Adler-32 partial sums of the copied bytes accumulated from the same registers:
// A += sum d_i, C += sum (base + i) * d_i for the bytes src[0,n) whose packet-relative index starts at `base`.
__device__ __forceinline__ void coop_copy_adler(uint8_t* dst, const uint8_t* src, uint64_t n, uint32_t tid,
                                                uint32_t nthreads, uint32_t base, bool want, uint32_t& A, uint64_t& C)
{
    if (n == 0) return;
    uint64_t head = (16 - ((uintptr_t)dst & 15)) & 15;
    if (head > n) head = n;
    if (tid < head) { const uint32_t d = src[tid]; dst[tid] = (uint8_t)d; A += d; C += (uint64_t)(base + tid) * d; }
    const uint64_t body = (n - head) >> 4;
    const uint8_t* s = src + head;
    uint4* d4 = (uint4*)(dst + head);
    auto sums = [&](const uint4& v, uint64_t i) {
        if (!want) return;
        const uint32_t s0 = __builtin_amdgcn_sad_u8(v.x, 0u, 0u), s1 = __builtin_amdgcn_sad_u8(v.y, 0u, 0u);
        const uint32_t s2 = __builtin_amdgcn_sad_u8(v.z, 0u, 0u), s3 = __builtin_amdgcn_sad_u8(v.w, 0u, 0u);
        auto w3 = [](uint32_t x) { return ((x >> 8) & 0xFF) + 2 * ((x >> 16) & 0xFF) + 3 * (x >> 24); };
        const uint32_t t = w3(v.x) + (w3(v.y) + 4 * s1) + (w3(v.z) + 8 * s2) + (w3(v.w) + 12 * s3);
        const uint32_t sum = s0 + s1 + s2 + s3;
        A += sum;
        C += (uint64_t)(base + (uint32_t)head + (uint32_t)(i << 4)) * sum + t;
    };
    uint64_t i = tid;
    // four independent 16-byte loads in flight per thread, then four aligned stores (as coop_copy does)
    for (; i + 3 * (uint64_t)nthreads < body; i += 4 * (uint64_t)nthreads) {
        uint4 v0, v1, v2, v3;
        __builtin_memcpy(&v0, s + (i << 4), 16);
        __builtin_memcpy(&v1, s + ((i + nthreads) << 4), 16);
        __builtin_memcpy(&v2, s + ((i + 2 * (uint64_t)nthreads) << 4), 16);
        __builtin_memcpy(&v3, s + ((i + 3 * (uint64_t)nthreads) << 4), 16);
        d4[i] = v0; d4[i + nthreads] = v1; d4[i + 2 * (uint64_t)nthreads] = v2; d4[i + 3 * (uint64_t)nthreads] = v3;
        sums(v0, i); sums(v1, i + nthreads); sums(v2, i + 2 * (uint64_t)nthreads); sums(v3, i + 3 * (uint64_t)nthreads);
    }
    for (; i < body; i += nthreads) {
        uint4 v;
        __builtin_memcpy(&v, s + (i << 4), 16);
        d4[i] = v;
        sums(v, i);
    }
    const uint64_t done = head + (body << 4);
    const uint64_t tail = n - done;
    if (tid < tail) {
        const uint32_t d = src[done + tid];
        dst[done + tid] = (uint8_t)d;
        A += d; C += (uint64_t)(base + (uint32_t)done + tid) * d;
    }
}

__global__ __launch_bounds__(256) void k_encode_l0(zz_l0_params Q)
{
    __shared__ uint64_t red_a[4], red_c[4];
    const zz_packet_params& P = Q.pk;
    const uint32_t tid = threadIdx.x;
    const bool want = P.cks_kind == ZZ_CKS_ADLER;
    for (uint32_t k = blockIdx.x; k < P.npk; k += gridDim.x) {
        const uint64_t off = (uint64_t)k * P.packet_size;
        const uint32_t len = (uint32_t)((P.n - off) < P.packet_size ? (P.n - off) : P.packet_size);
        const bool is_final = P.last_is_final && k == P.npk - 1;
        const uint8_t* src = P.src + off;
        uint8_t* d = Q.dst + (uint64_t)k * (Q.stream_mode ? (uint64_t)P.packet_size + 5 : l0_packet_bytes(P.packet_size, false));
        uint32_t A = 0;
        uint64_t C = 0;      // Adler-32 partial sums of this thread's bytes (the copy and the checksum share one read)
        if (is_final || Q.stream_mode) {
            if (tid == 0) put_stored_header(d, is_final ? 1 : 0, len);
            coop_copy_adler(d + 5, src, len, tid, blockDim.x, 0, want, A, C);
        } else {
            uint8_t* tail = d;
            if (len > 1) {
                if (tid == 0) put_stored_header(d, 0, len - 1);
                coop_copy_adler(d + 5, src, len - 1, tid, blockDim.x, 0, want, A, C);
                tail = d + 5 + (len - 1);
            }
            if (tid == 0) {
                put_stored_header(tail, 0, 1);
                const uint32_t last = src[len - 1];
                tail[5] = (uint8_t)last;
                A += last; C += (uint64_t)(len - 1) * last;
            }
        }
        if (want) {
            const uint64_t At = wave_sum64(A), Ct = wave_sum64(C);
            if ((tid & 63) == 0) { red_a[tid >> 6] = At; red_c[tid >> 6] = Ct; }
            __syncthreads();
            if (tid == 0) {
                const uint64_t Aa = red_a[0] + red_a[1] + red_a[2] + red_a[3];
                const uint64_t Cc = red_c[0] + red_c[1] + red_c[2] + red_c[3];
                zz_cks c;
                c.a = (uint32_t)(Aa % ZZ_ADLER_MOD);
                c.b = (uint32_t)(((uint64_t)len * Aa - Cc) % ZZ_ADLER_MOD);   // b = sum (len - i) d_i
                P.cks[k] = c;
            }
            __syncthreads();
        }
    }
}

}  // namespace zz
