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
};

__device__ __forceinline__ void put_stored_header(uint8_t* d, uint32_t final, uint32_t n)
{
    d[0] = (uint8_t)final;                 // BFINAL, BTYPE=00, padded to a byte (encoder.cpp:495-496)
    d[1] = (uint8_t)(n & 0xFF);            // LEN  (encoder.cpp:497)
    d[2] = (uint8_t)(n >> 8);
    d[3] = (uint8_t)(~n & 0xFF);           // NLEN (encoder.cpp:498)
    d[4] = (uint8_t)((~n >> 8) & 0xFF);
}

__global__ __launch_bounds__(256) void k_encode_l0(zz_l0_params Q)
{
    const zz_packet_params& P = Q.pk;
    const uint32_t tid = threadIdx.x;
    for (uint32_t k = blockIdx.x; k < P.npk; k += gridDim.x) {
        const uint64_t off = (uint64_t)k * P.packet_size;
        const uint32_t len = (uint32_t)((P.n - off) < P.packet_size ? (P.n - off) : P.packet_size);
        const bool is_final = P.last_is_final && k == P.npk - 1;
        const uint8_t* src = P.src + off;
        uint8_t* d = Q.dst + (uint64_t)k * l0_packet_bytes(P.packet_size, false);
        if (P.cks_kind == ZZ_CKS_ADLER && tid < ZZ_WAVE) {
            zz_cks c = wave_adler(src, len);
            if (tid == 0) P.cks[k] = c;
        }
        if (is_final) {
            if (tid == 0) put_stored_header(d, 1, len);
            coop_copy(d + 5, src, len, tid, blockDim.x);
        } else {
            uint8_t* tail = d;
            if (len > 1) {
                if (tid == 0) put_stored_header(d, 0, len - 1);
                coop_copy(d + 5, src, len - 1, tid, blockDim.x);
                tail = d + 5 + (len - 1);
            }
            if (tid == 0) {
                put_stored_header(tail, 0, 1);
                tail[5] = src[len - 1];
            }
        }
    }
}

}  // namespace zz
