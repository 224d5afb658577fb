// zz_compact.h -- in-order join of packet outputs: prefix sum of packet sizes, byte compaction, and the
// container header/trailer. Replaces the serial memmove join of zzflate.cpp:134-155 and the header /
// trailer writers (zzflate.cpp:28-63, :170-192).
#pragma once
#include "zz_checksum.h"

namespace zz {

struct zz_result {
    uint64_t stream_bytes;   // compacted DEFLATE bytes (no header/trailer)
    uint64_t total_bytes;    // header + stream + trailer (0 for shard calls)
    uint32_t err;            // 0 ok; bit0 slot overflow, bit1 destination too small
    uint32_t cks_a, cks_b;   // shard checksum partial (start value 0 semantics)
    uint64_t cks_len;
};

#define ZZ_SCAN_THREADS 1024
// exclusive prefix sum of sizes[0..npk) into offsets; total to res->stream_bytes. One workgroup, 4096 packets per
// round: thread t takes four consecutive sizes (one 16-byte load), wave scan + scan of the 16 wave totals, running
// carry between rounds.
__global__ __launch_bounds__(ZZ_SCAN_THREADS) void k_scan_sizes(const uint32_t* sizes, uint32_t npk,
                                                                uint64_t* offsets, zz_result* res)
{
    __shared__ uint32_t wtot[ZZ_SCAN_THREADS / ZZ_WAVE];
    const uint32_t t = threadIdx.x, lane = t & 63, wv = t >> 6;
    uint64_t carry = 0;
    for (uint32_t r0 = 0; r0 < npk; r0 += 4 * ZZ_SCAN_THREADS) {
        const uint32_t k = r0 + 4 * t;
        uint32_t v[4] = { 0, 0, 0, 0 };
        if (k + 4 <= npk) { const uint4 q = *(const uint4*)(sizes + k); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
        else for (uint32_t i = 0; i < 4; ++i) if (k + i < npk) v[i] = sizes[k + i];
        const uint32_t mine = v[0] + v[1] + v[2] + v[3];          // < 2^18 each: a round stays far below 2^32
        const uint32_t incl = wave_scan_incl(mine);
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        uint32_t wbase = 0, total = 0;
        for (uint32_t i = 0; i < ZZ_SCAN_THREADS / ZZ_WAVE; ++i) { const uint32_t x = wtot[i]; if (i < wv) wbase += x; total += x; }
        uint64_t o = carry + wbase + (incl - mine);
        for (uint32_t i = 0; i < 4; ++i) { if (k + i < npk) offsets[k + i] = o; o += v[i]; }
        carry += total;
        __syncthreads();
    }
    if (t == 0) res->stream_bytes = carry;
}

// copy every packet's bytes from its slot to dst + offsets[k]; skipped entirely if the destination is too
// small (err bit1; the reference would silently truncate)
__global__ __launch_bounds__(256) void k_compact(const uint8_t* slots, uint32_t slot_stride, const uint32_t* sizes,
                                                 const uint64_t* offsets, uint32_t npk, uint8_t* dst,
                                                 uint64_t dst_cap, zz_result* res)
{
    if (res->stream_bytes > dst_cap) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&res->err, 2u);
        return;
    }
    for (uint32_t k = blockIdx.x; k < npk; k += gridDim.x)
        coop_copy(dst + offsets[k], slots + (uint64_t)k * slot_stride, sizes[k], threadIdx.x, blockDim.x);
}

// one big copy (the sequential-stream mode has a single 'packet'): 64 KiB per workgroup trip
__global__ __launch_bounds__(256) void k_copy_stream(const uint8_t* src, const uint32_t* size, uint8_t* dst, uint64_t dst_cap,
                                                     zz_result* res)
{
    const uint64_t n = *size;
    if (n > dst_cap) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&res->err, 2u);
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) res->stream_bytes = n;
    for (uint64_t off = (uint64_t)blockIdx.x << 16; off < n; off += (uint64_t)gridDim.x << 16)
        coop_copy(dst + off, src + off, n - off < 65536 ? n - off : 65536, threadIdx.x, blockDim.x);
}

// header (zzflate.cpp:28-48) + trailer (zzflate.cpp:170-192) around a finished stream; one thread.
//   zlib : 78 01 ... adler32x(1, src, n) big-endian
//   gzip : 1f 8b 08 00 00000000 00 ff ... crc32 LE, (uint32)n LE
__global__ void k_finalize(uint8_t* dst, uint64_t cap, int format, const zz_cks_total* cks, uint64_t n, zz_result* res)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t hl = format == ZZ_FMT_ZLIB ? 2 : format == ZZ_FMT_GZIP ? 10 : 0;
    const uint32_t tl = format == ZZ_FMT_ZLIB ? 4 : format == ZZ_FMT_GZIP ? 8 : 0;
    const uint64_t total = hl + res->stream_bytes + tl;
    if (cks) { res->cks_a = cks->a; res->cks_b = cks->b; res->cks_len = cks->len; }
    if (res->err || total > cap) { res->err |= 2u; res->total_bytes = 0; return; }
    if (format == ZZ_FMT_ZLIB) {
        dst[0] = 0x78; dst[1] = 0x01;
        uint32_t part = ((uint32_t)cks->b << 16) | cks->a;
        uint32_t ad = adler_combine(1u, part, n);
        uint8_t* t = dst + hl + res->stream_bytes;
        t[0] = (uint8_t)(ad >> 24); t[1] = (uint8_t)(ad >> 16); t[2] = (uint8_t)(ad >> 8); t[3] = (uint8_t)ad;
    } else if (format == ZZ_FMT_GZIP) {
        const uint8_t gz[10] = { 0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xFF };
        for (int i = 0; i < 10; ++i) dst[i] = gz[i];
        uint8_t* t = dst + hl + res->stream_bytes;
        uint32_t c = cks->a, l = (uint32_t)n;
        t[0] = (uint8_t)c; t[1] = (uint8_t)(c >> 8); t[2] = (uint8_t)(c >> 16); t[3] = (uint8_t)(c >> 24);
        t[4] = (uint8_t)l; t[5] = (uint8_t)(l >> 8); t[6] = (uint8_t)(l >> 16); t[7] = (uint8_t)(l >> 24);
    }
    res->total_bytes = total;
}

}  // namespace zz
