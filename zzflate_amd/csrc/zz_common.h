// zz_common.h -- shared constants and launch-parameter structs for the MI355X DEFLATE encoder.
//
// Vocabulary follows the reference (jandevaan/zzflate): a *packet* is one independently encoded input
// range (the lambda at zzflate.cpp:101-125); its output is a byte-aligned run of complete DEFLATE blocks,
// so packet outputs concatenate by plain byte copy (zzflate.cpp:134-155). One packet = one wavefront.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ZZ_WAVE 64
#define ZZ_HASH_BITS 13                 // encoder.h:41
#define ZZ_HASH_SIZE (1 << ZZ_HASH_BITS)
#define ZZ_MAX_LEN 258                  // encoder.h:47
#define ZZ_MAX_PACKET 32768             // positions fit 15 bits -> u16 table entries (pos+1, 0 = empty)
#define ZZ_BATCH_LEN 16384              // encoder.cpp:227

enum { ZZ_FMT_ZLIB = 0, ZZ_FMT_GZIP = 1, ZZ_FMT_DEFLATE = 2 };   // zzflate.h:8
enum { ZZ_CKS_NONE = 0, ZZ_CKS_ADLER = 1, ZZ_CKS_CRC = 2 };

// Per-packet checksum partial. Adler: (a, b) of the packet computed with start value 0
// (adler.cpp:5-15 `combine` semantics). CRC: a = crc32 of the packet with start value 0, b unused.
struct zz_cks { uint32_t a, b; };

struct zz_packet_params {
    const uint8_t* src;       // first byte of this shard
    uint64_t n;               // shard length in bytes
    uint64_t halo;            // readable bytes in front of src (level >= 2 backward extension, encoder.cpp:404)
    uint32_t packet_size;     // bytes per packet (last one may be shorter)
    uint32_t npk;             // number of packets in the shard
    int last_is_final;        // last packet of the shard carries BFINAL
    int cks_kind;             // ZZ_CKS_*
    uint8_t* slots;           // npk worst-case sized output slots
    uint32_t slot_stride;     // bytes per slot (multiple of 16)
    uint32_t* sizes;          // out: bytes written per packet
    zz_cks* cks;              // out: checksum partial per packet (may be null when cks_kind == NONE)
    uint32_t* err;            // out: sticky error word (slot overflow etc.)
    unsigned long long* prof; // diagnostic builds (-DZZ_PROF) only: per-phase cycle sums; ignored otherwise
    uint32_t warm;            // level 1: bytes in front of a packet hashed into its table before the parse (0: cold, the reference)
    const uint8_t* tail;      // k_encode_l1p: 128 bytes, the shard's last min(n, 64) bytes followed by zeros (k_fill_tail)
    uint32_t dbg_viol;        // tests: report the LDS-order check as failed (error bit 4) whatever it saw
};
// bits of *err
#define ZZ_ERR_SLOT_OVERFLOW 1u
#define ZZ_ERR_LDS_ORDER 4u   // a kernel saw a slot keep a LOWER lane's store of an instruction a higher lane took part in

// ---- the sequential stream's output buffers (outputbitstream.h:171-201) ------------------------------------------
// A single Encoder asks EnsureOutputLength(length) at every block start. With a caller-owned buffer the answer is the
// room left in it; through the callback API the library owns chunks of 1,000,000 bytes and opens a new one when the
// current one is not "enough" (more than twice the length asked for, or more than 2^18 bytes). Block sizes at level 1
// follow from that answer (encoder.cpp:331-337), and the chunks are what the callback receives (zzflate.cpp:207-215),
// so the rule is shared by the stream kernels (which apply it) and the host (which replays the kernels' log of
// (bytes stored, length asked for) pairs to find the chunk boundaries).
#define ZZ_CHUNK_BYTES 1000000ll
struct zz_chunker {
    uint64_t chunk_start;     // bytes stored when the current chunk was opened
    uint32_t nchunks;         // chunks opened so far
};
// returns AvailableBytes() after the call; *opened = a new chunk starts at `stored`
__host__ __device__ inline int64_t zz_chunk_ensure(zz_chunker& k, uint64_t stored, int64_t length, bool* opened)
{
    const int64_t avail = k.nchunks ? (int64_t)(k.chunk_start + ZZ_CHUNK_BYTES) - (int64_t)stored : 0;
    *opened = false;
    if (avail > 2 * length || avail > (1 << 18)) return avail;          // IsEnough, outputbitstream.h:192-201
    k.chunk_start = stored;
    k.nchunks++;
    *opened = true;
    return ZZ_CHUNK_BYTES;
}
// what the stream kernels get besides zz_packet_params
struct zz_stream_ctl {
    uint64_t cap;             // fixed form: bytes of the caller's buffer behind the container header (encoder sees dest+hl)
    int chunked;              // 0: caller-owned buffer of `cap` bytes, 1: library-owned chunks (callback API)
    uint64_t* log;            // chunked: (bytes stored, length asked for) per EnsureOutputLength call, or null
    uint32_t log_cap;         // pairs that fit
    uint32_t* log_n;          // pairs written (may exceed log_cap: the host then reports an error)
    uint32_t* truncated;      // set when the reference would have stopped early (no room left): the stream is incomplete
};
__device__ __forceinline__ void zz_log_ensure(const zz_stream_ctl& C, uint32_t& nlog, uint64_t stored, uint64_t need)
{
    if ((threadIdx.x & 63) == 0 && C.log && nlog < C.log_cap) { C.log[2 * nlog] = stored; C.log[2 * nlog + 1] = need; }
    nlog++;
}

#ifdef ZZ_PROF
#define ZZ_PROF_DECL unsigned long long prof_acc[16] = {0}; unsigned long long prof_last; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(prof_last) :: "memory");
// stamp form of cdna_hip_programming.md section 7: one asm statement, fenced against the scheduler
#define ZZ_T(i) do { unsigned long long _t; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    prof_acc[i] += _t - prof_last; prof_last = _t; } while (0)
// diagnostic only: drain outstanding vector-memory operations so that the next stamp prices their latency
#define ZZ_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define ZZ_C(i, v) do { prof_acc[i] += (v); } while (0)
#define ZZ_PROF_FLUSH(P) do { if (threadIdx.x == 0 && (P).prof) for (int _i = 0; _i < 16; ++_i) atomicAdd(&(P).prof[_i], prof_acc[_i]); } while (0)
// one set of counters per wavefront of the workgroup (zz_debug_read_prof_sets)
#define ZZ_PROF_FLUSH_W(P, w) do { if ((threadIdx.x & 63) == 0 && (P).prof) for (int _i = 0; _i < 16; ++_i) atomicAdd(&(P).prof[16 * (w) + _i], prof_acc[_i]); } while (0)
#else
#define ZZ_PROF_DECL
#define ZZ_T(i)
#define ZZ_DRAIN()
#define ZZ_C(i, v)
#define ZZ_PROF_FLUSH(P)
#define ZZ_PROF_FLUSH_W(P, w)
#endif
