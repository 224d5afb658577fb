/* zzflate_amd.h -- C ABI of the MI355X-native zzflate encoder path (libzzflate_amd.so).
 *
 * Drop-in boundary for jandevaan/zzflate's encoder entry points (reference zzflate/zzflate.h:8-19):
 *
 *   reference (C++ linkage)                                   this library
 *   ---------------------------------------------------------------------------------------------------
 *   enum Format {Zlib, Gzip, Deflate}          zzflate.h:8     ZZ_ZLIB / ZZ_GZIP / ZZ_DEFLATE (same values)
 *   struct Config {format; level; threaded}    zzflate.h:10-15 zz_config (same 8-byte layout)
 *   ZzFlateEncode(dest,&len,src,n,cfg)         zzflate.h:17    zz_encode()            [+ the C++ symbol itself,
 *   ZzFlateEncodeToCallback(src,n,cfg,fn)      zzflate.h:19    zz_encode_callback()    see include/zzflate.h]
 *   adler32x / combine                         adler.cpp:5-43  zz_adler32 / zz_adler32_combine
 *   crc32                                      crc.h:7         zz_crc32 (+ zz_crc32_combine, new)
 *
 * Plain pointers and sizes only. The *_device entry points take HIP device pointers (what
 * torch.Tensor.data_ptr() returns on ROCm) and a hipStream_t passed as void*.
 *
 * Semantics: `threaded != 0` selects packet mode = the reference's threaded path (zzflate.cpp:97-155) with
 * fixed-size ranges ("packets", default 32 KiB) instead of hardware_concurrency() ranges; every packet is
 * bit-identical to the reference's packet recipe (zzflate.cpp:101-125) wherever that recipe yields a valid
 * DEFLATE encoding, and always valid otherwise. `threaded == 0` asks for the reference's single Encoder over the
 * whole input (zzflate.cpp:84-95): produced on the device too and bit-identical to the reference wherever the
 * reference's stream is a valid encoding of the input (its multi-block level-1 stream is not when a short match crosses a
 * block cut -- defect D12 in DESIGN.md: lengths stop at the block end here), including what
 * depends on where the output goes -- at level 1 the block lengths follow from the room in the caller's buffer
 * (encoder.cpp:331-337; zztest/Test.cpp passes dest = input size) or, through the callback, from the library's
 * 1,000,000-byte chunks (outputbitstream.h:171-201), and the callback receives exactly the reference's chunks. At
 * levels 1..3 that stream is one dependency chain, so it is made by a single wavefront -- a compatibility mode,
 * not a throughput mode. There is no CPU path.
 *
 * Error convention (zzflate.cpp:229-234): *dest_len = ~0 for a bad level or a destination that cannot hold
 * the container header. This library additionally detects a destination that is too small for the stream
 * (the reference silently truncates) and reports it the same way. Functions returning int return 0 on
 * success and a negative ZZ_E_* code otherwise; zz_last_error() gives the message for this thread.
 */
#ifndef ZZFLATE_AMD_H
#define ZZFLATE_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ZZ_ZLIB = 0, ZZ_GZIP = 1, ZZ_DEFLATE = 2 };       /* zzflate.h:8 */
typedef struct { int32_t format; uint8_t level; uint8_t threaded; } zz_config; /* zzflate.h:10-15, sizeof 8 */

enum {
    ZZ_OK = 0,
    ZZ_E_LEVEL = -1,        /* level not in 0..3 (zzflate.cpp:201,230) */
    ZZ_E_NOSPACE = -2,      /* destination too small */
    ZZ_E_HIP = -3,          /* HIP runtime error / no device */
    ZZ_E_ARG = -4,          /* bad argument (packet size, null pointer) */
    ZZ_E_UNSUPPORTED = -5   /* the device lacks a property the requested mode needs (zz_ctx_set_warm_window, zz_ctx_set_extended_levels) */
};

#define ZZ_DEFAULT_PACKET 32768u
#define ZZ_MAX_PACKET_SIZE 32768u

typedef struct zz_ctx zz_ctx;

/* ---- contexts (device, workspace, timing) ------------------------------------------------------- */
int zz_ctx_create(int device, zz_ctx** out);
void zz_ctx_destroy(zz_ctx* ctx);
/* bytes of device workspace currently held */
uint64_t zz_ctx_workspace_bytes(const zz_ctx* ctx);
/* record HIP events around the encode kernel of each call; read with zz_ctx_last_kernel_ms */
void zz_ctx_enable_timing(zz_ctx* ctx, int on);
/* duration of the last call's dominant (encode) kernel in ms, from HIP events on the launch stream;
 * negative if timing was off */
double zz_ctx_last_kernel_ms(zz_ctx* ctx);

/* Warm window (beyond the reference). The reference's threaded mode gives every range a cold hash
 * table and so loses the matches that would reach back into the previous range; its single Encoder carries the table
 * across blocks (FixHashTable, encoder.cpp:320-327). With a warm window of `bytes` (0..32768) every packet (levels >= 1)
 * starts with the last `bytes` bytes in front of it hashed into its table (every position, per hash the highest), so
 * matches may cross packet boundaries while packets still encode independently. The stream is valid DEFLATE but no
 * longer the reference's threaded stream, hence a separate switch: 0 = off (default). Shards must make the window
 * available as their halo. Env ZZFLATE_WARM_WINDOW sets it for the host entry points. */
int zz_ctx_set_warm_window(zz_ctx* ctx, uint32_t bytes);
/* Levels beyond the reference (BASELINE configs[3] asks for a "level 6 (longer hash chains)" the reference does
 * not have: it rejects every level above 3, zzflate.cpp:201,230-234). Off by default, so that the drop-in keeps that error. When
 * on, levels 4, 5, 6 are accepted by the packet-mode entry points: hash chains of depth 2 / 4 / 8 (four-byte keys) over a window
 * of 8 / 32 / 32 KiB in front of every packet, one-step lazy matching, one dynamic block per packet with code lengths by
 * package-merge. Not comparable with any reference stream; defined by the oracle, checked bit for bit against it,
 * by inflate, and by ratio (mixed corpus: 0.461 at level 3, 0.422 at level 6). Shards must make the window available as their
 * halo. Workspace: 4 bytes per input byte of a call, for at most 1 GiB of input at a time (larger calls go through in batches),
 * and 128 MiB. Env ZZFLATE_EXTENDED_LEVELS=1 switches them on for the host entry points. */
int zz_ctx_set_extended_levels(zz_ctx* ctx, int on);

/* ---- sizes -------------------------------------------------------------------------------------- */
/* worst-case output bytes for n input bytes (container included) */
uint64_t zz_bound(uint64_t n, int format, int level, uint32_t packet_size);

/* ---- host-buffer entry points (drop-in for zzflate.h:17,19) -------------------------------------- */
/* dest_len: in = capacity, out = bytes written or ~0. Packet mode (threaded != 0) fans the input out over every
 * visible GPU (env ZZFLATE_DEVICES = "0,1,..." or "all"; env ZZFLATE_DEVICE = one index), the device analogue of the
 * reference's std::async fan-out over all cores (zzflate.cpp:97-155): buffers longer than one slab (64 MiB, env
 * ZZFLATE_SLAB_MIB) are cut into slabs of whole packets, slab i goes to device i mod D, and on every device H2D,
 * encode and D2H of different slabs overlap. Only two slabs of input and output are resident per device, so the input
 * may be larger than HBM. Re-entrant: concurrent calls (and calls from inside a callback) borrow separate contexts. */
int zz_encode(uint8_t* dest, uint64_t* dest_len, const uint8_t* src, uint64_t n, const zz_config* cfg);
/* callback(user, chunk, bytes) is invoked in order: header, stream chunks of <= 1,000,000 bytes
 * (outputbitstream.h:183), trailer. With threaded == 0 the chunks are the reference's own (see
 * zz_encode_stream_chunks_device). The callback's return value is ignored, as in the reference. Nothing is delivered
 * before the first part of the stream has encoded successfully. */
typedef int (*zz_callback)(void* user, const uint8_t* chunk, uint64_t bytes);
int zz_encode_callback(const uint8_t* src, uint64_t n, const zz_config* cfg, zz_callback cb, void* user);
/* packet size used by the host entry points (env ZZFLATE_PACKET_SIZE overrides the default) */
int zz_set_packet_size(uint32_t packet_size);
uint32_t zz_get_packet_size(void);

/* ---- device-resident entry points ---------------------------------------------------------------- */
/* Whole stream: d_src[0,n) -> d_dst (container header, packets, trailer). *out_len = bytes or ~0. */
int zz_encode_device(zz_ctx* ctx, const void* d_src, uint64_t n, void* d_dst, uint64_t cap, uint64_t* out_len,
                     int format, int level, uint32_t packet_size, void* hip_stream);

/* The same call in two halves: zz_encode_device_async enqueues the whole pipeline on `hip_stream` and returns without
 * waiting; zz_encode_finish waits for it and returns the length (or the error, as zz_encode_device). With two contexts
 * on two streams, call i+1 can be enqueued before call i is finished: its encode kernel fills the CUs call i's last
 * packets leave idle and runs under call i's compaction and result copy. One enqueued call per context at a time. */
int zz_encode_device_async(zz_ctx* ctx, const void* d_src, uint64_t n, void* d_dst, uint64_t cap, int format, int level,
                           uint32_t packet_size, void* hip_stream);
int zz_encode_finish(zz_ctx* ctx, uint64_t* out_len);

/* The reference's sequential whole-buffer stream (threaded == 0, zzflate.cpp:84-95) for device-resident data, in
 * the form ZzFlateEncode gives it (caller-owned buffer of `cap` bytes): level 0 (stored blocks of 65535 bytes,
 * parallel), level 1 (fixed-Huffman blocks; ONE for the whole input when (cap - header - 1) * 8 / 9 - 8 >= n, else as
 * many as encoder.cpp:331-337 cuts for that capacity) and levels 2,3 (dynamic blocks cut at 20,000 records / 500,000
 * bytes, hash table carried across blocks), the latter two produced by a single wavefront: bit-identical to the
 * reference, far slower than packet mode. n < 2 GiB (the reference funnels lengths through int). Where the reference
 * would run out of room and silently leave a truncated stream, ZZ_E_NOSPACE is returned. */
int zz_encode_stream_device(zz_ctx* ctx, const void* d_src, uint64_t n, void* d_dst, uint64_t cap, uint64_t* out_len,
                            int format, int level, void* hip_stream);
/* The reference's OWN threaded == true split (zzflate.cpp:67-78 divideInRanges, :97-155 WriteDeflateStream) for device-resident
 * data: `count` ranges of ceil(n / count) bytes -- std::thread::hardware_concurrency() of them in the reference, the caller's
 * number here --, every range a fresh encoder whose non-final output ends with one stored byte, joined in order; n < 100 * count
 * gives the single encoder's stream (:84). Bit-identical to what ZzFlateEncode(threaded = true) writes on a machine with `count`
 * hardware threads at levels 0, 2, 3 (tests/golden/ranges.json holds such streams); level 1 returns ZZ_E_LEVEL: the reference's
 * threaded level-1 stream does not inflate, use packet mode. One wavefront per range: a compatibility
 * mode (packet mode -- zz_encode_device -- is the throughput mode and cuts <= 32 KiB ranges instead).
 * Host entry points take this path when ZZFLATE_RANGES=<count> is set (threaded != 0, levels 0, 2, 3). n < 2 GiB. */
int zz_encode_ranges_device(zz_ctx* ctx, const void* d_src, uint64_t n, void* d_dst, uint64_t cap, uint64_t* out_len,
                            int format, int level, uint32_t count, void* hip_stream);
/* The same stream in the form ZzFlateEncodeToCallback gives it (zzflate.cpp:197-222): the encoder writes into
 * library-owned chunks of 1,000,000 bytes and opens a new one when the current one is not "enough" for the next block
 * (outputbitstream.h:171-201), which at level 1 also decides the block lengths. d_dst receives header + stream +
 * trailer contiguously; chunk_sizes[0..*nchunks) (up to max_chunks are written) are the byte counts of the chunks in
 * order, i.e. the sizes the reference's callback sees between the header call and the trailer call. */
int zz_encode_stream_chunks_device(zz_ctx* ctx, const void* d_src, uint64_t n, void* d_dst, uint64_t cap, uint64_t* out_len,
                                   int format, int level, uint64_t* chunk_sizes, uint32_t max_chunks, uint32_t* nchunks,
                                   void* hip_stream);

/* One shard of a stream (multi-GPU: ranks own contiguous packet ranges). d_src points at the shard's first
 * byte; `halo` bytes in front of it are readable input of the same stream (level >= 2 backward match
 * extension reads up to 258 of them; pass 0 for the first shard). No header/trailer is written; the shard's
 * checksum partial comes back for zz_adler32_combine / zz_crc32_combine:
 *   checksum == ZZ_ZLIB: cks = Adler-32 of the shard with start value 0 ((b<<16)|a)
 *   checksum == ZZ_GZIP: cks = CRC-32 of the shard;   ZZ_DEFLATE: none. */
int zz_encode_shard_device(zz_ctx* ctx, const void* d_src, uint64_t n, uint64_t halo, int is_last_shard,
                           void* d_dst, uint64_t cap, uint64_t* out_len, uint32_t* cks, int checksum,
                           int level, uint32_t packet_size, void* hip_stream);

/* The shard call in two halves (as zz_encode_device_async / zz_encode_finish), for ranks that enqueue the next step's shard
 * before the previous one's size and checksum have been exchanged. */
int zz_encode_shard_device_async(zz_ctx* ctx, const void* d_src, uint64_t n, uint64_t halo, int is_last_shard, void* d_dst,
                                 uint64_t cap, int checksum, int level, uint32_t packet_size, void* hip_stream);
int zz_encode_shard_finish(zz_ctx* ctx, uint64_t* out_len, uint32_t* cks, int checksum);

/* Fan-out and join over several GPUs of ONE process, for data already resident on them -- WriteDeflateStream's
 * std::async fan-out and in-order memmove join (zzflate.cpp:97-155) with devices for threads and xGMI peer copies for
 * memmove; no torch, no RCCL. Shard i = d_src[i][0, n[i]) lives on the device of ctxs[i] (one context per shard; a device
 * may appear more than once); the shards are consecutive ranges of one stream, every one but the last a whole number of
 * packets; halo[i] (halo may be NULL = all 0) = readable bytes of the stream in front of d_src[i] on that device, as
 * zz_encode_shard_device wants them. All shards are encoded concurrently; each is pulled to its final offset in d_dst --
 * on the device of ctxs[0], where shard 0 is encoded in place -- with hipMemcpyPeerAsync as soon as the shards in front of
 * it have finished; checksums are folded on the host; header and trailer are written. *out_len = bytes or ~0. */
int zz_encode_multi_device(zz_ctx* const* ctxs, int nshards, const void* const* d_src, const uint64_t* n, const uint64_t* halo,
                           void* d_dst, uint64_t cap, uint64_t* out_len, int format, int level, uint32_t packet_size);

/* Self-verification: inflates, on the device, every packet of the stream the LAST zz_encode_device /
 * zz_encode_shard_device call on this context produced, and compares with that call's input (both buffers must still
 * be in place). *bad_packets = packets that do not decode to their input, *first_bad_packet = the lowest such packet
 * (~0 if none). A checker for tests, benchmarks and deployments that want an end-to-end guarantee; the reference has
 * no decoder (decoder.h is an empty stub). */
int zz_verify_last_device(zz_ctx* ctx, uint64_t* bad_packets, uint64_t* first_bad_packet, void* hip_stream);

/* Random access: where packet k (input bytes [k*packet_size, (k+1)*packet_size)) of the stream produced by the LAST
 * zz_encode_device / zz_encode_shard_device call on this context lies. *offset counts from the first byte of the
 * DEFLATE stream (behind the container header), *bytes is the packet's length. Packets are independent byte-aligned
 * runs of complete blocks (zzflate.cpp:101-125), so each can be inflated on its own. */
int zz_packet_extent_device(zz_ctx* ctx, uint64_t packet, uint64_t* offset, uint64_t* bytes, void* hip_stream);

/* container pieces for assembling shards on the host */
int zz_header(int format, uint8_t out[10]);                                   /* returns 0/2/10 */
int zz_trailer(int format, uint32_t cks_total, uint64_t n, uint8_t out[8]);   /* returns 0/4/8  */

/* ---- checksums (host utilities, adler.cpp / crc.cpp semantics) ------------------------------------- */
uint32_t zz_adler32(uint32_t start, const uint8_t* p, uint64_t n);
uint32_t zz_adler32_combine(uint32_t first, uint32_t second_start0, uint64_t len_second);
uint32_t zz_crc32(const uint8_t* p, uint64_t n, uint32_t start);
uint32_t zz_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2);

/* ---- synthetic inputs (BASELINE.json configs), generated on the device ------------------------------ */
enum { ZZ_GEN_TEXT = 0, ZZ_GEN_RANDOM = 1, ZZ_GEN_LOG = 2, ZZ_GEN_MIX = 3 };
/* fills d_buf[0,n); byte i is a pure function of (kind, seed, first_byte + i), in 64 KiB blocks (first_byte must
 * be a multiple of 65536) */
int zz_generate_device(zz_ctx* ctx, int kind, uint64_t seed, uint64_t first_byte, void* d_buf, uint64_t n,
                       void* hip_stream);
/* the same bytes computed on the host (for parity sampling) */
int zz_generate_host(int kind, uint64_t seed, uint64_t first_byte, uint8_t* buf, uint64_t n);

const char* zz_last_error(void);
const char* zz_version(void);
/* the compile-time experiment switches this binary carries, space-separated; "" for the product build (some of them write
 * deliberately wrong streams for timing runs: a deployment can assert on "") */
const char* zz_build_flags(void);

#ifdef __cplusplus
}
#endif
#endif
