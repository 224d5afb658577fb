/* oracle/zzoracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C11) of the zzflate encoder path, used only as the parity checker by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg. See zzoracle.c for per-function citations
 * into the reference (file:line relative to /root/reference/zzflate/).
 *
 * Pinned against: the compiled reference (oracle/_ref, built by oracle/Makefile from the reference
 * sources) on all 11 Canterbury files x levels 0..3 x 3 containers, whole-stream and packet mode, and
 * against the reference's own six known-answer tests (tests/test_oracle_kat.py); golden hashes of the
 * reference's outputs are committed under tests/golden/.
 */
#ifndef ZZORACLE_H
#define ZZORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ZZO_ZLIB = 0, ZZO_GZIP = 1, ZZO_DEFLATE = 2 };
#define ZZO_ERROR (~(uint64_t)0)

/* Whole stream, reference threaded=false semantics through the fixed-buffer API (zzflate.cpp:225-242).
 * Returns bytes written or ZZO_ERROR (bad level / no room for the header). */
uint64_t zzo_encode(uint8_t* dest, uint64_t cap, const uint8_t* src, uint64_t n, int format, int level);

/* Whole stream through the callback API (zzflate.cpp:197-222): library-owned 1,000,000-byte chunks.
 * The concatenation of all callback payloads is written to dest; chunk_sizes (may be NULL) receives up
 * to max_chunks callback payload sizes in call order; *nchunks the number of callbacks. */
uint64_t zzo_encode_callback(uint8_t* dest, uint64_t cap, const uint8_t* src, uint64_t n, int format,
                             int level, uint64_t* chunk_sizes, int max_chunks, int* nchunks);

/* Packet mode = reference threaded=true semantics (zzflate.cpp:97-155) with fixed-size ranges of
 * packet_size bytes instead of hardware_concurrency() ranges. */
uint64_t zzo_encode_packets(uint8_t* dest, uint64_t cap, const uint8_t* src, uint64_t n, int format,
                            int level, uint64_t packet_size);

/* One packet (the lambda at zzflate.cpp:101-125) over base[off, off+len). Returns bytes written. */
uint64_t zzo_packet(int level, const uint8_t* base, uint64_t off, uint64_t len, int is_final,
                    uint8_t* out, uint64_t cap);

/* The same with a warm window (NOT in the reference; the product's zz_ctx_set_warm_window and its extended levels 4..6,
 * SURVEY.md 8f.2/8f.3): at levels >= 1 the last `warm` bytes in front of a packet are entered into its hash table before
 * it is parsed. warm = 0 is the function above. */
uint64_t zzo_encode_packets_warm(uint8_t* dest, uint64_t cap, const uint8_t* src, uint64_t n, int format,
                                 int level, uint64_t packet_size, uint64_t warm);
uint64_t zzo_packet_warm(int level, const uint8_t* base, uint64_t off, uint64_t len, int is_final,
                         uint8_t* out, uint64_t cap, uint64_t warm);

/* Extended levels 4, 5, 6 (NOT in the reference, which rejects level > 3: zzflate.cpp:201,230-234): bounded
 * hash chains (depth 2 / 4 / 8) over a window of 8 / 32 / 32 KiB in front of the packet, one-step lazy matching,
 * package-merge code lengths. Packet mode only: pass level 4..6 to zzo_encode_packets / zzo_packet (warm is ignored
 * there: the levels bring their own window). The definition is in zzoracle.c ("Extended levels"). */
void zzo_pm_lengths(const int* freqs, int n, int maxlen, int* out);             /* optimal length-limited code lengths */

/* timing driver for bench.py's cpu_baseline leg (kind "port"): see zzoracle.c */
double zzo_bench(int mode, const uint8_t* base, uint64_t nslices, uint64_t slice_bytes, uint64_t nitems, int threads,
                 int format, int level, uint32_t P, uint64_t* produced, double* thread_secs);

/* checksums */
uint32_t zzo_adler32(uint32_t start, const uint8_t* p, uint64_t n);             /* adler.cpp:17-43  */
uint32_t zzo_adler_combine(uint32_t first, uint32_t second, uint64_t len2);    /* adler.cpp:5-15   */
uint32_t zzo_crc32(const uint8_t* p, uint64_t n, uint32_t start);               /* crc.cpp:24-33    */
uint32_t zzo_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2);        /* (no reference twin) */

/* Huffman pieces, exposed for the known-answer tests */
void zzo_calc_lengths(const int* freqs, int n, int maxlen, int* out);           /* huffman.cpp:122-154 */
void zzo_generate(const int* lengths, int n, int* out_len, uint32_t* out_bits); /* huffman.h:49-81     */
uint32_t zzo_reverse(uint32_t v, int len);                                       /* huffman.cpp:11-33   */
int zzo_from_lengths(const int* lengths, int n, int* freqs19, uint8_t* out_value, uint8_t* out_payload);
int zzo_dist_bucket(int d);                                                      /* luts.cpp:116-1160   */
void zzo_length_record(int len, int* sym, int* extra, int* extra_bits);          /* luts.cpp:5-58       */
void zzo_fixed_code(int sym, int* len, uint32_t* bits);                          /* fixedhuffmanluts.cpp:5 */
void zzo_fixed_lcode(int mlen, int* len, uint32_t* bits);                        /* fixedhuffmanluts.cpp:8-46 */
void zzo_fixed_dcode(int bucket, int* len, uint32_t* bits);                      /* fixedhuffmanluts.cpp:49-55 */
uint64_t zzo_bitstream(const uint64_t* bits, const int* counts, int n, uint8_t* out, uint64_t cap,
                       int* before_flush);                                       /* outputbitstream.h:83-124 */

#ifdef __cplusplus
}
#endif
#endif
