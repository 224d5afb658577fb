/* oracle/zzoracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the zzflate encoder path in plain C11. It is the parity checker for the HIP
 * kernels: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * library (zzflate_amd/csrc) never links, includes or calls anything in this directory.
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference/zzflate/).
 * Tables are generated from RFC 1951 (sections 3.2.5, 3.2.6) rather than copied. Where the reference has
 * defects the restatement emits the valid encoding the reference would have produced
 * had its reads stopped at the end of the data:
 *   D1/D2  level-1 match lengths are clamped to the bytes left in the block;
 *   D3     level->=2 literals are counted by position, not through the record array;
 *   D4     backward match extension stops at input offset 0;
 *   D11    (found while pinning this file, not in SURVEY.md) level >= 2: when more than 258 pending
 *          literals match backward (a repeat of period >= 259 met with a cold table, e.g. at a packet
 *          start), the reference clamps the match to 258 so that it ends BEFORE the probe position,
 *          then re-probes positions it already inserted, finds distance 0, emits a distance-0 match
 *          and indexes distanceLut/distanceFrequencies out of bounds (heap corruption; the stream is
 *          always invalid). Here backward extension is capped at 258 bytes, so a match always reaches
 *          the probe position; this changes nothing whenever the reference's output is valid;
 *   D5-D7  64-bit sizes, always-correct Adler-32.
 * so that "bit-identical wherever the reference's output is a valid encoding of the input" holds.
 * D8 (empty input -> no block) and D9 (silent truncation on a too-small buffer) are restated as they
 * are; the product library diverges there on purpose and its tests say so.
 *
 * Parity pinned: tests/test_oracle_vs_ref.py (against oracle/_ref, the compiled reference) and
 * tests/test_oracle_golden.py (against committed hashes of the reference's outputs).
 */
#define _POSIX_C_SOURCE 200809L
#include "zzoracle.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------------
 * Constant tables, generated at first use from RFC 1951 3.2.5 (luts.cpp:5-110 hold the same data)
 * ---------------------------------------------------------------------------------------------- */
enum { HASH_BITS = 13, HASH_SIZE = 1 << HASH_BITS,   /* encoder.h:41-43 */
       MAX_RECORDS = 20000,                          /* encoder.h:44    */
       BATCH_LEN = 16384,                            /* encoder.cpp:227 (min(16384,...)) */
       MAX_DIST = 0x8000, MAX_LEN = 258 };           /* encoder.h:46-47 */

typedef struct { int len; uint32_t bits; } code_t;   /* outputbitstream.h:14-24 */

static int g_tables_ready = 0;
static uint16_t g_len_sym[259];        /* lengthTable[].code          luts.cpp:5-58  */
static uint8_t g_len_extra_val[259];   /* lengthTable[].extraBits                    */
static uint8_t g_len_extra_bits[259];  /* lengthTable[].extraBitLength               */
static uint8_t g_sym_extra_bits[286];  /* extraLengthBits             luts.cpp:66-77 */
static uint16_t g_dist_base[30];       /* distanceTable               luts.cpp:79-110 */
static uint8_t g_dist_extra[30];       /* extraDistanceBits           luts.cpp:64    */
static const uint8_t g_order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 }; /* RFC 1951 3.2.7; luts.cpp:62 */
static code_t g_fix_codes[288];        /* codes_f   fixedhuffmanluts.cpp:5     */
static code_t g_fix_lcodes[259];       /* lcodes_f  fixedhuffmanluts.cpp:8-46  */
static code_t g_fix_dcodes[30];        /* dcodes_f  fixedhuffmanluts.cpp:49-55 */
static uint32_t g_crc_table[256];      /* Crc32Lookup crc.cpp:5-22 */

/* huffman.cpp:11-33: reverse the low `len` bits of v */
uint32_t zzo_reverse(uint32_t v, int len)
{
    uint32_t r = 0;
    for (int i = 0; i < len; ++i) r |= ((v >> i) & 1u) << (len - 1 - i);
    return r;
}

/* huffman.h:49-81: canonical code assignment (RFC 1951 3.2.2), stored bit-reversed for LSB-first output.
 * Codes of zero-length symbols are left untouched, as in the reference. */
static void generate_codes(const int* lengths, int n, code_t* codes)
{
    int bl_count[16] = { 0 };
    for (int i = 0; i < n; ++i) bl_count[lengths[i]]++;
    uint32_t next_code[16] = { 0 };
    uint32_t c = 0;
    bl_count[0] = 0;
    for (int b = 1; b < 16; ++b) {
        c = (c + (uint32_t)bl_count[b - 1]) << 1;
        next_code[b] = c;
    }
    for (int i = 0; i < n; ++i) {
        int len = lengths[i];
        if (len <= 0) continue;
        if (next_code[len] >> len) break; /* huffman.h:75-76: over-subscribed code, never for real trees */
        codes[i].len = len;
        codes[i].bits = zzo_reverse(next_code[len], len);
        next_code[len]++;
    }
}

/* encoder.cpp:121-124 Merge: symbol code first, then the extra bits above it */
static code_t merge_code(code_t first, int extra_bits, uint32_t extra_val)
{
    code_t r;
    r.len = first.len + extra_bits;
    r.bits = (extra_val << first.len) | first.bits;
    return r;
}

static void init_tables(void)
{
    if (g_tables_ready) return;
    /* length symbols: RFC 1951 3.2.5 table 1 */
    int base = 3;
    memset(g_sym_extra_bits, 0, sizeof g_sym_extra_bits);
    g_len_sym[0] = g_len_sym[1] = g_len_sym[2] = 0;
    g_len_extra_val[0] = g_len_extra_val[1] = g_len_extra_val[2] = 0;
    g_len_extra_bits[0] = g_len_extra_bits[1] = g_len_extra_bits[2] = 0;
    for (int sym = 257; sym <= 284; ++sym) {
        int eb = sym < 265 ? 0 : (sym - 261) / 4;
        g_sym_extra_bits[sym] = (uint8_t)eb;
        for (int k = 0; k < (1 << eb) && base + k <= 258; ++k) {
            g_len_sym[base + k] = (uint16_t)sym;
            g_len_extra_val[base + k] = (uint8_t)k;
            g_len_extra_bits[base + k] = (uint8_t)eb;
        }
        base += 1 << eb;
    }
    /* length 258 has its own symbol 285 with no extra bits (overrides 284+31) */
    g_len_sym[258] = 285; g_len_extra_val[258] = 0; g_len_extra_bits[258] = 0;
    g_sym_extra_bits[285] = 0;
    /* distance symbols: RFC 1951 3.2.5 table 2 */
    int dbase = 1;
    for (int d = 0; d < 30; ++d) {
        int eb = d < 4 ? 0 : (d - 2) / 2;
        g_dist_base[d] = (uint16_t)dbase;
        g_dist_extra[d] = (uint8_t)eb;
        dbase += 1 << eb;
    }
    /* fixed Huffman code: RFC 1951 3.2.6 (huffman.cpp:35-51 defaultTableLengths) */
    int fl[288];
    for (int i = 0; i < 288; ++i) fl[i] = (i <= 143 || i >= 280) ? 8 : (i <= 255 ? 9 : 7);
    memset(g_fix_codes, 0, sizeof g_fix_codes);
    generate_codes(fl, 288, g_fix_codes);
    for (int l = 0; l < 259; ++l)
        g_fix_lcodes[l] = merge_code(g_fix_codes[g_len_sym[l]], g_len_extra_bits[l], g_len_extra_val[l]);
    for (int d = 0; d < 30; ++d) { g_fix_dcodes[d].len = 5; g_fix_dcodes[d].bits = zzo_reverse((uint32_t)d, 5); }
    /* crc.cpp:5-20 PrepareTable(0xEDB88320) */
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int j = 0; j < 8; ++j) c = (c >> 1) ^ ((c & 1u) * 0xEDB88320u);
        g_crc_table[i] = c;
    }
    g_tables_ready = 1;
}

/* distanceLut[d] (luts.cpp:116-1160) == FindDistance(d) (encoder.cpp:51-61): bucket of distance d */
int zzo_dist_bucket(int d)
{
    init_tables();
    if (d <= 0) return 255;
    for (int n = 1; n < 30; ++n)
        if (d < g_dist_base[n]) return n - 1;
    return d <= 32768 ? 29 : -1;
}

static uint8_t g_dist_lut[32769];
static int g_dist_lut_ready = 0;
static void init_dist_lut(void)
{
    if (g_dist_lut_ready) return;
    for (int d = 0; d <= 32768; ++d) g_dist_lut[d] = (uint8_t)zzo_dist_bucket(d);
    g_dist_lut_ready = 1;
}

void zzo_length_record(int len, int* sym, int* extra, int* extra_bits)
{
    init_tables();
    *sym = g_len_sym[len]; *extra = g_len_extra_val[len]; *extra_bits = g_len_extra_bits[len];
}
void zzo_fixed_code(int sym, int* len, uint32_t* bits) { init_tables(); *len = g_fix_codes[sym].len; *bits = g_fix_codes[sym].bits; }
void zzo_fixed_lcode(int mlen, int* len, uint32_t* bits) { init_tables(); *len = g_fix_lcodes[mlen].len; *bits = g_fix_lcodes[mlen].bits; }
void zzo_fixed_dcode(int b, int* len, uint32_t* bits) { init_tables(); *len = g_fix_dcodes[b].len; *bits = g_fix_dcodes[b].bits; }

/* ------------------------------------------------------------------------------------------------
 * Bit packer -- outputbitstream.h:47-224
 * ---------------------------------------------------------------------------------------------- */
#define CHUNK_CAP 1000000 /* outputbitstream.h:183 */

typedef struct {
    uint8_t* out;         /* physical destination (all chunks concatenated in chunk mode) */
    uint64_t cap;
    uint64_t pos;         /* bytes stored so far == stream - start summed over chunks */
    uint64_t acc;         /* _bitBuffer */
    int used;             /* _usedBitCount */
    int fixed;            /* fixedOutputBuffer */
    /* chunk mode (stream == nullptr in the reference): */
    int nchunks;
    uint64_t chunk_start; /* pos at which the current chunk began */
    uint64_t* chunk_sizes; int max_chunks; int chunks_reported;
    int overflow;
} bits_t;

static void bs_init(bits_t* b, uint8_t* out, uint64_t cap, int fixed)
{
    memset(b, 0, sizeof *b);
    b->out = out; b->cap = cap; b->fixed = fixed;
}

static void bs_store(bits_t* b, const void* p, uint64_t k)
{
    if (b->pos + k <= b->cap) memcpy(b->out + b->pos, p, k);
    else b->overflow = 1;
    b->pos += k;
}

/* outputbitstream.h:83-98 AppendToBitStream */
static void bs_put(bits_t* b, uint64_t bits, int count)
{
    b->acc |= bits << b->used;
    b->used += count;
    if (b->used < 64) return;
    bs_store(b, &b->acc, 8); /* little-endian host, as the reference's unaligned 8-byte store :217-221 */
    b->used -= 64;
    int sh = count - b->used;
    b->acc = sh >= 64 ? 0 : bits >> sh;
}
static void bs_put_code(bits_t* b, code_t c) { bs_put(b, c.bits, c.len); }
/* outputbitstream.h:100-103 */
static void bs_pad(bits_t* b) { bs_put(b, 0, (-b->used) & 7); }
/* outputbitstream.h:105-124 (chunk bookkeeping handled in bs_finish) */
static void bs_flush(bits_t* b)
{
    if (b->used != 0) {
        bs_pad(b);
        while (b->used >= 8) {
            uint8_t byte = (uint8_t)(b->acc & 0xFF);
            bs_store(b, &byte, 1);
            b->used -= 8;
            b->acc >>= 8;
        }
    }
}
/* outputbitstream.h:126-152 */
static void bs_u16(bits_t* b, uint32_t v) { bs_pad(b); bs_put(b, v & 0xFFFF, 16); }
static void bs_u32(bits_t* b, uint32_t v) { bs_pad(b); bs_put(b, v, 32); }
static void bs_be32(bits_t* b, uint32_t v)
{
    bs_pad(b);
    bs_put(b, (v >> 24) & 0xFF, 8); bs_put(b, (v >> 16) & 0xFF, 8);
    bs_put(b, (v >> 8) & 0xFF, 8);  bs_put(b, v & 0xFF, 8);
}
/* outputbitstream.h:155-160 */
static void bs_bytes(bits_t* b, const uint8_t* p, uint64_t k) { bs_flush(b); bs_store(b, p, k); }

/* outputbitstream.h:167 AvailableBytes */
static int64_t bs_available(const bits_t* b)
{
    if (b->fixed) return (int64_t)b->cap - (int64_t)b->pos;
    if (b->nchunks == 0) return 0;
    return (int64_t)(b->chunk_start + CHUNK_CAP) - (int64_t)b->pos;
}
/* outputbitstream.h:171-201 EnsureOutputLength / IsEnough */
static int64_t bs_ensure(bits_t* b, int64_t length)
{
    int64_t avail = bs_available(b);
    if (b->fixed) return avail;
    if (avail > 2 * length || avail > (1 << 18)) return avail;
    if (b->nchunks != 0) {
        if (b->chunk_sizes && b->chunks_reported < b->max_chunks)
            b->chunk_sizes[b->chunks_reported] = b->pos - b->chunk_start;
        b->chunks_reported++;
    }
    b->chunk_start = b->pos; /* pending accumulator bits carry over into the new chunk */
    b->nchunks++;
    return CHUNK_CAP;
}
static void bs_finish_chunks(bits_t* b)
{
    if (!b->fixed && b->nchunks != 0) {
        if (b->chunk_sizes && b->chunks_reported < b->max_chunks)
            b->chunk_sizes[b->chunks_reported] = b->pos - b->chunk_start;
        b->chunks_reported++;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Encoder state -- encoder.h:37-116
 * ---------------------------------------------------------------------------------------------- */
typedef struct { int32_t start; uint16_t dist; uint16_t len; } token_t; /* position form of compressionRecord, encoder.h:29-34 */

typedef struct {
    int level;
    int64_t table[HASH_SIZE];  /* hashtable, encoder.h:76 (int in the reference, D7) */
    bits_t bs;
    const uint8_t* gbase;      /* first byte of the whole input (backward extension stops here, D4) */
    const uint8_t* gend;       /* one past the last readable byte; reads beyond see zeros */
    token_t* tokens; int ntok; int tokcap;
} enc_t;

static void enc_init(enc_t* e, int level, uint8_t* out, uint64_t cap, int fixed, const uint8_t* gbase,
                     const uint8_t* gend)
{
    init_tables();
    init_dist_lut();
    e->level = level;
    for (int i = 0; i < HASH_SIZE; ++i) e->table[i] = -100000; /* encoder.cpp:533-536 */
    bs_init(&e->bs, out, cap, fixed);
    e->gbase = gbase; e->gend = gend;
    e->tokens = NULL; e->ntok = 0; e->tokcap = 0;
}
static void enc_free(enc_t* e) { free(e->tokens); e->tokens = NULL; }

static inline uint64_t load_le(const enc_t* e, const uint8_t* p, int k)
{
    uint64_t v = 0;
    if (p + k <= e->gend) { memcpy(&v, p, (size_t)k); return v; }
    for (int i = 0; i < k; ++i)
        if (p + i < e->gend) v |= (uint64_t)p[i] << (8 * i);
    return v;
}

/* encoder.cpp:11-17 CalcHash: 13-bit multiplicative hash of the 3 bytes at p */
static inline uint32_t calc_hash(const enc_t* e, const uint8_t* p)
{
    uint32_t v = (uint32_t)load_le(e, p, 4);
    v = ((v << 8) >> 8) * 0x00d68664u;
    return v >> (32 - HASH_BITS);
}

/* common-prefix length of a[0..) and b[0..), at most maxlen. Restates the 64-bit xor + ZeroCount
 * (gcc.h:10-13) followed by remain() (encoder.cpp:64-90). */
static inline int match_forward(const enc_t* e, const uint8_t* a, const uint8_t* b, int maxlen)
{
    int k = 0;
    while (k < maxlen) {
        uint8_t x = a + k < e->gend ? a[k] : 0;
        uint8_t y = b + k < e->gend ? b[k] : 0;
        if (x != y) break;
        ++k;
    }
    return k;
}

/* encoder.cpp:143-147 */
static void start_block(enc_t* e, int type, int final)
{
    bs_put(&e->bs, (uint64_t)(final ? 1 : 0), 1);
    bs_put(&e->bs, (uint64_t)type, 2);
}

/* encoder.cpp:135-141 WriteDistance */
static void write_distance(enc_t* e, const code_t* dcodes, int dist)
{
    int bucket = g_dist_lut[dist];
    bs_put_code(&e->bs, dcodes[bucket]);
    bs_put(&e->bs, (uint64_t)(dist - g_dist_base[bucket]), g_dist_extra[bucket]);
}

/* encoder.cpp:320-327 FixHashTable */
static void fix_hash(enc_t* e, int64_t offset)
{
    for (int i = 0; i < HASH_SIZE; ++i) e->table[i] -= offset;
}

/* encoder.cpp:482-502 WriteUncompressedBlock */
static int64_t block_stored(enc_t* e, const uint8_t* src, int64_t byteCount, int final)
{
    int64_t length = byteCount < 0xFFFF ? byteCount : 0xFFFF;
    int64_t avail = bs_ensure(&e->bs, 6 + length);
    if (avail <= 40) return 0;
    if (length > avail - 6) length = avail - 6;
    start_block(e, 0, final && length == byteCount);
    bs_pad(&e->bs);
    bs_u16(&e->bs, (uint32_t)length);
    bs_u16(&e->bs, (uint32_t)(~length) & 0xFFFF);
    bs_bytes(&e->bs, src, (uint64_t)length);
    return length;
}

/* encoder.cpp:305-317 UncompressedFallback */
static int64_t stored_fallback(enc_t* e, int64_t length, const uint8_t* src, int final)
{
    int64_t written = 0;
    while (written < length) {
        int64_t c = block_stored(e, src + written, length - written, final);
        if (c <= 0) return 0;
        written += c;
    }
    return written;
}

/* encoder.cpp:329-373 WriteBlockFixedHuff: level 1, greedy single-probe LZ fused with fixed-Huffman
 * emission. Hash key = bytes i+1..i+3, stored value = i, compare from i => minimum match 4 (:356). */
static int64_t block_fixed(enc_t* e, const uint8_t* src, int64_t byteCount, int final)
{
    int64_t avail = bs_ensure(&e->bs, byteCount) - 1;      /* :331 */
    int64_t bitsAvail = avail * 8;                          /* :332 (int overflow D6 not restated) */
    int64_t n = bitsAvail / 9 - 8;                          /* :333 */
    if (n > byteCount) n = byteCount;
    if (n != byteCount) final = 0;                          /* :334-337 */
    start_block(e, 1, final);                               /* :338 */
    for (int64_t i = 0; i < n; ++i) {
        const uint8_t* p = src + i;
        uint32_t h = calc_hash(e, p + 1);                   /* :344 */
        int64_t dist = i - e->table[h];                     /* :345 */
        e->table[h] = i;                                    /* :346 */
        if (dist > 0 && dist <= MAX_DIST) {                 /* :348, inclusive */
            int64_t left = n - i;
            int maxlen = left < MAX_LEN ? (int)left : MAX_LEN; /* :354 remain(...,n-i) + D1 clamp */
            int len = match_forward(e, p, p - dist, maxlen);   /* :350-354 */
            if (len > 3) {                                  /* :356 */
                bs_put_code(&e->bs, g_fix_lcodes[len]);     /* :358 */
                write_distance(e, g_fix_dcodes, (int)dist); /* :359 */
                i += len - 1;                               /* :361-362 */
                continue;
            }
        }
        bs_put_code(&e->bs, g_fix_codes[*p]);               /* :367 */
    }
    fix_hash(e, n);                                         /* :370 */
    bs_put_code(&e->bs, g_fix_codes[256]);                  /* :371 */
    return n;
}

/* ---- Huffman builder: huffman.cpp:53-154 with libstdc++'s heap algorithm (bits/stl_heap.h, GCC 11:
 * __push_heap :134-148, __adjust_heap :223-248, __pop_heap :253-265, __make_heap :339-360) restated,
 * because the comparator only orders by frequency and the element movements decide ties. ---------- */
typedef struct { int freq; int id; } hrec_t;                 /* huffman.h:10-14  */
typedef struct { int freq; int left; int right; int bits; } titem_t; /* huffman.h:16-22 */

/* comparator `greater`: huffman.cpp:55-62 */
#define HGREATER(a, b) ((a).freq > (b).freq)

static void heap_push(hrec_t* h, int hole, int top, hrec_t value)
{
    int parent = (hole - 1) / 2;
    while (hole > top && HGREATER(h[parent], value)) {
        h[hole] = h[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    h[hole] = value;
}
static void heap_adjust(hrec_t* h, int hole, int len, hrec_t value)
{
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (HGREATER(h[child], h[child - 1])) child--;
        h[hole] = h[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        h[hole] = h[child - 1];
        hole = child - 1;
    }
    heap_push(h, hole, top, value);
}
static void heap_make(hrec_t* h, int len)
{
    if (len < 2) return;
    int parent = (len - 2) / 2;
    for (;;) {
        hrec_t v = h[parent];
        heap_adjust(h, parent, len, v);
        if (parent == 0) return;
        parent--;
    }
}
/* std::pop_heap on [0,len): moves the top to h[len-1] */
static void heap_pop(hrec_t* h, int len)
{
    if (len > 1) {
        hrec_t v = h[len - 1];
        h[len - 1] = h[0];
        heap_adjust(h, 0, len - 1, v);
    }
}

/* huffman.cpp:67-120 CalculateTree. tree must hold 2*n items. returns max leaf depth, *ntree items. */
static int calculate_tree(const int* freqs, int n, int minFreq, titem_t* tree, int* ntree)
{
    hrec_t recs[288];
    int nrec = 0, nt = 0;
    for (int i = 0; i < n; ++i) {
        if (freqs[i] == 0) { tree[nt++] = (titem_t){ 0, i, -1, 0 }; continue; }
        int f = freqs[i] > minFreq ? freqs[i] : minFreq;
        tree[nt++] = (titem_t){ f, i, -1, 0 };
        recs[nrec++] = (hrec_t){ f, i };
    }
    heap_make(recs, nrec);
    while (nrec >= 2) {
        heap_pop(recs, nrec); hrec_t a = recs[--nrec];
        heap_pop(recs, nrec); hrec_t b = recs[--nrec];
        int sum = a.freq + b.freq;
        tree[nt++] = (titem_t){ sum, a.id, b.id, 0 };
        recs[nrec++] = (hrec_t){ sum, nt - 1 };
        heap_push(recs, nrec - 1, 0, recs[nrec - 1]);
    }
    int maxLength = 0;
    for (int i = nt - 1; i != 0; --i) {              /* :108, index 0 skipped as in the reference */
        titem_t it = tree[i];
        if (it.right == -1) { if (it.bits > maxLength) maxLength = it.bits; continue; }
        tree[it.left].bits = it.bits + 1;
        tree[it.right].bits = it.bits + 1;
    }
    *ntree = nt;
    return maxLength;
}

/* huffman.cpp:122-154 CalcLengths: frequency-floor length limiting (not package-merge) */
void zzo_calc_lengths(const int* freqs, int n, int maxlen, int* out)
{
    titem_t tree[2 * 288];
    int minFreq = 0, nt = 0;
    for (;;) {
        int mx = calculate_tree(freqs, n, minFreq, tree, &nt);
        if (mx <= maxlen) {
            for (int i = 0; i < nt; ++i) {
                if (tree[i].right != -1) break;
                out[tree[i].left] = tree[i].freq == 0 ? 0 : (tree[i].bits > 1 ? tree[i].bits : 1);
            }
            return;
        }
        int total = 0;
        for (int i = 0; i < n; ++i) total += freqs[i];
        int step = total / (1 << maxlen);
        minFreq += step > 1 ? step : 1;
    }
}

void zzo_generate(const int* lengths, int n, int* out_len, uint32_t* out_bits)
{
    code_t c[288];
    memset(c, 0, sizeof c);
    generate_codes(lengths, n, c);
    for (int i = 0; i < n; ++i) { out_len[i] = c[i].len; out_bits[i] = c[i].bits; }
}

typedef struct { uint8_t value, payload; } lrec_t;          /* huffman.h:26-35 */

/* huffman.cpp:158-189 AddRecords */
static int add_records(lrec_t* v, int nv, int value, int count)
{
    if (count == 0) return nv;
    if (value == 0) {
        while (count >= 3) {
            int w = count < 138 ? count : 138;
            count -= w;
            v[nv++] = (lrec_t){ (uint8_t)(w < 11 ? 17 : 18), (uint8_t)w };
        }
    } else {
        v[nv++] = (lrec_t){ (uint8_t)value, 0 };
        count--;
        while (count >= 3) {
            int w = count < 6 ? count : 6;
            count -= w;
            v[nv++] = (lrec_t){ 16, (uint8_t)w };
        }
    }
    for (int i = 0; i < count; ++i) v[nv++] = (lrec_t){ (uint8_t)value, 0 };
    return nv;
}
/* huffman.cpp:191-216 FromLengths: RLE of one code-length array; meta frequencies accumulate */
static int from_lengths(const int* lengths, int n, int* freqs19, lrec_t* out)
{
    int nv = 0, cur = -1, count = 0;
    for (int i = 0; i < n; ++i) {
        if (lengths[i] == cur) { count++; continue; }
        nv = add_records(out, nv, cur, count);
        cur = lengths[i];
        count = 1;
    }
    nv = add_records(out, nv, cur, count);
    for (int i = 0; i < nv; ++i) freqs19[out[i].value]++;
    return nv;
}
int zzo_from_lengths(const int* lengths, int n, int* freqs19, uint8_t* out_value, uint8_t* out_payload)
{
    lrec_t r[320];
    int nv = from_lengths(lengths, n, freqs19, r);
    for (int i = 0; i < nv; ++i) { out_value[i] = r[i].value; out_payload[i] = r[i].payload; }
    return nv;
}

/* encoder.cpp:20-46 WriteLengths: emit==0 only counts bits (LengthCounter) */
static int64_t write_lengths(enc_t* e, const lrec_t* recs, int n, const code_t* meta, int emit)
{
    int64_t bits = 0;
    for (int i = 0; i < n; ++i) {
        code_t c = meta[recs[i].value];
        int eb = 0; uint32_t ev = 0;
        switch (recs[i].value) {
        case 16: eb = 2; ev = (uint32_t)recs[i].payload - 3; break;
        case 17: eb = 3; ev = (uint32_t)recs[i].payload - 3; break;
        case 18: eb = 7; ev = (uint32_t)recs[i].payload - 11; break;
        default: break;
        }
        bits += c.len + eb;
        if (emit) { bs_put_code(&e->bs, c); if (eb) bs_put(&e->bs, ev, eb); }
    }
    return bits;
}

static void push_token(enc_t* e, int64_t start, int64_t dist, int len)
{
    if (e->ntok == e->tokcap) {
        e->tokcap = e->tokcap ? e->tokcap * 2 : 4096;
        e->tokens = (token_t*)realloc(e->tokens, sizeof(token_t) * (size_t)e->tokcap);
    }
    e->tokens[e->ntok++] = (token_t){ (int32_t)start, (uint16_t)dist, (uint16_t)len };
}

/* encoder.cpp:375-440 FirstPass (+ :474-480 AddHashEntries): one batch [startPos,end) of the token pass.
 * Tokens are kept by position; *nrec counts what the reference's record array would hold. */
static int64_t first_pass(enc_t* e, const uint8_t* src, int64_t startPos, int64_t end, int* nrec)
{
    if (startPos == end) return startPos;                    /* :377-378 */
    int64_t backRefEnd = startPos + 1;                       /* :380 */
    int64_t j = startPos + 1;                                /* :383, the batch's first byte is never probed */
    while (j < end) {
        const uint8_t* s = src + j;
        uint32_t h = calc_hash(e, s);                        /* :388 */
        int64_t dist = j - e->table[h];                      /* :389 */
        e->table[h] = j;                                     /* :390 */
        if (dist >= MAX_DIST) { j++; continue; }             /* :392-396 */
        int fwd = match_forward(e, s, s - dist, MAX_LEN);    /* :399-402 */
        /* :404 countMatchBackward(s, s-dist, j-backRefEnd), stopping at input offset 0 (D4) */
        int64_t maxBack = j - backRefEnd;
        int64_t room = (s - dist) - e->gbase;
        if (maxBack > room) maxBack = room;
        if (maxBack > MAX_LEN) maxBack = MAX_LEN;            /* D11, see file header */
        int bwd = 0;
        while (bwd < maxBack && s[-1 - bwd] == (s - dist)[-1 - bwd]) bwd++;
        int m = fwd + bwd;                                   /* :406 */
        if (m < 4) { j++; continue; }                        /* :407-411 */
        if (m > MAX_LEN) m = MAX_LEN;                        /* :412-415 */
        int64_t matchStart = j - bwd;                        /* :416 */
        for (int64_t q = matchStart + 1; q < matchStart + 1 + m; ++q) /* :418 AddHashEntries */
            e->table[calc_hash(e, src + q)] = q;
        push_token(e, matchStart, dist, m);                  /* :420 */
        (*nrec)++;
        backRefEnd = matchStart + m;                         /* :422 */
        j = backRefEnd + 1;                                  /* :424 */
        if (*nrec == MAX_RECORDS) { end = 0; break; }        /* :426-430 */
    }
    if (backRefEnd > end) return backRefEnd;                 /* :435-436 */
    (*nrec)++;                                               /* :438 closing literal record */
    return end;
}

/* ---- optimal length-limited code lengths by package-merge -- NOT in the reference (its limiter is the frequency
 * floor above, huffman.cpp:122-154); used by the extended levels 4..6 only. Larmore/Hirschberg 1990 in
 * the list form: the symbols with a non-zero count, sorted by (count, symbol), are the leaves; list 1 is the leaves;
 * list l+1 is the merge of the leaves with the packages of list l (consecutive pairs, weights added; an unpaired last
 * item is dropped), a leaf going first where weights are equal; of the last list (l = maxlen) the first 2m-2 items
 * are taken, of every earlier list twice as many items as packages were taken from its successor; a symbol's length
 * is the number of lists in which its leaf is among the items taken. Since every list holds the leaves in the same
 * order, "the leaves taken from list l" is a count a_l, and the symbol of sorted rank r gets length #{l : r < a_l}.
 * This is also the product's formulation (zz_level6.h), so ties are decided identically. ---------------------------- */
static void pm_lengths(const int* freqs, int n, int maxlen, int* out)
{
    int sym[288]; int64_t w[288];
    int m = 0;
    for (int i = 0; i < n; ++i) out[i] = 0;
    for (int i = 0; i < n; ++i) if (freqs[i] != 0) { sym[m] = i; w[m] = freqs[i]; m++; }
    if (m == 0) return;
    if (m == 1) { out[sym[0]] = 1; return; }
    for (int i = 1; i < m; ++i) {                           /* insertion sort by (count, symbol): stable on the symbol order */
        int s0 = sym[i]; int64_t w0 = w[i]; int j = i;
        while (j > 0 && w[j - 1] > w0) { sym[j] = sym[j - 1]; w[j] = w[j - 1]; j--; }
        sym[j] = s0; w[j] = w0;
    }
    /* isleaf[l][k]: item k of list l+1 is a leaf. Lists are cut at 2m-2 items: nothing behind that is ever taken. */
    static _Thread_local uint8_t isleaf[16][2 * 288];
    int64_t cur[2 * 288], nxt[2 * 288];
    int ncur = m, lim = 2 * m - 2;
    int lens_l[16];
    for (int k = 0; k < m; ++k) { cur[k] = w[k]; isleaf[0][k] = 1; }
    lens_l[0] = m;
    for (int l = 1; l < maxlen; ++l) {
        int np = ncur / 2, a = 0, b = 0, k = 0;
        while (k < lim && (a < m || b < np)) {
            int64_t pw = b < np ? cur[2 * b] + cur[2 * b + 1] : 0;
            if (a < m && (b >= np || w[a] <= pw)) { nxt[k] = w[a++]; isleaf[l][k] = 1; }
            else { nxt[k] = pw; isleaf[l][k] = 0; b++; }
            k++;
        }
        memcpy(cur, nxt, sizeof(int64_t) * (size_t)k);
        ncur = k; lens_l[l] = k;
    }
    int need = lim;
    for (int l = maxlen - 1; l >= 0; --l) {
        if (need > lens_l[l]) need = lens_l[l];
        int a = 0;
        for (int k = 0; k < need; ++k) a += isleaf[l][k];
        for (int r = 0; r < a; ++r) out[sym[r]]++;
        need = 2 * (need - a);
    }
}

/* The back half of WriteBlock2Pass (encoder.cpp:253-303): histograms over the tokens, code construction, exact size,
 * stored fallback or dynamic header + body. `pm`: code lengths by package-merge (extended levels) instead of the
 * reference's CalcLengths. */
static int64_t emit_dynamic(enc_t* e, const uint8_t* src, int64_t length, int64_t byteCount, int final, int pm)
{
    void (*calc)(const int*, int, int, int*) = pm ? pm_lengths : zzo_calc_lengths;
    /* :253 + :442-471 GetFrequencies, by position */
    int symF[286] = { 0 }, distF[30] = { 0 };
    {
        int64_t pos = 0;
        for (int t = 0; t < e->ntok; ++t) {
            token_t tk = e->tokens[t];
            for (; pos < tk.start; ++pos) symF[src[pos]]++;
            symF[g_len_sym[tk.len]]++;
            distF[g_dist_lut[tk.dist]]++;
            pos += tk.len;
        }
        for (; pos < length; ++pos) symF[src[pos]]++;
        symF[256]++;
    }
    /* :255-265 ComputeCodes x2 (:171-176), CountBits (:178-187), meta code */
    int lens[286]; int metaF[19] = { 0 };
    code_t codes[286], dcodes[30], meta[19];
    memset(codes, 0, sizeof codes); memset(dcodes, 0, sizeof dcodes); memset(meta, 0, sizeof meta);
    lrec_t symRecs[320], distRecs[64];
    int64_t bits = 0;
    calc(symF, 286, 15, lens);
    generate_codes(lens, 286, codes);
    int nSymRecs = from_lengths(lens, 286, metaF, symRecs);
    for (int i = 0; i < 286; ++i) bits += (int64_t)symF[i] * (lens[i] + g_sym_extra_bits[i]);
    calc(distF, 30, 15, lens);
    generate_codes(lens, 30, dcodes);
    int nDistRecs = from_lengths(lens, 30, metaF, distRecs);
    for (int i = 0; i < 30; ++i) bits += (int64_t)distF[i] * (lens[i] + g_dist_extra[i]);
    int metaLens[19];
    calc(metaF, 19, 7, metaLens);
    generate_codes(metaLens, 19, meta);
    /* :267-271 exact size */
    int64_t total = 3 + 5 + 5 + 4 + 3 * 19 + bits;
    total += write_lengths(e, symRecs, nSymRecs, meta, 0);
    total += write_lengths(e, distRecs, nDistRecs, meta, 0);
    int64_t required = (total + 8) / 8;
    if (required >= length) return stored_fallback(e, length, src, final); /* :273-274 */
    int64_t avail = bs_ensure(&e->bs, required);              /* :276-278 */
    if (avail < required) return 0;
    start_block(e, 2, length < byteCount ? 0 : final);       /* :280 */
    bs_put(&e->bs, 286 - 257, 5);                            /* :283-285 */
    bs_put(&e->bs, 30 - 1, 5);
    bs_put(&e->bs, 19 - 4, 4);
    for (int i = 0; i < 19; ++i) bs_put(&e->bs, (uint64_t)metaLens[g_order[i]], 3); /* :287-290 */
    write_lengths(e, symRecs, nSymRecs, meta, 1);            /* :292-293 */
    write_lengths(e, distRecs, nDistRecs, meta, 1);
    /* :296 CreateMergedLengthCodes (:126-133) folded into emission; :298 WriteRecords (:149-169) */
    {
        int64_t pos = 0;
        for (int t = 0; t < e->ntok; ++t) {
            token_t tk = e->tokens[t];
            for (; pos < tk.start; ++pos) bs_put_code(&e->bs, codes[src[pos]]);
            bs_put_code(&e->bs, merge_code(codes[g_len_sym[tk.len]], g_len_extra_bits[tk.len], g_len_extra_val[tk.len]));
            write_distance(e, dcodes, tk.dist);
            pos += tk.len;
        }
        for (; pos < length; ++pos) bs_put_code(&e->bs, codes[src[pos]]);
    }
    bs_put_code(&e->bs, codes[256]);                         /* :300 */
    return length;
}

/* encoder.cpp:217-303 WriteBlock2Pass */
static int64_t block_dynamic(enc_t* e, const uint8_t* src, int64_t byteCount, int final)
{
    int nrec = 0;
    e->ntok = 0;
    int64_t target = byteCount - MAX_LEN > 0 ? byteCount - MAX_LEN : 0; /* :222 */
    int64_t length = 0;
    while (target > 0 && nrec < MAX_RECORDS) {               /* :225-234 */
        int64_t batch = target < BATCH_LEN ? target : BATCH_LEN;
        int64_t newEnd = first_pass(e, src, length, length + batch, &nrec);
        target -= newEnd - length;
        length = newEnd;
    }
    if (target <= 0 && nrec < MAX_RECORDS) { nrec++; length = byteCount; } /* :236-245 */
    fix_hash(e, length);                                     /* :248 */
    return emit_dynamic(e, src, length, byteCount, final, 0);
}

/* ------------------------------------------------------------------------------------------------
 * Extended levels 4, 5, 6 -- NOT in the reference, which has one slot per hash, no chains, no lazy
 * matching, and rejects level > 3 (encoder.h:41-43,76; encoder.cpp:388-424; zzflate.cpp:201,230-234).
 * SURVEY.md 8f.2 defines the extension: bounded hash chains, lazy matching, package-merge code lengths.
 * This is the executable definition the product's levels 4..6 are tested against (packet mode only):
 *
 *   window   the last min(bytes in front of the packet, X.window) bytes in front of the packet may be matched;
 *   chains   every position q in [-window, target), target = n - 16 (so that the 16 bytes counted at a position lie
 *            inside the data; the reference's level 2 stops 258 short, encoder.cpp:222), is entered under a 13-bit
 *            hash of its FOUR bytes (x_hash4; the reference hashes three, encoder.cpp:11-17: four-byte keys waste
 *            far fewer chain entries on collisions), ascending; prev(q) = the nearest earlier entered position with
 *            the same hash;
 *   match    for q in [0, target): candidates prev(q), prev(prev(q)), ... -- at most X.depth of them, while the
 *            distance stays below 32768; a candidate's length is the common prefix with q, counted up to X_CAP = 16
 *            bytes; the longest wins, the nearer one among equals. It is a match if that length is >= 4;
 *   lazy     position q defers (stays a literal) if it has a match shorter than X_CAP and q+1 has a longer one
 *            (counted the same way) -- except where (q & 63) == 63 (the product works in groups of 64 positions);
 *   parse    greedy over the positions that have a match and do not defer; a match that reached X_CAP bytes is
 *            extended against its candidate up to 258 bytes (and the end of the data); nothing else depends on
 *            the parse, which is what lets the product find all matches in parallel;
 *   block    one dynamic block per packet as at level 2 (encoder.cpp:253-303: stored fallback, 286/30/19 header),
 *            code lengths by package-merge.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { int depth; int window; } xlevel_t;
enum { X_CAP = 16, X_TAIL = 16 };
static const xlevel_t g_xlevels[3] = { { 2, 8192 }, { 4, 32768 }, { 8, 32768 } };   /* levels 4, 5, 6 */
static inline uint32_t x_hash4(const enc_t* e, const uint8_t* p)      /* 13 bits of the FOUR bytes at p (Knuth's multiplier) */
{
    return ((uint32_t)load_le(e, p, 4) * 2654435761u) >> (32 - HASH_BITS);
}
static int64_t block_extended(enc_t* e, const uint8_t* src, int64_t n, int final, int64_t before)
{
    const xlevel_t X = g_xlevels[e->level - 4];
    const int64_t W = before < X.window ? before : X.window;
    const int64_t target = n - X_TAIL > 0 ? n - X_TAIL : 0;
    int32_t* prev = (int32_t*)malloc(sizeof(int32_t) * (size_t)(W + target + 1));
    uint8_t* L = (uint8_t*)calloc((size_t)target + 2, 1);
    uint16_t* Dd = (uint16_t*)calloc((size_t)target + 2, 2);
    const int32_t NONE = INT32_MIN;
    int32_t head[HASH_SIZE];
    for (int i = 0; i < HASH_SIZE; ++i) head[i] = NONE;
    for (int64_t q = -W; q < target; ++q) {                  /* chains: every position, ascending */
        uint32_t h = x_hash4(e, src + q);
        prev[q + W] = head[h];
        head[h] = (int32_t)q;
    }
    for (int64_t q = 0; q < target; ++q) {                   /* best of the chain, lengths counted up to X_CAP */
        int best = 0; int64_t bdist = 0;
        int32_t c = prev[q + W];
        for (int k = 0; k < X.depth && c != NONE && q - c < MAX_DIST; ++k, c = prev[c + W]) {
            int len = match_forward(e, src + q, src + c, X_CAP);    /* (q + X_CAP <= n: nothing to clamp) */
            if (len > best) { best = len; bdist = q - c; }
        }
        if (best >= 4) { L[q] = (uint8_t)best; Dd[q] = (uint16_t)bdist; }
    }
    e->ntok = 0;
    for (int64_t p = 0; p < target;) {                       /* greedy parse with one-step lazy evaluation */
        int len = L[p];
        /* ZZO_LAZY_EVERYWHERE (a measurement build only, tools/lazy_artefact.py: what the exception for lane 63 costs in ratio) */
#ifdef ZZO_LAZY_EVERYWHERE
        const int defer = len && len < X_CAP && L[p + 1] > len;
#else
        const int defer = len && len < X_CAP && (p & 63) != 63 && L[p + 1] > len;   /* (L[target] = 0) */
#endif
        if (!len || defer) { p++; continue; }
        if (len == X_CAP) len = match_forward(e, src + p, src + p - Dd[p], n - p < MAX_LEN ? (int)(n - p) : MAX_LEN);
        push_token(e, p, Dd[p], len);
        p += len;
    }
    free(prev); free(L); free(Dd);
    return emit_dynamic(e, src, n, n, final, 1);
}
void zzo_pm_lengths(const int* freqs, int n, int maxlen, int* out) { pm_lengths(freqs, n, maxlen, out); }

/* encoder.cpp:506-527 WriteDeflateBlock */
static int64_t write_block(enc_t* e, const uint8_t* src, int64_t len, int final)
{
    if (e->level == 0) return block_stored(e, src, len, final);
    if (e->level == 1) return block_fixed(e, src, len, final);
    if (e->level >= 4) return block_extended(e, src, len, final, src - e->gbase);   /* (packet mode only: len <= 32768) */
    if (len > 500000) { len = 500000; final = 0; }           /* :518-522 */
    return block_dynamic(e, src, len, final);
}

/* encoder.cpp:539-552 AddData */
static int add_data(enc_t* e, const uint8_t* start, const uint8_t* end, int final)
{
    while (start != end) {
        int64_t r = write_block(e, start, end - start, final);
        if (r <= 0) return 0;
        start += r;
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------------
 * Checksums
 * ---------------------------------------------------------------------------------------------- */
/* adler.cpp:17-43 adler32x -- with the modulo applied often enough to stay exact (D5) */
uint32_t zzo_adler32(uint32_t start, const uint8_t* p, uint64_t n)
{
    uint64_t a = start & 0xFFFF, b = start >> 16;
    while (n) {
        uint64_t k = n < 5552 ? n : 5552;
        for (uint64_t i = 0; i < k; ++i) { a += p[i]; b += a; }
        a %= 65521; b %= 65521;
        p += k; n -= k;
    }
    return (uint32_t)((b << 16) | a);
}
/* adler.cpp:5-15 combine: `second` was computed with start value 0 */
uint32_t zzo_adler_combine(uint32_t first, uint32_t second, uint64_t len2)
{
    uint64_t a = (first & 0xFFFF) + (second & 0xFFFF);
    uint64_t b = (first >> 16) + (second >> 16);
    b += (len2 % 65521) * (first & 0xFFFF);
    return (uint32_t)(((b % 65521) << 16) | (a % 65521));
}
/* crc.cpp:24-33 */
uint32_t zzo_crc32(const uint8_t* p, uint64_t n, uint32_t start)
{
    init_tables();
    uint32_t c = ~start;
    for (uint64_t i = 0; i < n; ++i) c = (c >> 8) ^ g_crc_table[(c & 0xFF) ^ p[i]];
    return ~c;
}
/* CRC-32 of A||B from crc(A), crc(B), |B|: crc(A) * x^(8|B|) mod P, xor crc(B). The reference has no
 * such function; checked against zzo_crc32 on split buffers. */
static uint32_t gf2_mulmod(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
    for (uint32_t m = 0x80000000u; m; m >>= 1) {
        if (a & m) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}
uint32_t zzo_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2)
{
    uint32_t xp = 0x80000000u;       /* x^0 in reflected form */
    uint32_t sq = 0x00800000u;       /* x^8 */
    for (uint64_t k = len2; k; k >>= 1) {
        if (k & 1) xp = gf2_mulmod(xp, sq);
        sq = gf2_mulmod(sq, sq);
    }
    return gf2_mulmod(crc1, xp) ^ crc2;
}

/* ------------------------------------------------------------------------------------------------
 * Containers and entry points -- zzflate.cpp
 * ---------------------------------------------------------------------------------------------- */
/* zzflate.cpp:28-48 */
static int header_bytes(int format, uint8_t* h)
{
    static const uint8_t gz[10] = { 0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xFF };
    switch (format) {
    case ZZO_ZLIB: {
        uint8_t cmf = (uint8_t)(8 | (7 << 4)), flg = 0;
        int rem = (cmf * 0x100 + flg) % 31;
        flg = (uint8_t)(flg | ((31 - rem) & 0xF)); /* FCHECK is a 4-bit field in the reference's header struct; 31-rem = 1 here */
        h[0] = cmf; h[1] = flg; return 2;
    }
    case ZZO_GZIP: memcpy(h, gz, 10); return 10;
    default: return 0;
    }
}
/* zzflate.cpp:170-192 AppendChecksum */
static void append_checksum(bits_t* b, int format, const uint8_t* src, uint64_t n)
{
    if (format == ZZO_ZLIB) bs_be32(b, zzo_adler32(1, src, n));
    else if (format == ZZO_GZIP) { bs_u32(b, zzo_crc32(src, n, 0)); bs_u32(b, (uint32_t)n); }
}

uint64_t zzo_encode(uint8_t* dest, uint64_t cap, const uint8_t* src, uint64_t n, int format, int level)
{
    uint8_t h[10];
    int hl = header_bytes(format, h);
    if (level < 0 || level > 3 || cap < (uint64_t)hl) return ZZO_ERROR; /* zzflate.cpp:229-234 */
    memcpy(dest, h, (size_t)hl);
    enc_t* e = (enc_t*)malloc(sizeof(enc_t));
    enc_init(e, level, dest + hl, cap - (uint64_t)hl, 1, src, src + n);  /* :84-95 */
    add_data(e, src, src + n, 1);
    bs_flush(&e->bs);
    uint64_t count = (uint64_t)hl + e->bs.pos;
    enc_free(e); free(e);
    bits_t t;
    bs_init(&t, count <= cap ? dest + count : dest, count <= cap ? cap - count : 0, 1); /* :237-239 */
    append_checksum(&t, format, src, n);
    bs_flush(&t);
    return count + t.pos;
}

uint64_t zzo_encode_callback(uint8_t* dest, uint64_t cap, const uint8_t* src, uint64_t n, int format,
                             int level, uint64_t* chunk_sizes, int max_chunks, int* nchunks)
{
    int calls = 0;
    if (nchunks) *nchunks = 0;
    if (level < 0 || level > 3) return 0;                    /* zzflate.cpp:201-202 */
    uint8_t h[10];
    int hl = header_bytes(format, h);
    if ((uint64_t)hl <= cap) memcpy(dest, h, (size_t)hl);
    if (chunk_sizes && calls < max_chunks) chunk_sizes[calls] = (uint64_t)hl;  /* :205 */
    calls++;
    enc_t* e = (enc_t*)malloc(sizeof(enc_t));
    enc_init(e, level, dest + hl, cap > (uint64_t)hl ? cap - (uint64_t)hl : 0, 0, src, src + n);
    e->bs.chunk_sizes = chunk_sizes ? chunk_sizes + calls : NULL;
    e->bs.max_chunks = max_chunks - calls;
    add_data(e, src, src + n, 1);
    bs_flush(&e->bs);
    bs_finish_chunks(&e->bs);
    calls += e->bs.chunks_reported;                          /* :207-215 one callback per chunk */
    uint64_t count = (uint64_t)hl + e->bs.pos;
    enc_free(e); free(e);
    bits_t t;
    bs_init(&t, count <= cap ? dest + count : dest, count <= cap ? cap - count : 0, 1);
    append_checksum(&t, format, src, n);                     /* :217-221 */
    bs_flush(&t);
    if (chunk_sizes && calls < max_chunks) chunk_sizes[calls] = t.pos;
    calls++;
    if (nchunks) *nchunks = calls;
    return count + t.pos;
}

/* Warm window -- NOT in the reference (the product's zz_ctx_set_warm_window). The reference's threaded
 * mode starts every range with a cold table (zzflate.cpp:101-125); its single Encoder carries the table across blocks
 * (FixHashTable, encoder.cpp:320-327). The warm window sits between the two: before a packet (level >= 1) is parsed, every
 * position of the last `warm` bytes in front of it (as far as the stream has them) is entered into the table under the
 * hash CalcHash uses for that position (the three bytes behind it at level 1, its own three at level >= 2), ascending, so
 * the highest position per hash stays. This is the executable definition the product's warm mode is tested against. */
static void prehash(enc_t* e, const uint8_t* s, uint64_t before, uint64_t warm)
{
    int64_t W = (int64_t)(before < warm ? before : warm);
    /* the key of a position: level 1 hashes the three bytes BEHIND it (encoder.cpp:344), level >= 2 its own three
     * (encoder.cpp:388) */
    const int k = e->level == 1 ? 1 : 0;
    for (int64_t q = -W; q < 0; ++q) e->table[calc_hash(e, s + q + k)] = q;
}

/* zzflate.cpp:101-125: one packet with a fresh encoder (cold table; warm > 0: see prehash) */
uint64_t zzo_packet_warm(int level, const uint8_t* base, uint64_t off, uint64_t len, int is_final,
                         uint8_t* out, uint64_t cap, uint64_t warm)
{
    enc_t* e = (enc_t*)malloc(sizeof(enc_t));
    const uint8_t* s = base + off;
    const uint8_t* end = s + len;
    if (is_final) {
        enc_init(e, level, out, cap, 1, base, end);
        if (warm && level >= 1 && level <= 3) prehash(e, s, off, warm);
        add_data(e, s, end, 1);                              /* :110-113 */
    } else {
        enc_init(e, level, out, cap, 1, base, end - (len ? 1 : 0));
        if (len) {
            if (warm && level >= 1 && level <= 3) prehash(e, s, off, warm);
            add_data(e, s, end - 1, 0);                      /* :116 */
            e->level = 0;                                    /* :119 SetLevel(0) */
            e->gend = end;
            add_data(e, end - 1, end, 0);                    /* :120 one stored byte = byte alignment */
        }
    }
    bs_flush(&e->bs);                                        /* :122 */
    uint64_t r = e->bs.pos;
    if (e->bs.overflow) r = ZZO_ERROR;
    enc_free(e); free(e);
    return r;
}
uint64_t zzo_packet(int level, const uint8_t* base, uint64_t off, uint64_t len, int is_final,
                    uint8_t* out, uint64_t cap)
{
    return zzo_packet_warm(level, base, off, len, is_final, out, cap, 0);
}

uint64_t zzo_encode_packets_warm(uint8_t* dest, uint64_t cap, const uint8_t* src, uint64_t n, int format,
                                 int level, uint64_t packet_size, uint64_t warm)
{
    uint8_t h[10];
    int hl = header_bytes(format, h);
    if (level < 0 || level > 6 || cap < (uint64_t)hl || packet_size == 0) return ZZO_ERROR;   /* 4..6: the extended levels */
    memcpy(dest, h, (size_t)hl);
    uint64_t count = (uint64_t)hl;
    uint64_t npk = (n + packet_size - 1) / packet_size;       /* fixed-size ranges replace :67-78,:97-99 */
    for (uint64_t k = 0; k < npk; ++k) {
        uint64_t off = k * packet_size;
        uint64_t len = n - off < packet_size ? n - off : packet_size;
        uint64_t w = zzo_packet_warm(level, src, off, len, k == npk - 1, dest + count, cap - count, warm);
        if (w == ZZO_ERROR) return ZZO_ERROR;
        count += w;                                          /* :134-155 in-order join, contiguous */
    }
    bits_t t;
    bs_init(&t, dest + count, cap - count, 1);
    append_checksum(&t, format, src, n);
    bs_flush(&t);
    if (t.overflow) return ZZO_ERROR;
    return count + t.pos;
}
uint64_t zzo_encode_packets(uint8_t* dest, uint64_t cap, const uint8_t* src, uint64_t n, int format,
                            int level, uint64_t packet_size)
{
    return zzo_encode_packets_warm(dest, cap, src, n, format, level, packet_size, 0);
}

/* The reference's OWN threaded=true split (zzflate.cpp:67-78 divideInRanges, :81-156 WriteDeflateStream): `count` ranges of
 * step = ceil(n / count) bytes -- count = std::thread::hardware_concurrency() there, a parameter here --, every range the
 * packet recipe above (fresh encoder; non-final ranges end with the one stored byte), joined in order; an input shorter
 * than 100 * count bytes goes through the single encoder (:84). A range's encoder sees destLen / count bytes of room (:98);
 * with a roomy destination that changes nothing at levels 0, 2, 3. (Level 1's threaded stream is invalid in the reference,
 * a fixed block's length follows from the room, and the non-final ranges' blocks are never closed.) */
uint64_t zzo_encode_ranges(uint8_t* dest, uint64_t cap, const uint8_t* src, uint64_t n, int format, int level, uint32_t count)
{
    if (count == 0) return ZZO_ERROR;
    if (n < 100ull * count) return zzo_encode(dest, cap, src, n, format, level);     /* :84 */
    uint8_t h[10];
    int hl = header_bytes(format, h);
    if (level < 0 || level > 3 || cap < (uint64_t)hl) return ZZO_ERROR;
    memcpy(dest, h, (size_t)hl);
    uint64_t total = (uint64_t)hl;
    const uint64_t step = (n + count - 1) / count;            /* :70 */
    /* D10: with count near sqrt(n) or above, divideInRanges' trailing boundaries lie past n (the reference then reads past its
     * input: undefined, nothing to restate) -- refused, as zz_encode_ranges_device refuses it */
    if ((uint64_t)(count - 1) * step >= n) return ZZO_ERROR;
    for (uint32_t k = 0; k < count; ++k) {
        const uint64_t off = step * k;                        /* :73 */
        const uint64_t end = k + 1 == count ? n : step * (k + 1);   /* :75 */
        uint64_t w = zzo_packet(level, src, off, end - off, k + 1 == count, dest + total, cap - total);   /* :101-125 */
        if (w == ZZO_ERROR) return ZZO_ERROR;
        total += w;                                           /* :134-155 */
    }
    bits_t t;
    bs_init(&t, dest + total, cap - total, 1);
    append_checksum(&t, format, src, n);
    bs_flush(&t);
    if (t.overflow) return ZZO_ERROR;
    return total + t.pos;
}

/* outputbitstream.h:83-124 driven as zztest/TestBitOutput.cpp does */
uint64_t zzo_bitstream(const uint64_t* bits, const int* counts, int n, uint8_t* out, uint64_t cap,
                       int* before_flush)
{
    bits_t b;
    memset(out, 0, (size_t)cap);
    bs_init(&b, out, cap, 1);
    for (int i = 0; i < n; ++i) bs_put(&b, bits[i], counts[i]);
    if (before_flush) *before_flush = out[0];
    bs_flush(&b);
    return b.pos;
}

/* ---- timing driver for bench.py's cpu_baseline leg when oracle/_ref is absent (kind "port"): the twin of zzref_bench in
 * ref_harness.cpp -- native threads, static partition (item j is slice j mod nslices and belongs to thread j mod threads),
 * every thread first-touches its own output buffer. mode 0: zzo_encode per slice; mode 1: zzo_packet over P-byte ranges. */
typedef struct {
    int mode, t, threads, format, level; uint32_t P;
    const uint8_t* base; uint64_t nslices, slice_bytes, nitems; uint64_t* produced; double* secs;
    pthread_barrier_t* start; double* w0;
} bench_arg_t;
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
static void* bench_worker(void* vp)
{
    bench_arg_t* a = (bench_arg_t*)vp;
    const uint64_t cap = 2 * a->slice_bytes + 4096;
    uint8_t* out = (uint8_t*)malloc(cap);
    memset(out, 1, cap);                                      /* allocated and touched before the clock starts */
    pthread_barrier_wait(a->start);
    const double t0 = now_s();
    if (a->t == 0) *a->w0 = t0;
    for (uint64_t j = (uint64_t)a->t; j < a->nitems; j += (uint64_t)a->threads) {
        const uint64_t sl = j % a->nslices, at = sl * a->slice_bytes;
        uint64_t r = 0;
        if (a->mode == 0) r = zzo_encode(out, cap, a->base + at, a->slice_bytes, a->format, a->level);
        else
            for (uint64_t off = at; off < at + a->slice_bytes; off += a->P) {
                const uint64_t ln = at + a->slice_bytes - off < a->P ? at + a->slice_bytes - off : a->P;
                r += zzo_packet(a->level, a->base, off, ln, (sl == a->nslices - 1 && off + ln == at + a->slice_bytes) ? 1 : 0, out, cap);
            }
        a->produced[sl] = r;
    }
    a->secs[a->t] = now_s() - t0;
    free(out);
    return NULL;
}
double zzo_bench(int mode, const uint8_t* base, uint64_t nslices, uint64_t slice_bytes, uint64_t nitems, int threads,
                 int format, int level, uint32_t P, uint64_t* produced, double* thread_secs)
{
    if (threads < 1) threads = 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    bench_arg_t* args = (bench_arg_t*)malloc(sizeof(bench_arg_t) * (size_t)threads);
    double w0 = 0;
    pthread_barrier_t start;
    pthread_barrier_init(&start, NULL, (unsigned)threads);
    for (int t = 0; t < threads; ++t) {
        args[t] = (bench_arg_t){ mode, t, threads, format, level, P, base, nslices, slice_bytes, nitems, produced, thread_secs, &start, &w0 };
        if (t) pthread_create(&th[t], NULL, bench_worker, &args[t]);
    }
    bench_worker(&args[0]);
    for (int t = 1; t < threads; ++t) pthread_join(th[t], NULL);
    const double w = now_s() - w0;
    pthread_barrier_destroy(&start);
    free(th); free(args);
    return w;
}
