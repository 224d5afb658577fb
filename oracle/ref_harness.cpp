// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A thin extern "C" driver around the *unmodified* reference (jandevaan/zzflate), compiled from the
// sources where they lie under /root/reference by oracle/Makefile into oracle/_ref/libzzref.so.
// Nothing in here restates the reference's algorithm: it only calls the reference's public entry
// points (zzflate.h:17-19, encoder.h:79-97, huffman.h:38-81, outputbitstream.h, crc.h, adler.cpp)
// the way the reference's own callers do (zzflate.cpp:101-125 for packets; zztest/*.cpp for KATs).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the resulting .so.
//
// Harness rules for packets come from SURVEY.md section 8c (isolation + guard bytes for level 1,
// real preceding bytes for level >= 2, one guard byte before packet 0).

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>
#include <memory>

#include "zzflate.h"
#include "encoder.h"
#include "crc.h"

extern "C" {

// ---- whole-stream entry points (zzflate.cpp:225-242, :197-222) -------------------------------

// The level-1 pass reads up to 7 bytes past the input end and does not clamp the match length
// (SURVEY.md App. B D1), so the input is copied in front of seeded guard bytes: a stream that inflates
// to the input is then independent of what followed it in memory.
static std::vector<uint8_t> guarded_copy(const uint8_t* src, uint64_t n, uint32_t guard_seed)
{
    std::vector<uint8_t> v(n + 64);
    if (n) memcpy(v.data(), src, n);
    uint32_t s = guard_seed * 2654435761u + 777u;
    for (uint64_t i = 0; i < 64; ++i) {
        s = s * 1664525u + 1013904223u;
        v[n + i] = (uint8_t)(s >> 24);
    }
    return v;
}

// returns bytes written, or ~0 on the reference's error convention
uint64_t zzref_encode(uint8_t* dest, uint64_t cap, const uint8_t* src_in, uint64_t n, int format,
                      int level, int threaded, uint32_t guard_seed)
{
    auto copy = guarded_copy(src_in, n, guard_seed);
    const uint8_t* src = copy.data();
    Config cfg;
    cfg.format = (Format)format;
    cfg.level = (uint8_t)level;
    cfg.threaded = threaded != 0;
    size_t len = cap;
    ZzFlateEncode(dest, &len, src, n, &cfg);
    return len;
}

// callback API; concatenates all chunks into dest (cap must be ample). returns total bytes, and the
// number of callbacks in *ncalls.
uint64_t zzref_encode_callback(uint8_t* dest, uint64_t cap, const uint8_t* src_in, uint64_t n, int format,
                               int level, int threaded, int* ncalls, uint32_t guard_seed)
{
    auto copy = guarded_copy(src_in, n, guard_seed);
    const uint8_t* src = copy.data();
    Config cfg;
    cfg.format = (Format)format;
    cfg.level = (uint8_t)level;
    cfg.threaded = threaded != 0;
    uint64_t total = 0;
    int calls = 0;
    ZzFlateEncodeToCallback(src, n, &cfg, [&](const uint8_t* p, size_t c) -> bool {
        if (total + c <= cap) memcpy(dest + total, p, c);
        total += c;
        calls++;
        return false;
    });
    if (ncalls) *ncalls = calls;
    return total;
}

// ---- the parity unit: one packet, driven exactly as the lambda at zzflate.cpp:101-125 ----------
//
//   base      : start of the whole input buffer
//   off, len  : the packet is base[off .. off+len)
//   is_final  : last packet of the stream (-> AddData(s,e,true)); otherwise
//               AddData(s,e-1,false); SetLevel(0); AddData(e-1,e,false)
//   guard_seed: seeds the guard bytes placed after (and, for packet 0, before) the isolated copy
//   out, cap  : destination (cap should be >= 2*len + 1024)
// returns bytes written.
uint64_t zzref_packet(int level, const uint8_t* base, uint64_t off, uint64_t len, int is_final,
                      uint8_t* out, uint64_t cap, uint32_t guard_seed)
{
    const uint64_t GUARD = 64;
    // how many real preceding bytes to keep in front of the isolated copy (level >= 2 backward
    // extension reads them, encoder.cpp:404 region / :522 in this checkout)
    uint64_t pre = 0;
    if (level >= 2) pre = off < 40000 ? off : 40000;
    std::vector<uint8_t> scratch(GUARD + pre + len + GUARD);
    uint32_t s = guard_seed * 2654435761u + 12345u;
    for (uint64_t i = 0; i < GUARD; ++i) {
        s = s * 1664525u + 1013904223u;
        scratch[i] = (uint8_t)(s >> 24);
    }
    memcpy(&scratch[GUARD], base + off - pre, pre + len);
    uint64_t body = is_final ? len : (len ? len - 1 : 0);
    // guard bytes directly after the bytes handed to the compressing AddData
    uint8_t* pkt = &scratch[GUARD + pre];
    uint8_t last = len ? base[off + len - 1] : 0;
    for (uint64_t i = 0; i < GUARD; ++i) {
        s = s * 1664525u + 1013904223u;
        scratch[GUARD + pre + body + i] = (uint8_t)(s >> 24);
    }

    auto enc = std::make_unique<Encoder>(level, out, (int64_t)cap);
    if (is_final) {
        enc->AddData(pkt, pkt + len, true);
    } else if (len) {
        enc->AddData(pkt, pkt + len - 1, false);
        enc->SetLevel(0);
        enc->AddData(&last, &last + 1, false);
    }
    enc->stream.Flush();
    return enc->stream.byteswritten();
}

// un-guarded, zero-copy variant for timing the reference as the CPU baseline (bench.py): the caller
// guarantees >= 8 readable bytes after src[n).
uint64_t zzref_encode_inplace(uint8_t* dest, uint64_t cap, const uint8_t* src, uint64_t n, int format,
                              int level, int threaded)
{
    Config cfg;
    cfg.format = (Format)format;
    cfg.level = (uint8_t)level;
    cfg.threaded = threaded != 0;
    size_t len = cap;
    ZzFlateEncode(dest, &len, src, n, &cfg);
    return len;
}

// ---- timing driver for bench.py's cpu_baseline leg -----------------------------------------------------------
// Runs the reference over `nitems` work items on `threads` native threads and returns the wall time in seconds. Item j is
// slice (j mod nslices) of base (slices of slice_bytes bytes) and belongs to thread j mod threads: a static partition,
// no lock, no interpreter in the loop. Every thread allocates (and so first-touches) its own output buffer.
//   mode 0: one ZzFlateEncode(threaded=false) call per slice -- the reference as its own callers run it (zztest/Test.cpp:161);
//   mode 1: the packet recipe (zzflate.cpp:101-125) over the slice's P-byte ranges -- the GPU's work and the GPU's bytes;
//           only the very last packet of the last slice is final.
// produced[s] = output bytes of slice s (as last computed), thread_secs[t] = busy time of thread t.
double zzref_bench(int mode, const uint8_t* base, uint64_t nslices, uint64_t slice_bytes, uint64_t nitems, int threads,
                   int format, int level, uint32_t P, uint64_t* produced, double* thread_secs)
{
    using clk = std::chrono::steady_clock;
    if (threads < 1) threads = 1;
    const uint64_t cap = 2 * slice_bytes + 4096;
    std::atomic<int> ready(0), go(0);
    clk::time_point w0;
    auto work = [&](int t) {
        std::vector<uint8_t> out(cap, 1);                    // allocated and touched before the clock starts
        ready.fetch_add(1);
        if (t == 0) { while (ready.load() < threads) std::this_thread::yield(); w0 = clk::now(); go.store(1); }
        else while (!go.load()) std::this_thread::yield();
        const auto t0 = clk::now();
        for (uint64_t j = (uint64_t)t; j < nitems; j += (uint64_t)threads) {
            const uint64_t sl = j % nslices, at = sl * slice_bytes;
            uint64_t r = 0;
            if (mode == 0) r = zzref_encode_inplace(out.data(), cap, base + at, slice_bytes, format, level, 0);
            else
                for (uint64_t off = at; off < at + slice_bytes; off += P) {
                    const uint64_t ln = at + slice_bytes - off < P ? at + slice_bytes - off : P;
                    r += zzref_packet(level, base, off, ln, (sl == nslices - 1 && off + ln == at + slice_bytes) ? 1 : 0, out.data(), cap, 1);
                }
            produced[sl] = r;
        }
        thread_secs[t] = std::chrono::duration<double>(clk::now() - t0).count();
    };
    std::vector<std::thread> th;
    for (int t = 1; t < threads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    return std::chrono::duration<double>(clk::now() - w0).count();
}

// ---- checksums (adler.cpp:5-43, crc.cpp:24-33) ---------------------------------------------------
// what the reference's threaded=true path divides the input by on THIS machine (zzflate.cpp:97): goldens made from
// zzref_encode(threaded=1) record it (tests/golden/make_ranges.py)
uint32_t zzref_hardware_concurrency() { return std::thread::hardware_concurrency(); }
uint32_t zzref_adler32x(uint32_t start, const uint8_t* p, uint64_t n) { return adler32x(start, p, n); }
uint32_t zzref_combine(uint32_t a, uint32_t b, uint64_t lenb) { return combine(a, b, lenb); }
uint32_t zzref_crc32(const uint8_t* p, uint64_t n, uint32_t start) { return crc32(p, n, start); }

// ---- Huffman (huffman.h:38-81, huffman.cpp) -------------------------------------------------------
void zzref_calc_lengths(const int* freqs, int n, int maxlen, int* out)
{
    std::vector<int> f(freqs, freqs + n), l;
    CalcLengths(f, l, maxlen);
    for (int i = 0; i < n; ++i) out[i] = l[i];
}

void zzref_generate(const int* lengths, int n, int* out_len, uint32_t* out_bits)
{
    std::vector<int> l(lengths, lengths + n);
    std::vector<code> c(n);
    for (auto& x : c) { x.length = 0; x.bits = 0; }
    huffman::generate<code>(l, &c[0]);
    for (int i = 0; i < n; ++i) { out_len[i] = c[i].length; out_bits[i] = c[i].bits; }
}

uint32_t zzref_reverse(uint32_t v, int len) { return huffman::reverse(v, len); }

void zzref_default_table_lengths(int* out288)
{
    auto l = huffman::defaultTableLengths();
    for (int i = 0; i < 288; ++i) out288[i] = l[i];
}

// code-length RLE; freqs19 is accumulated into (as the reference does); returns record count
int zzref_from_lengths(const int* lengths, int n, int* freqs19, uint8_t* out_value, uint8_t* out_payload)
{
    std::vector<int> l(lengths, lengths + n);
    std::vector<int> f(freqs19, freqs19 + 19);
    auto recs = FromLengths(l, f);
    for (int i = 0; i < 19; ++i) freqs19[i] = f[i];
    for (size_t i = 0; i < recs.size(); ++i) { out_value[i] = recs[i].value; out_payload[i] = recs[i].payLoad; }
    return (int)recs.size();
}

// ---- distance buckets (encoder.h:94-95; zztest/TestHuffman.cpp:10-32) ------------------------------
int zzref_find_distance(int d) { return Encoder::FindDistance(d); }
int zzref_read_lut(int d) { return Encoder::ReadLut(d); }

// ---- bit packer (outputbitstream.h:83-124; zztest/TestBitOutput.cpp) -------------------------------
// appends n (bits,count) pairs, flushes, returns bytes written; *before_flush = out[0] before Flush
uint64_t zzref_bitstream(const uint64_t* bits, const int* counts, int n, uint8_t* out, uint64_t cap,
                         int* before_flush)
{
    memset(out, 0, cap);
    outputbitstream strm(out, cap);
    for (int i = 0; i < n; ++i) strm.AppendToBitStream(bits[i], counts[i]);
    if (before_flush) *before_flush = out[0];
    strm.Flush();
    return strm.byteswritten();
}

// merged length codes (encoder.cpp:121-133 / CreateMergedLengthCodes is public static)
void zzref_merged_length_codes(const int* sym_len, const uint32_t* sym_bits, int* out_len, uint32_t* out_bits)
{
    code syms[286];
    for (int i = 0; i < 286; ++i) { syms[i].length = sym_len[i]; syms[i].bits = sym_bits[i]; }
    code l[259];
    Encoder::CreateMergedLengthCodes(l, syms);
    for (int i = 0; i < 259; ++i) { out_len[i] = l[i].length; out_bits[i] = l[i].bits; }
}

int zzref_sizeof_config() { return (int)sizeof(Config); }

}  // extern "C"
