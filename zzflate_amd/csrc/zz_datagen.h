// zz_datagen.h -- counter-based synthetic inputs for BASELINE.json's configs.
//
// Block b (64 KiB) of a stream is a pure function of (kind, seed, b): the device fills HBM without PCIe
// traffic (one thread per block) and the host can regenerate any slice for parity sampling. Integer
// arithmetic only, so host and device agree bit for bit.
//   TEXT   "enwik-style": Zipf-like words over a 65,536-word seeded vocabulary, punctuation, newlines,
//          some <tag>/[[link]] markup                                    (config 2)
//   RANDOM splitmix64 bytes, incompressible                              (config 3)
//   LOG    timestamped log lines from 4,096 seeded templates             (config 5)
//   MIX    16 MiB segments cycling through twelve families: prose, XML, C source, HTML, binary records, bi-level image
//          rows, 16-bit gradients, database dump, executable-like, random, DNA-like, log text            (config 4)
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#define ZZ_GEN_BLOCK 65536u

namespace zzgen {

__host__ __device__ inline uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct rng { uint64_t s; };
__host__ __device__ inline uint64_t next(rng& r) { r.s += 0x9E3779B97F4A7C15ull; return mix64(r.s); }

struct sink { uint8_t* p; uint32_t n, cap; };
__host__ __device__ inline void put(sink& o, uint8_t c) { if (o.n < o.cap) o.p[o.n] = c; o.n++; }

__host__ __device__ inline uint8_t letter(uint32_t h)
{
    // 64 letters, roughly English frequencies
    const char* L = "eeeeeeeetttttaaaaaoooooiiiiinnnnssssshhhhrrrrdddlllccuummwwffggyp";
    return (uint8_t)L[h & 63];
}
// Zipf-like rank in [0, 65535]: octave = min of two uniform draws (about 45 % of the words come from the 16 most
// frequent, 75 % from the first 256, 94 % from the first 4096 -- roughly English), uniform inside the octave
__host__ __device__ inline uint32_t zipf_rank(uint64_t u)
{
    uint32_t k1 = (uint32_t)(u & 15), k2 = (uint32_t)((u >> 4) & 15);
    uint32_t k = k1 < k2 ? k1 : k2;
    uint32_t r = (uint32_t)(u >> 8);
    return ((1u << k) | (r & ((1u << k) - 1))) - 1;
}
__host__ __device__ inline void put_word(sink& o, uint64_t seed, uint32_t rank, bool cap)
{
    uint64_t h = mix64(seed * 0x100000001B3ull + rank);
    uint32_t oct = 32 - (uint32_t)__builtin_clz(rank + 1);        // 1..16
    uint32_t len = 1 + oct / 3 + (uint32_t)(h & 3) + ((h >> 2) & 1);   // frequent words are short
    h >>= 3;
    for (uint32_t i = 0; i < len; ++i) {
        if ((i & 7) == 7) h = mix64(h + i);
        uint8_t c = letter((uint32_t)h);
        h >>= 6;
        if (cap && i == 0) c = (uint8_t)(c - 32);
        put(o, c);
    }
    // a third of the rarer words end in a common suffix, as inflected English words do
    if (oct > 5) {
        const uint64_t hs = mix64(h ^ rank);
        if ((hs & 3) == 0) {
            const char* suf[8] = { "ing", "ed", "ly", "tion", "s", "er", "es", "ment" };
            const char* x = suf[(hs >> 2) & 7];
            for (int i = 0; x[i]; ++i) put(o, (uint8_t)x[i]);
        }
    }
}
__host__ __device__ inline void put_dec(sink& o, uint32_t v, int width)
{
    char tmp[10];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v && n < 10);
    for (int i = n; i < width; ++i) put(o, '0');
    while (n) put(o, (uint8_t)tmp[--n]);
}

__host__ __device__ inline void gen_text(sink& o, uint64_t seed, uint64_t block)
{
    rng r{ mix64(seed ^ (block * 0xD1342543DE82EF95ull)) };
    uint32_t col = 0, wrap = 60 + (uint32_t)(next(r) % 40);
    bool cap = true;
    while (o.n < o.cap) {
        uint64_t u = next(r);
        uint32_t start = o.n;
        uint32_t sel = (uint32_t)(u >> 40) & 63;
        if (sel == 0) {            // <tag> word </tag>
            put(o, '<'); put_word(o, seed, zipf_rank(u) & 255, false); put(o, '>');
            put_word(o, seed, zipf_rank(next(r)), false);
            put(o, '<'); put(o, '/'); put_word(o, seed, zipf_rank(u) & 255, false); put(o, '>');
        } else if (sel == 1) {     // [[link]]
            put(o, '['); put(o, '['); put_word(o, seed, zipf_rank(u), true); put(o, ']'); put(o, ']');
        } else if (sel == 2) {     // a number
            put_dec(o, (uint32_t)(u >> 20) % 3000, 1);
        } else {
            put_word(o, seed, zipf_rank(u), cap);
        }
        cap = false;
        uint32_t q = (uint32_t)(u >> 48) & 31;
        if (q == 0) { put(o, '.'); cap = true; }
        else if (q < 3) put(o, ',');
        col += o.n - start + 1;
        if (col >= wrap) { put(o, '\n'); col = 0; wrap = 60 + (uint32_t)(next(r) % 40); }
        else put(o, ' ');
    }
}

__host__ __device__ inline void gen_random(sink& o, uint64_t seed, uint64_t block)
{
    uint64_t base = block * (ZZ_GEN_BLOCK / 8);
    for (uint32_t i = 0; i < o.cap; i += 8) {
        uint64_t v = mix64(seed + (base + i / 8) * 0x9E3779B97F4A7C15ull);
        for (uint32_t j = 0; j < 8 && i + j < o.cap; ++j) o.p[i + j] = (uint8_t)(v >> (8 * j));
    }
    o.n = o.cap;
}

__host__ __device__ inline void gen_log(sink& o, uint64_t seed, uint64_t block)
{
    rng r{ mix64(seed ^ (block * 0xA0761D6478BD642Full)) };
    // monotone timestamps: ~600 lines per block, 1..50 ms apart
    uint64_t ms = block * 20000ull;
    const char* levels[4] = { "INFO", "INFO", "WARN", "ERROR" };
    while (o.n < o.cap) {
        uint64_t u = next(r);
        ms += 1 + (u & 31);
        uint32_t sec = (uint32_t)(ms / 1000), day = sec / 86400;
        uint32_t sod = sec % 86400;
        put_dec(o, 2026, 4); put(o, '-'); put_dec(o, 1 + (day / 28) % 12, 2); put(o, '-'); put_dec(o, 1 + day % 28, 2);
        put(o, 'T'); put_dec(o, sod / 3600, 2); put(o, ':'); put_dec(o, (sod / 60) % 60, 2); put(o, ':');
        put_dec(o, sod % 60, 2); put(o, '.'); put_dec(o, (uint32_t)(ms % 1000), 3); put(o, 'Z'); put(o, ' ');
        const char* hs = "host-";
        for (int i = 0; hs[i]; ++i) put(o, (uint8_t)hs[i]);
        put_dec(o, (uint32_t)(u >> 8) & 63, 2); put(o, ' ');
        uint32_t tmpl = zipf_rank(u >> 16) & 4095;
        put_word(o, seed ^ 0x5EC, tmpl & 31, false);            // service name
        put(o, '['); put_dec(o, 1000 + ((tmpl * 7919u) % 30000), 1); put(o, ']'); put(o, ':'); put(o, ' ');
        const char* lv = levels[(u >> 30) & 3];
        for (int i = 0; lv[i]; ++i) put(o, (uint8_t)lv[i]);
        put(o, ' ');
        uint64_t th = mix64(seed + tmpl);
        uint32_t nw = 3 + (uint32_t)(th & 7);
        for (uint32_t i = 0; i < nw; ++i) {                      // the template's fixed words
            put_word(o, seed, (uint32_t)(mix64(th + i) & 2047), false);
            put(o, ' ');
        }
        uint32_t nkv = 1 + (uint32_t)((th >> 8) & 3);
        for (uint32_t i = 0; i < nkv; ++i) {                     // k=v with varying values
            put_word(o, seed, (uint32_t)(mix64(th + 100 + i) & 255), false);
            put(o, '=');
            put_dec(o, (uint32_t)(next(r) >> 40) % 100000, 1);
            put(o, i + 1 < nkv ? ' ' : '\n');
        }
    }
}

__host__ __device__ inline void gen_records(sink& o, uint64_t seed, uint64_t block)   // kennedy.xls-like
{
    rng r{ mix64(seed ^ (block * 0x8EBC6AF09C88C6E3ull)) };
    uint32_t row = (uint32_t)(block * 1000);
    while (o.n < o.cap) {
        uint64_t u = next(r);
        put(o, 0x7E); put(o, 0x02); put(o, 0x0A); put(o, 0x00);
        put(o, (uint8_t)row); put(o, (uint8_t)(row >> 8)); put(o, (uint8_t)(u & 15)); put(o, 0);
        put(o, 0x0F); put(o, 0x00);
        uint32_t v = (uint32_t)(u >> 16) % 5000;
        put(o, (uint8_t)v); put(o, (uint8_t)(v >> 8)); put(o, 0); put(o, 0x40);
        if ((u & 15) == 15) row++;
    }
}
__host__ __device__ inline void gen_dna(sink& o, uint64_t seed, uint64_t block)
{
    rng r{ mix64(seed ^ (block * 0x589965CC75374CC3ull)) };
    const char* b = "ACGT";
    while (o.n < o.cap) {
        uint64_t u = next(r);
        for (int i = 0; i < 32; ++i) { put(o, (uint8_t)b[u & 3]); u >>= 2; }
    }
}
__host__ __device__ inline void gen_gradient(sink& o, uint64_t seed, uint64_t block)   // 16-bit noisy ramp
{
    rng r{ mix64(seed ^ (block * 0x1D8E4E27C47D124Full)) };
    uint32_t v = (uint32_t)(block * 37) & 0xFFFF;
    while (o.n < o.cap) {
        uint64_t u = next(r);
        for (int i = 0; i < 16; ++i) {
            v = (v + 3 + (uint32_t)(u & 3)) & 0xFFFF;
            u >>= 4;
            put(o, (uint8_t)v); put(o, (uint8_t)(v >> 8));
        }
    }
}

__host__ __device__ inline void put_str(sink& o, const char* x) { for (int i = 0; x[i]; ++i) put(o, (uint8_t)x[i]); }
__host__ __device__ inline void put_indent(sink& o, uint32_t depth) { for (uint32_t i = 0; i < depth; ++i) { put(o, ' '); put(o, ' '); } }

// XML (Silesia "xml"-like): nested elements from a small tag vocabulary, attributes, short text nodes
__host__ __device__ inline void gen_xml(sink& o, uint64_t seed, uint64_t block)
{
    rng r{ mix64(seed ^ (block * 0xC2B2AE3D27D4EB4Full)) };
    uint32_t stack[8], depth = 0;
    put_str(o, "<?xml version=\"1.0\" encoding=\"UTF-8\"?>\n");
    while (o.n < o.cap) {
        uint64_t u = next(r);
        const uint32_t act = (uint32_t)(u & 7);
        if (depth < 7 && (act < 4 || depth == 0)) {
            const uint32_t tag = zipf_rank(u >> 8) & 63;
            put_indent(o, depth); put(o, '<'); put_word(o, seed ^ 0x3C, tag, false);
            if (u & 0x100) { put_str(o, " id=\""); put_dec(o, (uint32_t)(u >> 20) % 100000, 1); put(o, '"'); }
            if (u & 0x200) { put(o, ' '); put_word(o, seed ^ 0x3D, (uint32_t)(u >> 44) & 15, false); put_str(o, "=\""); put_word(o, seed, zipf_rank(u >> 24), false); put(o, '"'); }
            put(o, '>'); put(o, '\n');
            stack[depth++] = tag;
        } else if (act < 6 && depth > 0) {
            put_indent(o, depth);
            const uint32_t nw = 1 + (uint32_t)((u >> 8) & 7);
            for (uint32_t i = 0; i < nw; ++i) { put_word(o, seed, zipf_rank(next(r)), false); put(o, i + 1 < nw ? ' ' : '\n'); }
        } else if (depth > 0) {
            --depth;
            put_indent(o, depth); put(o, '<'); put(o, '/'); put_word(o, seed ^ 0x3C, stack[depth], false); put(o, '>'); put(o, '\n');
        }
    }
}
// C source (Canterbury fields.c / Silesia "samba"-like): declarations, calls, control flow, comments
__host__ __device__ inline void gen_csrc(sink& o, uint64_t seed, uint64_t block)
{
    rng r{ mix64(seed ^ (block * 0x27D4EB2F165667C5ull)) };
    const char* types[8] = { "int", "char *", "unsigned long", "static int", "struct ", "void", "size_t", "const char *" };
    uint32_t depth = 0;
    while (o.n < o.cap) {
        uint64_t u = next(r);
        const uint32_t k = (uint32_t)(u & 15);
        put_indent(o, 2 * depth);
        if (k == 0 && depth == 0) {
            put_str(o, types[(u >> 4) & 7]); put(o, ' '); put_word(o, seed ^ 0xC1, zipf_rank(u >> 8) & 1023, false); put(o, '(');
            put_str(o, types[(u >> 20) & 7]); put(o, ' '); put_word(o, seed ^ 0xC2, (uint32_t)(u >> 24) & 63, false); put_str(o, ")\n{\n");
            depth = 1;
        } else if (k < 3 && depth > 0 && depth < 5) {
            put_str(o, (u & 16) ? "if (" : "while ("); put_word(o, seed ^ 0xC2, (uint32_t)(u >> 8) & 63, false);
            put_str(o, (u & 32) ? " != NULL) {\n" : " < "); if (!(u & 32)) { put_dec(o, (uint32_t)(u >> 30) & 255, 1); put_str(o, ") {\n"); }
            depth++;
        } else if (k < 5 && depth > 0) {
            depth--; put_indent(o, 2 * depth); put_str(o, "}\n"); if (depth == 0) put(o, '\n');
        } else if (k < 7) {
            put_str(o, "/* "); const uint32_t nw = 2 + (uint32_t)((u >> 8) & 7);
            for (uint32_t i = 0; i < nw; ++i) { put_word(o, seed, zipf_rank(next(r)), false); put(o, ' '); }
            put_str(o, "*/\n");
        } else if (k < 11) {
            put_word(o, seed ^ 0xC2, (uint32_t)(u >> 8) & 63, false); put_str(o, " = "); put_word(o, seed ^ 0xC1, zipf_rank(u >> 16) & 1023, false);
            put(o, '('); put_word(o, seed ^ 0xC2, (uint32_t)(u >> 40) & 63, false); put_str(o, ", "); put_dec(o, (uint32_t)(u >> 48) & 1023, 1); put_str(o, ");\n");
        } else if (k < 13) {
            put_str(o, types[(u >> 4) & 7]); put(o, ' '); put_word(o, seed ^ 0xC2, (uint32_t)(u >> 8) & 63, false); put_str(o, " = 0;\n");
        } else {
            put_str(o, "return "); put_word(o, seed ^ 0xC2, (uint32_t)(u >> 8) & 63, false); put_str(o, ";\n");
        }
    }
}
// HTML (Canterbury cp.html-like): tags with attributes around prose, links, lists
__host__ __device__ inline void gen_html(sink& o, uint64_t seed, uint64_t block)
{
    rng r{ mix64(seed ^ (block * 0x165667B19E3779F9ull)) };
    const char* tags[8] = { "p", "li", "h2", "td", "div", "span", "b", "em" };
    put_str(o, "<html><head><title>"); put_word(o, seed, zipf_rank(next(r)), true); put_str(o, "</title></head>\n<body>\n");
    while (o.n < o.cap) {
        uint64_t u = next(r);
        const char* t = tags[u & 7];
        put(o, '<'); put_str(o, t);
        if (u & 8) { put_str(o, " class=\""); put_word(o, seed ^ 0x48, (uint32_t)(u >> 8) & 31, false); put(o, '"'); }
        put(o, '>');
        const uint32_t nw = 3 + (uint32_t)((u >> 16) & 15);
        for (uint32_t i = 0; i < nw; ++i) {
            uint64_t v = next(r);
            if ((v & 31) == 0) {
                put_str(o, "<a href=\"http://www."); put_word(o, seed ^ 0x49, zipf_rank(v >> 8) & 255, false); put_str(o, ".org/");
                put_word(o, seed, zipf_rank(v >> 24), false); put_str(o, ".html\">"); put_word(o, seed, zipf_rank(v >> 40), false); put_str(o, "</a>");
            } else put_word(o, seed, zipf_rank(v), i == 0);
            put(o, i + 1 < nw ? ' ' : '.');
        }
        put(o, '<'); put(o, '/'); put_str(o, t); put(o, '>'); put(o, '\n');
    }
}
// database dump (Silesia "nci"/"osdb"-like): INSERT statements with ids, short strings, numbers, dates
__host__ __device__ inline void gen_dbdump(sink& o, uint64_t seed, uint64_t block)
{
    rng r{ mix64(seed ^ (block * 0x85EBCA77C2B2AE63ull)) };
    uint32_t id = (uint32_t)(block * 700);
    const uint32_t table = (uint32_t)(block >> 4) & 7;
    while (o.n < o.cap) {
        uint64_t u = next(r);
        put_str(o, "INSERT INTO "); put_word(o, seed ^ 0xDB, table, false); put_str(o, " VALUES (");
        put_dec(o, id++, 1); put_str(o, ", '"); put_word(o, seed, zipf_rank(u), true); put(o, ' '); put_word(o, seed, zipf_rank(u >> 20), true);
        put_str(o, "', "); put_dec(o, (uint32_t)(u >> 40) % 100000, 1); put(o, '.'); put_dec(o, (uint32_t)(u >> 8) % 100, 2);
        put_str(o, ", '20"); put_dec(o, 10 + (uint32_t)(u >> 50) % 17, 2); put(o, '-'); put_dec(o, 1 + (uint32_t)(u >> 54) % 12, 2); put(o, '-');
        put_dec(o, 1 + (uint32_t)(u >> 58) % 28, 2); put_str(o, "', "); put_str(o, (u & 1) ? "NULL" : "'Y'"); put_str(o, ");\n");
    }
}
// executable-like (Silesia "mozilla"/"ooffice"-like): x86-flavoured opcode mix, little-endian displacements that cluster,
// runs of padding, and an occasional string table
__host__ __device__ inline void gen_exe(sink& o, uint64_t seed, uint64_t block)
{
    rng r{ mix64(seed ^ (block * 0x9E3779B185EBCA87ull)) };
    const uint8_t ops[16] = { 0x8B, 0x89, 0xE8, 0x83, 0x55, 0x5D, 0xC3, 0x74, 0x75, 0xFF, 0x8D, 0x50, 0x6A, 0x85, 0x33, 0xEB };
    uint32_t fn_base = (uint32_t)(block * 0x10000);
    while (o.n < o.cap) {
        uint64_t u = next(r);
        const uint32_t k = (uint32_t)(u & 63);
        if (k == 0) {                                       // padding between functions
            const uint32_t n = 1 + (uint32_t)((u >> 8) & 15);
            for (uint32_t i = 0; i < n; ++i) put(o, 0xCC);
        } else if (k == 1) {                                // a string constant
            put_word(o, seed, zipf_rank(u >> 8), false); put(o, 0);
        } else {
            const uint8_t op = ops[(u >> 6) & 15];
            put(o, op);
            if (op == 0xE8) {                               // call rel32: targets cluster around a few functions
                const uint32_t t = fn_base + ((zipf_rank(u >> 12) & 255) << 6) - o.n;
                put(o, (uint8_t)t); put(o, (uint8_t)(t >> 8)); put(o, (uint8_t)(t >> 16)); put(o, (uint8_t)(t >> 24));
            } else if (op == 0x8B || op == 0x89 || op == 0x8D) {   // mov/lea r, [ebp+disp8]
                put(o, (uint8_t)(0x45 | ((u >> 12) & 0x38))); put(o, (uint8_t)(0xFC - 4 * ((u >> 20) & 7)));
            } else if (op == 0x83) { put(o, (uint8_t)(0xC0 | ((u >> 12) & 0x3F))); put(o, (uint8_t)((u >> 20) & 0x1F)); }
            else if (op == 0x74 || op == 0x75 || op == 0xEB || op == 0x6A) put(o, (uint8_t)((u >> 12) & 0x3F));
            else if (op == 0xFF) { put(o, 0x15); const uint32_t a = 0x00401000u + ((zipf_rank(u >> 12) & 127) << 2);
                put(o, (uint8_t)a); put(o, (uint8_t)(a >> 8)); put(o, (uint8_t)(a >> 16)); put(o, (uint8_t)(a >> 24)); }
            else if (op == 0x85 || op == 0x33) put(o, (uint8_t)(0xC0 | ((u >> 12) & 0x3F)));
        }
    }
}
// bi-level image rows (Canterbury ptt5-like): 216-byte scan lines, mostly white, black runs that repeat from line to line
__host__ __device__ inline void gen_bilevel(sink& o, uint64_t seed, uint64_t block)
{
    rng r{ mix64(seed ^ (block * 0xFF51AFD7ED558CCDull)) };
    uint8_t line[216];
    for (int i = 0; i < 216; ++i) line[i] = 0;
    while (o.n < o.cap) {
        uint64_t u = next(r);
        const uint32_t edits = (uint32_t)(u & 3);           // a few runs change per line, the rest is the line above
        for (uint32_t e = 0; e < edits; ++e) {
            uint64_t v = next(r);
            const uint32_t at = (uint32_t)(v % 216), len = 1 + (uint32_t)((v >> 8) & 15);
            const uint8_t fill = (v & 0x10000) ? 0xFF : 0x00;
            for (uint32_t i = 0; i < len && at + i < 216; ++i) line[at + i] = (i == 0 && fill) ? (uint8_t)(0xFF >> ((v >> 20) & 7)) : fill;
        }
        if ((u & 0xFF00) == 0) for (int i = 0; i < 216; ++i) line[i] = 0;    // a blank band now and then
        for (int i = 0; i < 216; ++i) put(o, line[i]);
    }
}

// MIX (BASELINE configs[3]): 16 MiB segments (256 blocks) cycling through twelve generators modelled on
// the members of the Silesia and Canterbury corpora
#define ZZ_GEN_MIX_FAMILIES 12
__host__ __device__ inline void gen_block(int kind, uint64_t seed, uint64_t block, uint8_t* dst, uint32_t cap)
{
    sink o{ dst, 0, cap };
    switch (kind) {
    case 0: gen_text(o, seed, block); break;
    case 1: gen_random(o, seed, block); break;
    case 2: gen_log(o, seed, block); break;
    default: {
        switch ((block >> 8) % ZZ_GEN_MIX_FAMILIES) {
        case 0: gen_text(o, seed, block); break;        // prose
        case 1: gen_xml(o, seed, block); break;         // XML
        case 2: gen_csrc(o, seed, block); break;        // C source
        case 3: gen_html(o, seed, block); break;        // HTML
        case 4: gen_records(o, seed, block); break;     // structured binary records (kennedy.xls)
        case 5: gen_bilevel(o, seed, block); break;     // bi-level image rows (ptt5)
        case 6: gen_gradient(o, seed, block); break;    // 16-bit noisy gradients
        case 7: gen_dbdump(o, seed, block); break;      // database dump
        case 8: gen_exe(o, seed, block); break;         // executable-like opcode mix
        case 9: gen_random(o, seed, block); break;      // already compressed
        case 10: gen_dna(o, seed, block); break;        // DNA-like, four symbols
        default: gen_log(o, seed, block); break;        // log text
        }
    }
    }
}

}  // namespace zzgen
