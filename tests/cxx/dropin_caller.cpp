// A caller written against the reference's public header only (zzflate.h:8-19): packet mode first (threaded = true),
// then the reference's own callers as zztest/Test.cpp:202-282 writes them (threaded = false, tight destination at
// level 1, callback API otherwise, gzip into a destination of input size), system zlib as the decoder. Prints
// "OK ..." per call, "ERR" when the library reports the reference's error convention (*destLen == ~0).
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <vector>
#include <fstream>
#include <zlib.h>
#include "zzflate.h"

static bool inflates_to(const std::vector<uint8_t>& comp, const std::vector<uint8_t>& want)
{
    std::vector<uint8_t> out(want.size() + 16);
    uLongf n = out.size();
    int rc = uncompress(out.data(), &n, comp.data(), comp.size());
    return rc == Z_OK && n == want.size() && memcmp(out.data(), want.data(), n) == 0;
}

int main(int argc, char** argv)
{
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> in((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    int bad = 0;
    {
        Config cfg = { Zlib, 1, true };
        std::vector<uint8_t> out(in.size() * 2 + 1024);
        size_t len = out.size();
        ZzFlateEncode(out.data(), &len, in.data(), in.size(), &cfg);
        if (len == ~(size_t)0) { printf("ERR encode\n"); bad++; }
        else { out.resize(len); printf("%s %zu\n", inflates_to(out, in) ? "OK" : "BAD", len); }
    }
    {
        Config cfg = { Zlib, 2, true };
        std::vector<uint8_t> out;
        int calls = 0;
        ZzFlateEncodeToCallback(in.data(), in.size(), &cfg, [&](const uint8_t* p, size_t n) -> bool {
            out.insert(out.end(), p, p + n); calls++; return false; });
        if (calls == 0) { printf("ERR callback\n"); bad++; }
        else printf("%s %zu %d\n", inflates_to(out, in) ? "OK" : "BAD", out.size(), calls);
    }
    // The reference's own callers, as written (zztest/Test.cpp:202-246 testroundtrip, :249-282 testroundtripgzip):
    // Config{Zlib, level} leaves `threaded` false; level 1 goes through ZzFlateEncode with a destination of
    // max(200, input size) bytes -- too small for one fixed-Huffman block, so the encoder cuts several (encoder.cpp:
    // 331-337) -- and every other level through ZzFlateEncodeToCallback.
    for (int level = 1; level <= 3; ++level) {
        Config cfg = { Zlib, (uint8_t)level };
        std::vector<uint8_t> compressed;
        if (cfg.level == 1) {
            compressed.resize(std::max((size_t)200, in.size()));
            size_t comp_len = compressed.size();
            ZzFlateEncode(&compressed[0], &comp_len, &in[0], in.size(), &cfg);
            if (comp_len == ~(size_t)0) { printf("ERR roundtrip %d\n", level); bad++; continue; }
            compressed.resize(comp_len);
        } else {
            ZzFlateEncodeToCallback(&in[0], in.size(), &cfg, [&compressed](const uint8_t* buffer, size_t count) -> bool {
                compressed.insert(compressed.end(), buffer, buffer + count);
                return false;
            });
            if (compressed.empty()) { printf("ERR roundtrip %d\n", level); bad++; continue; }
        }
        const bool ok = inflates_to(compressed, in);
        if (!ok) bad++;
        printf("%s roundtrip %d %zu\n", ok ? "OK" : "BAD", level, compressed.size());
    }
    {
        std::vector<uint8_t> compressed(in.size());                      // Test.cpp:254: dest = input size
        size_t comp_len = compressed.size();
        Config cfg = { Gzip, 1, false };
        ZzFlateEncode(&compressed[0], &comp_len, &in[0], in.size(), &cfg);
        if (comp_len == ~(size_t)0) { printf("ERR gzip\n"); bad++; }
        else {
            std::vector<uint8_t> out(in.size() + 16);
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            inflateInit2(&zs, 15 | 16);                                  // Test.cpp:87-140
            zs.next_in = compressed.data(); zs.avail_in = (uInt)comp_len;
            zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
            const int rc = inflate(&zs, Z_FINISH);
            const bool ok = rc == Z_STREAM_END && zs.total_out == in.size() && memcmp(out.data(), in.data(), in.size()) == 0;
            inflateEnd(&zs);
            if (!ok) bad++;
            printf("%s gzip %zu\n", ok ? "OK" : "BAD", comp_len);
        }
    }
    // adler.cpp API (zztest/Test.cpp:301-313)
    unsigned char v[9] = { 0, 1, 23, 30, 4, 69, 145, 32, 216 };
    printf("%s combine\n", combine(adler32x(1, v, 5), adler32x(0, v + 5, 4), 4) == adler32x(1, v, 9) ? "OK" : "BAD");
    printf("%s crc\n", crc32((const uint8_t*)"123456789", 9) == 0xCBF43926u ? "OK" : "BAD");
    return bad;
}
