// A caller written against the reference's public header only (zzflate.h:8-19), as zztest/Test.cpp:202-246
// is: fixed-buffer API at level 1, callback API at level 2, system zlib as the decoder. Prints "OK <bytes>"
// per call, "ERR" when the library reports the reference's error convention (*destLen == ~0).
#include <cstdio>
#include <cstring>
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
    // adler.cpp API (zztest/Test.cpp:301-313)
    unsigned char v[9] = { 0, 1, 23, 30, 4, 69, 145, 32, 216 };
    printf("%s combine\n", combine(adler32x(1, v, 5), adler32x(0, v + 5, 4), 4) == adler32x(1, v, 9) ? "OK" : "BAD");
    printf("%s crc\n", crc32((const uint8_t*)"123456789", 9) == 0xCBF43926u ? "OK" : "BAD");
    return bad;
}
