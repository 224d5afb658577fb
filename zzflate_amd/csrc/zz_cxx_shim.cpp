// zz_cxx_shim.cpp -- the reference's C++ entry points (zzflate.h:17,19; adler.cpp; crc.h:7) with their
// original mangled names, forwarding to the C ABI. Kept in its own translation unit so that the C ABI
// does not depend on <functional>.
#include "../../include/zzflate.h"
#include "../../include/zzflate_amd.h"

static_assert(sizeof(Config) == sizeof(zz_config), "Config layout must match zzflate.h:10-15");

static zz_config to_c(const Config* c)
{
    zz_config z;
    z.format = (int32_t)c->format;
    z.level = c->level;
    z.threaded = c->threaded ? 1 : 0;
    return z;
}

void ZzFlateEncode(uint8_t* dest, size_t* destLen, const uint8_t* source, size_t sourceLen, const Config* config)
{
    zz_config z = to_c(config);
    uint64_t len = *destLen;
    zz_encode(dest, &len, source, sourceLen, &z);   // on any failure len == ~0 (zzflate.cpp:232)
    *destLen = (size_t)len;
}

static int trampoline(void* user, const uint8_t* chunk, uint64_t bytes)
{
    auto* fn = static_cast<std::function<bool(const uint8_t*, size_t)>*>(user);
    (*fn)(chunk, (size_t)bytes);   // result ignored, as in zzflate.cpp:205-221
    return 0;
}

void ZzFlateEncodeToCallback(const uint8_t* source, size_t sourceLen, const Config* config,
                             std::function<bool(const uint8_t*, size_t)> callback)
{
    zz_config z = to_c(config);
    zz_encode_callback(source, sourceLen, &z, trampoline, &callback);
}

uint32_t adler32x(uint32_t startValue, const uint8_t* data, size_t len) { return zz_adler32(startValue, data, len); }
uint32_t combine(uint32_t first, uint32_t second, size_t lenSecond) { return zz_adler32_combine(first, second, lenSecond); }
uint32_t crc32(const uint8_t* buffer, size_t length, uint32_t startValue) { return zz_crc32(buffer, length, startValue); }
