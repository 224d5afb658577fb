// include/zzflate.h -- source-level drop-in for the reference's public header (zzflate/zzflate.h:1-22).
//
// Same enum, same 8-byte Config, same two entry points with C++ linkage, so that code written against
// jandevaan/zzflate (only zztest/Test.cpp:161,212,217,258 in the reference tree) recompiles and links
// against libzzflate_amd.so unchanged; the Itanium-mangled names are identical to the reference's
// (_Z13ZzFlateEncodePhPmPKhmPK6Config, _Z23ZzFlateEncodeToCallbackPKhmPK6ConfigSt8functionIFbS0_mEE).
// The implementations (zzflate_amd/csrc/zz_cxx_shim.cpp) forward to the C ABI in zzflate_amd.h.
#ifndef _ZZFLATE
#define _ZZFLATE

#include <stdint.h>
#include <stddef.h>
#include <functional>

enum Format { Zlib, Gzip, Deflate };

struct Config
{
	Format format;
	uint8_t level;
	bool threaded;
};

// *destLen: in = capacity of dest, out = bytes written, or ~0 on error
void ZzFlateEncode(uint8_t* dest, size_t* destLen, const uint8_t* source, size_t sourceLen, const Config* config);

// callback(chunk, bytes) is called in order: header, stream chunks (<= 1,000,000 bytes each), trailer
void ZzFlateEncodeToCallback(const uint8_t* source, size_t sourceLen, const Config* config,
                             std::function<bool(const uint8_t*, size_t)> callback);

// checksum helpers the reference exports from adler.cpp / crc.cpp (encoder.h:10-12, crc.h:7)
uint32_t adler32x(uint32_t startValue, const uint8_t* data, size_t len);
uint32_t combine(uint32_t first, uint32_t second, size_t lenSecond);
uint32_t crc32(const uint8_t* buffer, size_t length, uint32_t startValue = 0);

#endif
