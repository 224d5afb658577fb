// zz_api.hip -- host side of libzzflate_amd.so: contexts, workspaces, kernel launches and the C ABI
// declared in include/zzflate_amd.h. Mirrors the L4 layer of the reference (zzflate.cpp): container
// header, range split, fan-out, in-order join, trailer -- with the fan-out being "packets -> wavefronts".
//
// There is no CPU encode path in this library: every byte of DEFLATE output is produced by the HIP
// kernels, and every entry point fails with ZZ_E_HIP when no device is usable.
#include <hip/hip_runtime.h>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/zzflate_amd.h"
#include "zz_common.h"
#include "zz_wave.h"
#include "zz_checksum.h"
#include "zz_emit.h"
#include "zz_level0.h"
#include "zz_level1.h"
#include "zz_level1p.h"
#include "zz_level2.h"
#include "zz_stream2.h"
#include "zz_compact.h"
#include "zz_datagen.h"
#include "zz_verify.h"

using namespace zz;

static thread_local std::string g_err;
static void set_err(const std::string& s) { g_err = s; }
extern "C" const char* zz_last_error(void) { return g_err.c_str(); }
extern "C" const char* zz_version(void) { return "zzflate_amd 0.1 (gfx950)"; }
// Compile-time switches this binary was built with, as a space-separated list; "" for the product build. The ZZ_*_X_* ones and the probe write
// WRONG STREAMS (timing experiments: tools/pipe_probe.sh, tools/units_probe.sh); ZZ_PROF adds cycle stamps; the last one is the
// test build libzzflate_amd_careful.so. tests/test_abi.py holds the shipped library to "".
extern "C" const char* zz_build_flags(void)
{
    return ""
#ifdef ZZ_L1_PIPE_PROBE
        " ZZ_L1_PIPE_PROBE"
#endif
#ifdef ZZ_L1P_X_NOCROSS
        " ZZ_L1P_X_NOCROSS"
#endif
#ifdef ZZ_L1P_X_NOEMIT
        " ZZ_L1P_X_NOEMIT"
#endif
#ifdef ZZ_L1P_X_NOEXT
        " ZZ_L1P_X_NOEXT"
#endif
#ifdef ZZ_L2P_X_NOEXT
        " ZZ_L2P_X_NOEXT"
#endif
#if ZZ_L2P_FLAGS
        " ZZ_L2P_FLAGS"
#endif
#if !ZZ_L2P_XCHG
        " ZZ_L2P_XCHG=0"
#endif
#ifdef ZZ_PROF
        " ZZ_PROF"
#endif
#ifdef ZZ_L1_EMIT_PRIO
        " ZZ_L1_EMIT_PRIO"
#endif
#ifdef ZZ_ST_ALWAYS_CAREFUL
        " ZZ_ST_ALWAYS_CAREFUL"
#endif
        ;
}

#define HIPCHK(expr)                                                                     \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            set_err(std::string(#expr) + ": " + hipGetErrorString(_e));                  \
            return ZZ_E_HIP;                                                             \
        }                                                                                \
    } while (0)

struct zz_ctx {
    int device = 0;
    // workspace (grown on demand, kept across calls so steady-state calls do not allocate)
    uint8_t* slots = nullptr;   uint64_t slots_cap = 0;
    uint32_t* sizes = nullptr;  uint64_t* offsets = nullptr;  zz_cks* cks = nullptr;  uint64_t npk_cap = 0;
    uint8_t* l2_scratch = nullptr; uint64_t l2_scratch_cap = 0;
    zz_result* d_res = nullptr; zz_cks_total* d_cks_total = nullptr; uint32_t* d_err = nullptr;
    zz_result* h_res = nullptr;          // pinned
    unsigned long long* d_prof = nullptr; // 16 counters for diagnostic (-DZZ_PROF) builds
    uint8_t* d_tail = nullptr;            // 128 bytes: the end of the shard being encoded, then zeros (k_encode_l1p's over-reads)
    // staging for the host-buffer entry points
    uint8_t* stage_in = nullptr;  uint64_t stage_in_cap = 0;
    uint8_t* stage_out = nullptr; uint64_t stage_out_cap = 0;
    // slab pipeline of the host-buffer entry points: pinned slabs in and out, device output slabs, three streams
    uint8_t* pin_in[2] = { nullptr, nullptr };  uint64_t pin_in_cap = 0;
    uint8_t* pin_out[2] = { nullptr, nullptr }; uint64_t pin_out_cap = 0;
    uint8_t* slab_out[2] = { nullptr, nullptr }; uint64_t slab_out_cap = 0;
    uint8_t* slab_in[2] = { nullptr, nullptr };                                    // device input ring: halo + slab each
    hipStream_t s_in = nullptr, s_enc = nullptr, s_out = nullptr;
    hipEvent_t ev_in[2] = { nullptr, nullptr }, ev_out[2] = { nullptr, nullptr };
    // what the last packet-mode call did, for zz_verify_last_device
    zz_verify_params last = {};  bool have_last = false;
    unsigned long long* d_verify = nullptr;
    uint32_t* d_work = nullptr;          // level 2: packet counter of the persistent workgroups
    uint64_t* d_log = nullptr; uint64_t log_cap_bytes = 0;   // sequential stream, callback form: EnsureOutputLength log
    // a call that has been enqueued but not waited for (zz_encode_device_async .. zz_encode_finish)
    struct {
        bool active = false; hipStream_t st = nullptr; uint32_t npk = 0; int level = 0; bool whole = false;
        // the call itself, for the rerun behind a run-time LDS-order violation (encode_finish)
        bool order_checked = false;          // the launch relied on the LDS's lane order and checked it in the kernel
        const uint8_t* d_src = nullptr; uint64_t n = 0, halo = 0; bool last_is_final = false; uint8_t* d_dst = nullptr; uint64_t cap = 0;
        int format = 0, cks_kind = 0, level_asked = 0; uint32_t P = 0; uint32_t warm = 0;
    } pend;
    uint32_t* h_err = nullptr;           // pinned: the kernels' sticky error word
    uint32_t warm = 0;                   // levels >= 1: warm window in bytes (0 = cold packets, the reference's threaded mode)
    bool extended = false;               // levels 4..6 accepted (beyond the reference)
    bool timing = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool have_time = false;
};

// Small host values (an empty input's block, a result record whose size is known up front) reach the device as kernel
// ARGUMENTS -- copied at launch -- not as asynchronous copies from stack locals, which would still be read after an
// enqueue-only call (zz_encode_device_async) has returned.
// the last min(n, 64) bytes of a shard, then zeros: what k_encode_l1p reads instead of bytes past the shard's end
__global__ void k_fill_tail(const uint8_t* src, uint64_t n, uint8_t* tail)
{
    const uint64_t tn = n < 64 ? n : 64;
    const uint32_t i = threadIdx.x;
    tail[i] = i < tn ? src[n - tn + i] : (uint8_t)0;
}
__global__ void k_put_small(uint8_t* dst, uint64_t bytes, uint32_t nbytes, zz_result* res, uint64_t stream_bytes)
{
    for (uint32_t i = 0; i < nbytes; ++i) dst[i] = (uint8_t)(bytes >> (8 * i));
    if (res) res->stream_bytes = stream_bytes;
}

static int header_len(int format) { return format == ZZ_ZLIB ? 2 : format == ZZ_GZIP ? 10 : 0; }
static int trailer_len(int format) { return format == ZZ_ZLIB ? 4 : format == ZZ_GZIP ? 8 : 0; }

// worst-case bytes one packet can occupy in its slot, per level (multiple of 16)
static uint32_t slot_stride_for(int level, uint32_t P)
{
    uint64_t b;
    if (level == 0) b = (uint64_t)P + 10;
    else if (level == 1) b = ((uint64_t)9 * P + 10 + 7) / 8 + 6;   // 3 + 9 bits/byte + EOB, + 1-byte stored block
    else b = (uint64_t)P + 11;                                      // stored fallback is the worst case
    b += 8;                                                         // the ring stores whole words
    return (uint32_t)((b + 15) & ~15ull);
}

extern "C" uint64_t zz_bound(uint64_t n, int format, int level, uint32_t P)
{
    if (P == 0 || P > ZZ_MAX_PACKET_SIZE) P = ZZ_DEFAULT_PACKET;
    uint64_t npk = n ? (n + P - 1) / P : 1;
    uint64_t per;
    if (level == 0) per = (uint64_t)P + 10;
    else if (level == 1) per = ((uint64_t)9 * P + 10 + 7) / 8 + 6;
    else per = (uint64_t)P + 11;
    return header_len(format) + npk * per + trailer_len(format) + 16;
}

static bool lds_order_ok(int device);
extern "C" void zz_ctx_destroy(zz_ctx* c);
extern "C" int zz_ctx_create(int device, zz_ctx** out)
{
    if (!out) { set_err("null out"); return ZZ_E_ARG; }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        set_err(std::string("no HIP device available (hipGetDeviceCount: ") + hipGetErrorString(e) + ", count " +
                std::to_string(count) + "): this library has no CPU encode path");
        return ZZ_E_HIP;
    }
    if (device < 0 || device >= count) { set_err("bad device index"); return ZZ_E_ARG; }
    HIPCHK(hipSetDevice(device));
    zz_ctx* c = new zz_ctx();
    c->device = device;
    const int rc = [&]() -> int {
        HIPCHK(hipMalloc(&c->d_res, sizeof(zz_result)));
        HIPCHK(hipMalloc(&c->d_cks_total, sizeof(zz_cks_total)));
        HIPCHK(hipMalloc(&c->d_err, 4 * sizeof(uint32_t)));          // [0] slot overflow, [1] stream truncated, [2] log entries
        HIPCHK(hipMalloc(&c->d_work, 16 * sizeof(uint32_t)));
        HIPCHK(hipMalloc(&c->d_prof, 64 * sizeof(unsigned long long)));
        HIPCHK(hipMalloc(&c->d_tail, 128));
        HIPCHK(hipMemset(c->d_prof, 0, 64 * sizeof(unsigned long long)));
        HIPCHK(hipHostMalloc((void**)&c->h_res, sizeof(zz_result), hipHostMallocDefault));
        HIPCHK(hipHostMalloc((void**)&c->h_err, 4 * sizeof(uint32_t), hipHostMallocDefault));
        HIPCHK(hipEventCreate(&c->ev0));
        HIPCHK(hipEventCreate(&c->ev1));
        (void)lds_order_ok(device);          // the probe, once per device, HERE: the launch sites only read its cached verdict
        return ZZ_OK;
    }();
    if (rc) { zz_ctx_destroy(c); return rc; }                            // nothing allocated so far is left behind
    *out = c;
    return ZZ_OK;
}

extern "C" void zz_ctx_destroy(zz_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->pend.active) { (void)hipStreamSynchronize(c->pend.st); c->pend.active = false; }   // an enqueued call still uses the buffers
    (void)hipFree(c->slots); (void)hipFree(c->sizes); (void)hipFree(c->offsets); (void)hipFree(c->cks);
    (void)hipFree(c->l2_scratch);
    (void)hipFree(c->d_res); (void)hipFree(c->d_cks_total); (void)hipFree(c->d_err); (void)hipFree(c->d_prof); (void)hipFree(c->d_tail);
    (void)hipFree(c->stage_in); (void)hipFree(c->stage_out); (void)hipFree(c->d_verify); (void)hipFree(c->d_work); (void)hipFree(c->d_log);
    for (int i = 0; i < 2; ++i) {
        (void)hipHostFree(c->pin_in[i]); (void)hipHostFree(c->pin_out[i]); (void)hipFree(c->slab_out[i]); (void)hipFree(c->slab_in[i]);
        if (c->ev_in[i]) (void)hipEventDestroy(c->ev_in[i]);
        if (c->ev_out[i]) (void)hipEventDestroy(c->ev_out[i]);
    }
    if (c->s_in) (void)hipStreamDestroy(c->s_in);
    if (c->s_enc) (void)hipStreamDestroy(c->s_enc);
    if (c->s_out) (void)hipStreamDestroy(c->s_out);
    (void)hipHostFree(c->h_res); (void)hipHostFree(c->h_err);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    delete c;
}

extern "C" uint64_t zz_ctx_workspace_bytes(const zz_ctx* c)
{
    if (!c) return 0;
    return c->slots_cap + c->npk_cap * (4 + 8 + sizeof(zz_cks)) + c->l2_scratch_cap + c->stage_in_cap + c->stage_out_cap;
}
// diagnostic (not part of the public header): workgroups of the level's encode kernel the runtime places on one CU
extern "C" int zz_debug_occupancy(int level)
{
    int nb = -1;
    if (level == 1) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_encode_l1p, ZZ_L1P_THREADS, 0);
    else if (level == -1) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_encode_l1, ZZ_L1_THREADS, 0);
    else if (level >= 4) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_encode_l2_t<32768u, true>, ZZ_L2_THREADS, 0);
    else if (level >= 2) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_encode_l2_t<0u, false, true>, ZZ_L2P_THREADS, 0);
    else if (level == -2) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_encode_l2_t<0u, false>, ZZ_L2_THREADS, 0);
    else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_encode_l0, 256, 0);
    return nb;
}
// diagnostic (not part of the public header): does the LDS serve the lanes of one wavefront instruction that add to one
// address in ascending lane order, and one wavefront's instructions in issue order? k_l6_matches' counting sort takes its places
// from exactly that (zz_level6.h). Runs `trials` pairs of 64-key blocks per wavefront in 16-wavefront workgroups on every CU, keys
// drawn with many duplicates (shared dwords, shared banks, one heavy key, all equal ...); *bad = the number of returned counts
// that differ from the lane-ascending, block-ascending ones (0 on gfx950).
__global__ __launch_bounds__(1024) void k_lds_order_probe(uint32_t seed, uint32_t trials, unsigned long long* bad)
{
    __shared__ uint32_t T[16][256];                // 512 packed 16-bit counters per wavefront
    __shared__ uint16_t K[16][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = lane; i < 256; i += 64) T[wave][i] = 0;
    uint32_t x = seed ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u);
    unsigned long long nbad = 0;
    for (uint32_t t = 0; t < trials; ++t) {
        uint32_t k[2];
        for (int u = 0; u < 2; ++u) {
            x = x * 1664525u + 1013904223u;
            const uint32_t r = x >> 8;
            switch ((t + blockIdx.x) & 7u) {
            case 0: k[u] = r % 512u; break;                          // few duplicates
            case 1: k[u] = r % 16u; break;                           // many
            case 2: k[u] = r % 3u; break;                            // three values, two of them in one dword
            case 3: k[u] = 7u; break;                                // all the same
            case 4: k[u] = (r % 8u) * 64u; break;                    // one bank, different addresses
            case 5: k[u] = (r % 4u) * 64u + ((r >> 5) & 1u); break;  // one bank, halves of a dword
            case 6: k[u] = (lane & 1) ? 5u : r % 512u; break;        // one heavy key among light ones
            default: k[u] = r % 40u; break;
            }
            K[wave][u * 64 + lane] = (uint16_t)k[u];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        // two blocks back to back, no wait between them
        const uint32_t oa = atomicAdd(&T[wave][k[0] >> 1], 1u << ((k[0] & 1u) << 4));
        const uint32_t ob = atomicAdd(&T[wave][k[1] >> 1], 1u << ((k[1] & 1u) << 4));
        const uint32_t ra = (oa >> ((k[0] & 1u) << 4)) & 0xFFFFu, rb = (ob >> ((k[1] & 1u) << 4)) & 0xFFFFu;
        uint32_t wa = 0, wb = 0;                                     // what lane order gives
        for (int i = 0; i < 64; ++i) {
            if (i < lane && K[wave][i] == k[0]) ++wa;
            if (K[wave][i] == k[1]) ++wb;                            // all of block A come first ...
            if (i < lane && K[wave][64 + i] == k[1]) ++wb;           // ... then the lower lanes of block B
        }
        if (ra != wa) ++nbad;
        if (rb != wb) ++nbad;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        atomicSub(&T[wave][k[0] >> 1], 1u << ((k[0] & 1u) << 4));
        atomicSub(&T[wave][k[1] >> 1], 1u << ((k[1] & 1u) << 4));
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        // the same for the masked exchange the level-1 and level-2 parsers insert with (zz_level1.h): every lane puts lane + 1 into
        // its key's 16-bit half and must get back what the nearest lower lane with that key put there (0: none); afterwards the
        // half holds the highest lane's; and a non-returning one under a lane mask leaves the highest masked lane's
        {
            const uint32_t sh = (k[0] & 1u) << 4;
            const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&T[wave][k[0] >> 1];
            uint32_t old;
            asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(addr), "v"(0xFFFFu << sh), "v"((uint32_t)(lane + 1) << sh) : "memory");
            uint32_t wprev = 0, whigh = 0, wmask = 0;
            const uint64_t msk = ((uint64_t)x << 32 | (x * 2246822519u)) | (t & 1u ? 0ull : ~0ull >> (x & 63u));   // some lanes, or most
            for (int i = 0; i < 64; ++i)
                if (K[wave][i] == k[0]) {
                    if (i < lane) wprev = (uint32_t)i + 1;
                    whigh = (uint32_t)i + 1;
                    if ((msk >> i) & 1) wmask = (uint32_t)i + 65;
                }
            if (((old >> sh) & 0xFFFFu) != wprev) ++nbad;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (((T[wave][k[0] >> 1] >> sh) & 0xFFFFu) != whigh) ++nbad;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            ((uint16_t*)T[wave])[k[0]] = 0;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            const uint64_t mskw = ((uint64_t)readlane((uint32_t)(msk >> 32), 0) << 32) | readlane((uint32_t)msk, 0);   // one mask for the wavefront
            uint32_t wm2 = 0;
            for (int i = 0; i < 64; ++i) if (K[wave][i] == k[0] && ((mskw >> i) & 1)) wm2 = (uint32_t)i + 65;
            (void)wmask;
            if ((mskw >> lane) & 1)
                asm volatile("ds_mskor_b32 %0, %1, %2" :: "v"(addr), "v"(0xFFFFu << sh), "v"((uint32_t)(lane + 65) << sh) : "memory");
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (((T[wave][k[0] >> 1] >> sh) & 0xFFFFu) != wm2) ++nbad;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            ((uint16_t*)T[wave])[k[0]] = 0;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            // plain 16-bit stores: two blocks back to back; the slot must end up with the LAST store's highest lane (the warm
            // window's pre-hash enters its positions with nothing else: zz_level1.h warm_prehash)
            ((uint16_t*)T[wave])[k[0]] = (uint16_t)(lane + 1);
            ((uint16_t*)T[wave])[k[1]] = (uint16_t)(lane + 65);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            uint32_t ws = 0;
            for (int i = 0; i < 64; ++i) if (K[wave][i] == k[0]) ws = (uint32_t)i + 1;
            for (int i = 0; i < 64; ++i) if (K[wave][64 + i] == k[0]) ws = (uint32_t)i + 65;
            if (((uint16_t*)T[wave])[k[0]] != ws) ++nbad;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            ((uint16_t*)T[wave])[k[0]] = 0;
            ((uint16_t*)T[wave])[k[1]] = 0;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}
extern "C" int zz_debug_lds_atomic_order(zz_ctx* c, uint32_t trials, unsigned long long* bad, unsigned long long* checked)
{
    if (!c || !bad) return ZZ_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    unsigned long long* d = nullptr;
    HIPCHK(hipMalloc(&d, sizeof(unsigned long long)));
    HIPCHK(hipMemset(d, 0, sizeof(unsigned long long)));
    hipLaunchKernelGGL(k_lds_order_probe, dim3(512), dim3(1024), 0, 0, 0x5EEDu, trials, d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(bad, d, sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIPCHK(hipFree(d));
    if (checked) *checked = 512ull * 1024ull * 6ull * trials;
    return ZZ_OK;
}
// ---- the run-time guard for that property ---------------------------------------------------------------------------------
// Three things in this library are only correct where the LDS serves equal addresses of one instruction in ascending lane order
// and one wavefront's instructions in issue order -- which gfx950 does and its ISA manual does not promise: the warm window's
// pre-hash (zz_level1.h warm_prehash), the extended levels' counting sort (zz_level6.h) and the two-wavefront level-1 kernel
// (zz_level1p.h: a slot ends up with the HIGHEST lane's store). So the probe above runs once per device, the first time one of
// them is asked for, and its verdict is kept: where it fails, zz_ctx_set_warm_window(> 0) and zz_ctx_set_extended_levels(1) are
// refused with ZZ_E_UNSUPPORTED and level 1 runs its one-wavefront kernel (k_encode_l1, which asks the LDS for nothing of the
// kind); levels 0..3 with cold packets never depend on it. zz_debug_force_lds_order lets a test take the refusal path.
static std::mutex g_order_mu;
static std::atomic<int> g_order_verdict[64];     // per device: 0 unknown, 1 holds, -1 does not
static std::atomic<int> g_order_forced{-1};      // -1: probe; 0 / 1: the verdict every device gets (tests)
static std::atomic<int> g_force_violation{0};    // tests: the next level-1 / warm-window launches report a violated order (bit 4 of the error word)
extern "C" void zz_debug_force_lds_order(int verdict) { g_order_forced.store(verdict < 0 ? -1 : (verdict ? 1 : 0)); }
// (not part of the public header) n > 0: the next n encode launches that check the order in the kernel (k_encode_l1p, the warm window's
// pre-hash) behave as if they had seen it violated -- the test of the rerun / refusal path behind that check
extern "C" void zz_debug_force_lds_violation(int launches) { g_force_violation.store(launches < 0 ? 0 : launches); }
// (not part of the public header) forget a device's verdict (tests put back what a forced violation flipped)
extern "C" void zz_debug_reset_lds_order(int device) { if (device >= 0 && device < 64) g_order_verdict[device].store(0); }
// Runs the probe where the device has no verdict yet. Blocking (allocation, a kernel on the null stream, a synchronous copy): called
// from zz_ctx_create and the setters, never from an enqueue-only entry point -- the launch sites read the cached verdict
// (lds_order_cached: one atomic load, no HIP call, nothing that would break a stream capture or sit between two timing events).
static bool lds_order_ok(int device)
{
    const int forced = g_order_forced.load();
    if (forced >= 0) return forced == 1;
    if (device < 0 || device >= 64) return false;
    if (g_order_verdict[device].load() == 0) {
        std::lock_guard<std::mutex> lk(g_order_mu);
        if (g_order_verdict[device].load() == 0) {
            unsigned long long bad = 1, *d = nullptr;
            int prev = 0;
            bool ran = hipGetDevice(&prev) == hipSuccess && hipSetDevice(device) == hipSuccess && hipMalloc(&d, sizeof(bad)) == hipSuccess;
            if (ran) {
                ran = hipMemset(d, 0, sizeof(bad)) == hipSuccess;
                if (ran) {
                    hipLaunchKernelGGL(k_lds_order_probe, dim3(512), dim3(1024), 0, 0, 0xC0FFEEu ^ (uint32_t)device, 4u, d);   // 12.6 M checks, < 1 ms
                    ran = hipGetLastError() == hipSuccess && hipMemcpy(&bad, d, sizeof(bad), hipMemcpyDeviceToHost) == hipSuccess;
                }
                (void)hipFree(d);
                (void)hipSetDevice(prev);
            }
            if (!ran) (void)hipGetLastError();
            g_order_verdict[device].store((ran && bad == 0) ? 1 : -1);
        }
    }
    return g_order_verdict[device].load() == 1;
}
// the verdict as the launch sites read it: every context's creation has run the probe for its device
static inline bool lds_order_cached(int device)
{
    const int forced = g_order_forced.load(std::memory_order_relaxed);
    if (forced >= 0) return forced == 1;
    return device >= 0 && device < 64 && g_order_verdict[device].load(std::memory_order_relaxed) == 1;
}
// A kernel saw the order violated at run time (bit 4 of the error word: zz_level1p.h P2, zz_level1.h warm_prehash): the device loses
// its verdict for the rest of the process -- level 1 runs k_encode_l1 from here on, warm windows and extended levels are refused.
static void lds_order_revoke(int device) { if (device >= 0 && device < 64) g_order_verdict[device].store(-1); }
// (not part of the public header) the verdict for a device: 1 holds, 0 does not
extern "C" int zz_debug_lds_order_verdict(int device) { return lds_order_ok(device) ? 1 : 0; }
// diagnostic builds only (not part of the public header): read and clear the per-phase cycle counters
extern "C" int zz_debug_read_prof(zz_ctx* c, unsigned long long out[16])
{
    if (!c) return ZZ_E_ARG;
    HIPCHK(hipMemcpy(out, c->d_prof, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(c->d_prof, 0, 16 * sizeof(unsigned long long)));
    return ZZ_OK;
}
// the same for kernels that keep one set of 16 counters per wavefront of the workgroup (k_encode_l1p: set w = wavefront w)
extern "C" int zz_debug_read_prof_sets(zz_ctx* c, unsigned long long out[64])
{
    if (!c) return ZZ_E_ARG;
    HIPCHK(hipMemcpy(out, c->d_prof, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(c->d_prof, 0, 64 * sizeof(unsigned long long)));
    return ZZ_OK;
}
// warm window. At level 1 the last `bytes` bytes (at most 32768) in front of every packet -- as far as
// they exist in the stream: a shard must pass them as its halo -- are hashed into the packet's table before it is parsed,
// so that matches may reach back across the packet boundary. 0 (the default) gives the reference's threaded stream.
extern "C" int zz_ctx_set_warm_window(zz_ctx* c, uint32_t bytes)
{
    if (!c) { set_err("null ctx"); return ZZ_E_ARG; }
    if (bytes > 32768) { set_err("warm window must be 0..32768 bytes"); return ZZ_E_ARG; }
    if (bytes && !lds_order_ok(c->device)) {
        set_err("warm window refused: this device's LDS does not serve equal addresses in lane order (probe k_lds_order_probe failed)");
        return ZZ_E_UNSUPPORTED;
    }
    c->warm = bytes;
    return ZZ_OK;
}
// levels beyond the reference. Off by default, so that the entry points keep the reference's error
// convention for level > 3; on: levels 4, 5, 6 = chains of depth 2 / 4 / 8, lazy matching, package-merge (zz_level6.h).
extern "C" int zz_ctx_set_extended_levels(zz_ctx* c, int on)
{
    if (!c) { set_err("null ctx"); return ZZ_E_ARG; }
    if (on && !lds_order_ok(c->device)) {
        set_err("extended levels refused: this device's LDS does not serve equal addresses in lane order (probe k_lds_order_probe failed)");
        return ZZ_E_UNSUPPORTED;
    }
    c->extended = on != 0;
    return ZZ_OK;
}
extern "C" void zz_ctx_enable_timing(zz_ctx* c, int on) { if (c) { c->timing = on != 0; c->have_time = false; } }
extern "C" double zz_ctx_last_kernel_ms(zz_ctx* c)
{
    if (!c || !c->timing || !c->have_time) return -1.0;
    float ms = 0;
    if (hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) return -1.0;
    return ms;
}

static int ensure_workspace(zz_ctx* c, int level, uint64_t npk, uint32_t stride, int xdepth = 0, uint32_t P = 0)
{
    if (npk > c->npk_cap) {
        (void)hipFree(c->sizes); (void)hipFree(c->offsets); (void)hipFree(c->cks);
        c->sizes = nullptr; c->offsets = nullptr; c->cks = nullptr; c->npk_cap = 0;
        HIPCHK(hipMalloc(&c->sizes, npk * sizeof(uint32_t)));
        HIPCHK(hipMalloc(&c->offsets, npk * sizeof(uint64_t)));
        HIPCHK(hipMalloc(&c->cks, npk * sizeof(zz_cks)));
        c->npk_cap = npk;
    }
    if (level != 0) {
        uint64_t need = npk * stride;
        if (need > c->slots_cap) {
            (void)hipFree(c->slots); c->slots = nullptr; c->slots_cap = 0;
            HIPCHK(hipMalloc(&c->slots, need));
            c->slots_cap = need;
        }
    }
    if (level >= 2) {
        uint64_t need = l2_scratch_bytes((uint32_t)npk, xdepth, P);
        if (need > c->l2_scratch_cap) {
            (void)hipFree(c->l2_scratch); c->l2_scratch = nullptr; c->l2_scratch_cap = 0;
            HIPCHK(hipMalloc(&c->l2_scratch, need));
            c->l2_scratch_cap = need;
        }
    }
    return ZZ_OK;
}

// ZZFLATE_L1_KERNEL=classic (diagnostic, A/B): one parsing wavefront per packet (k_encode_l1) instead of the two of k_encode_l1p;
// the streams are the same
static bool l1_classic()
{
    static const bool classic = [] { const char* e = getenv("ZZFLATE_L1_KERNEL"); return e && !strcmp(e, "classic"); }();
    return classic;
}
// ZZFLATE_L2_KERNEL=classic (diagnostic, A/B): one parsing wavefront per packet at levels 2,3 instead of two; the streams are the same
static bool l2_classic()
{
    static const bool classic = [] { const char* e = getenv("ZZFLATE_L2_KERNEL"); return e && !strcmp(e, "classic"); }();
    return classic;
}
// (not part of the public header) which level-1 kernel a cold packet-mode call on this context would launch now: 2 = k_encode_l1p
// (two parsing wavefronts), 1 = k_encode_l1 (ZZFLATE_L1_KERNEL=classic, or the device's LDS-order verdict is negative)
extern "C" int zz_debug_l1_kernel(const zz_ctx* c) { return (c && !l1_classic() && lds_order_cached(c->device)) ? 2 : 1; }
// (the same for levels 2,3: 2 = k_encode_l2p, 1 = the two-wavefront k_encode_l2_t<0, false>)
extern "C" int zz_debug_l2_kernel(const zz_ctx* c) { return (c && !l2_classic() && (!ZZ_L2P_XCHG || lds_order_cached(c->device))) ? 2 : 1; }
// The common pipeline. with_container: write header/trailer (whole stream) or not (shard).
static int encode_finish(zz_ctx* c, zz_result* host_res);
// `host_res` == nullptr: enqueue only (zz_encode_device_async); the caller collects with encode_finish.
static int encode_common(zz_ctx* c, const uint8_t* d_src, uint64_t n, uint64_t halo, bool last_is_final,
                         uint8_t* d_dst, uint64_t cap, int format, int cks_kind, bool with_container, int level,
                         uint32_t P, hipStream_t st, zz_result* host_res, bool one_parser = false)
{
    const int level_asked = level;
    bool order_checked = false;
    if (c->pend.active) { set_err("a call enqueued with zz_encode_device_async has not been finished on this context"); return ZZ_E_ARG; }
    // Levels 4..6 are beyond the reference (which rejects them, zzflate.cpp:201,230) and only exist when switched on:
    // hash chains of depth 2 / 4 / 8 over a window of 8 / 32 / 32 KiB in front of every packet, lazy matching, package-merge
    // code lengths (zz_level6.h), in the level-2 kernel's frame (dynamic blocks, stored fallback).
    uint32_t warm = level >= 1 ? c->warm : 0;
    int xdepth = 0;
    if (level >= 4 && level <= 6 && c->extended) {
        xdepth = level == 4 ? 2 : level == 5 ? 4 : 8;
        warm = level == 4 ? 8192u : 32768u;            // the levels bring their own window (zz_ctx_set_warm_window is for levels 1..3)
        level = 2;
    }
    if (level < 0 || level > 3) { set_err("level must be 0..3 (zzflate.cpp:201,230)"); return ZZ_E_LEVEL; }
    if (P == 0 || P > ZZ_MAX_PACKET_SIZE) { set_err("packet size must be 1..32768"); return ZZ_E_ARG; }
    if (!d_dst || (!d_src && n)) { set_err("null buffer"); return ZZ_E_ARG; }
    HIPCHK(hipSetDevice(c->device));
    const int hl = with_container ? header_len(format) : 0;
    const int tl = with_container ? trailer_len(format) : 0;
    if (cap < (uint64_t)hl) { set_err("destination smaller than the container header"); return ZZ_E_NOSPACE; }
    const uint64_t npk64 = (n + P - 1) / P;
    if (npk64 > 0x7FFFFFFFull) { set_err("too many packets for one call"); return ZZ_E_ARG; }
    const uint32_t npk = (uint32_t)npk64;
    const uint32_t stride = slot_stride_for(level, P);
    c->have_time = false;
    c->have_last = false;            // whatever zz_verify_last_device could look at is about to change

    HIPCHK(hipMemsetAsync(c->d_res, 0, sizeof(zz_result), st));
    HIPCHK(hipMemsetAsync(c->d_cks_total, 0, sizeof(zz_cks_total), st));
    HIPCHK(hipMemsetAsync(c->d_err, 0, sizeof(uint32_t), st));

    if (npk == 0) {
        // empty input: the reference emits no block at all, which is not a valid stream;
        // emit one empty final block instead (stored at level 0, fixed otherwise)
        uint8_t blk[5]; uint32_t bl;
        if (!last_is_final) bl = 0;
        else if (level == 0) { blk[0] = 1; blk[1] = 0; blk[2] = 0; blk[3] = 0xFF; blk[4] = 0xFF; bl = 5; }
        else { blk[0] = 0x03; blk[1] = 0x00; bl = 2; }
        if ((uint64_t)hl + bl + tl > cap) { set_err("destination too small"); return ZZ_E_NOSPACE; }
        uint64_t packed = 0;
        for (uint32_t i = 0; i < bl; ++i) packed |= (uint64_t)blk[i] << (8 * i);
        hipLaunchKernelGGL(k_put_small, dim3(1), dim3(1), 0, st, d_dst + hl, packed, bl, c->d_res, (uint64_t)bl);
    } else {
        int rc = ensure_workspace(c, level, npk, stride, xdepth, P);
        if (rc) return rc;
        zz_packet_params pp;
        pp.src = d_src; pp.n = n; pp.halo = halo; pp.packet_size = P; pp.npk = npk;
        pp.last_is_final = last_is_final ? 1 : 0; pp.cks_kind = cks_kind; pp.warm = warm;
        pp.slots = c->slots; pp.slot_stride = stride; pp.sizes = c->sizes; pp.cks = c->cks; pp.err = c->d_err; pp.prof = c->d_prof; pp.tail = c->d_tail;
        pp.dbg_viol = 0;
        // does this launch rely on the LDS's lane order (and check it as it goes: error bit 4)? k_encode_l1p; the warm window's pre-hash
        const bool l1p = level == 1 && !warm && !one_parser && !l1_classic() && lds_order_cached(c->device);
        // ... and k_encode_l2p (levels 2,3, cold): its insert is an ordered exchange (zz_level2p.h, ZZ_L2P_XCHG)
        const bool l2p = level >= 2 && !warm && xdepth == 0 && !one_parser && !l2_classic() && (!ZZ_L2P_XCHG || lds_order_cached(c->device));
        order_checked = l1p || (warm != 0 && xdepth == 0) || (l2p && ZZ_L2P_XCHG);
        if (order_checked) {
            int left = g_force_violation.load();
            while (left > 0 && !g_force_violation.compare_exchange_weak(left, left - 1)) {}
            if (left > 0) pp.dbg_viol = 1;
        }

        if (c->timing) HIPCHK(hipEventRecord(c->ev0, st));   // the CRC-32 pass of the gzip container is part of the timed work
        if (cks_kind == ZZ_CKS_CRC) {
            // (Round 3 ran this pass BESIDE the encode kernel on a second stream, in front of it or behind it, with 256..2048
            // workgroups: 96.3-96.7 GB/s against 95.9 on 1 GiB of log lines at level 1, and slower at level 2 -- its LDS table
            // lookups and VALU work come out of what the encode kernel's waiting parsers leave each other. Not worth a
            // second stream: profiles/README.md.)
            uint32_t g = npk < 2048 ? npk : 2048;   // persistent: table + shift constants are built once per block
            hipLaunchKernelGGL(k_crc32_packets, dim3(g), dim3(ZZ_CRC_THREADS), 0, st, pp);
            pp.cks_kind = ZZ_CKS_NONE;   // the encode kernel must not overwrite the CRC partials
        }
        if (level == 0) {
            const uint64_t total = (uint64_t)(npk - 1) * l0_packet_bytes(P, false) +
                                   l0_packet_bytes((uint32_t)(n - (uint64_t)(npk - 1) * P), last_is_final);
            if ((uint64_t)hl + total + tl > cap) { set_err("destination too small"); return ZZ_E_NOSPACE; }
            zz_l0_params q; q.pk = pp; q.dst = d_dst + hl; q.stream_mode = 0;
            uint32_t g = npk < 16384 ? npk : 16384;
            hipLaunchKernelGGL(k_encode_l0, dim3(g), dim3(256), 0, st, q);
            hipLaunchKernelGGL(k_put_small, dim3(1), dim3(1), 0, st, (uint8_t*)nullptr, 0ull, 0u, c->d_res, total);
        } else if (level == 1) {
            hipLaunchKernelGGL(k_fill_tail, dim3(1), dim3(128), 0, st, pp.src, pp.n, c->d_tail);      // (what reads past the shard's end reads this)
            // ZZFLATE_L1_PAD_LDS (diagnostic): extra dynamic LDS per workgroup, to measure throughput vs. resident waves
            static const unsigned pad_lds = [] { const char* e = getenv("ZZFLATE_L1_PAD_LDS"); return e ? (unsigned)atoi(e) : 0u; }();
            // (a warm window exists only where the LDS-order verdict is positive: zz_ctx_set_warm_window; k_encode_l1w = the one-parser form, A/B)
            if (pp.warm && !l1_classic()) hipLaunchKernelGGL(k_encode_l1pw, dim3(npk), dim3(ZZ_L1P_THREADS), pad_lds, st, pp);
            else if (pp.warm) hipLaunchKernelGGL(k_encode_l1w, dim3(npk), dim3(ZZ_L1_THREADS), pad_lds, st, pp);
            else if (!l1p) hipLaunchKernelGGL(k_encode_l1, dim3(npk), dim3(ZZ_L1_THREADS), pad_lds, st, pp);
            else hipLaunchKernelGGL(k_encode_l1p, dim3(npk), dim3(ZZ_L1P_THREADS), pad_lds, st, pp);
        } else {
            hipLaunchKernelGGL(k_fill_tail, dim3(1), dim3(128), 0, st, pp.src, pp.n, c->d_tail);
            launch_level2(pp, c->l2_scratch, c->d_work, st, xdepth, l2p);
        }
        if (c->timing) { HIPCHK(hipEventRecord(c->ev1, st)); c->have_time = true; }
        if (level != 0) {
            hipLaunchKernelGGL(k_scan_sizes, dim3(1), dim3(ZZ_SCAN_THREADS), 0, st, c->sizes, npk, c->offsets, c->d_res);
            uint32_t g = npk < 65536 ? npk : 65536;
            hipLaunchKernelGGL(k_compact, dim3(g), dim3(256), 0, st, c->slots, stride, c->sizes, c->offsets, npk,
                               d_dst + hl, cap >= (uint64_t)(hl + tl) ? cap - hl - tl : 0, c->d_res);
        }
        if (cks_kind != ZZ_CKS_NONE)
            hipLaunchKernelGGL(k_cks_reduce, dim3(1), dim3(ZZ_RED_THREADS), 0, st, c->cks, npk, P, n, cks_kind, c->d_cks_total);
    }
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(1), 0, st, d_dst, cap, with_container ? format : (int)ZZ_FMT_DEFLATE,
                       c->d_cks_total, n, c->d_res);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->h_res, c->d_res, sizeof(zz_result), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(c->h_err, c->d_err, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    if (npk) {
        zz_verify_params& v = c->last;     // completed by encode_finish (stream_bytes)
        v.src = d_src; v.n = n; v.halo = halo; v.packet_size = P; v.npk = npk; v.last_is_final = last_is_final ? 1 : 0;
        v.stream = d_dst + hl; v.stream_bytes = 0;
        v.offsets = level ? c->offsets : nullptr; v.sizes = level ? c->sizes : nullptr;
        v.l0_stride = (uint32_t)l0_packet_bytes(P, false);
    }
    c->pend.active = true; c->pend.st = st; c->pend.npk = npk; c->pend.level = level; c->pend.whole = with_container;
    c->pend.order_checked = order_checked;
    c->pend.d_src = d_src; c->pend.n = n; c->pend.halo = halo; c->pend.last_is_final = last_is_final; c->pend.d_dst = d_dst; c->pend.cap = cap;
    c->pend.format = format; c->pend.cks_kind = cks_kind; c->pend.level_asked = level_asked; c->pend.P = P; c->pend.warm = warm;
    if (!host_res) return ZZ_OK;
    return encode_finish(c, host_res);
}
// wait for the call encode_common enqueued on this context and collect its result
static int encode_finish(zz_ctx* c, zz_result* host_res)
{
    if (!c->pend.active) { set_err("nothing enqueued on this context"); return ZZ_E_ARG; }
    c->pend.active = false;
    HIPCHK(hipStreamSynchronize(c->pend.st));
    *host_res = *c->h_res;
    if (c->h_err[0] & 4u) {
        // A kernel of this call saw the LDS leave a LOWER lane's store in a slot that a higher lane of the same instruction wrote too
        // (zz_level1p.h P2; zz_level1.h warm_prehash; zz_level2p.h's exchange): what it wrote is a valid stream, but not necessarily
        // the reference's. The device loses its verdict; levels 1..3 run the CALL again on the one-parser kernels, which ask the LDS
        // for nothing of the kind (same bytes where the two-parser ones are right); a warm window has no such form and is refused.
        lds_order_revoke(c->device);
        if (!c->pend.order_checked) { set_err("internal: LDS-order violation reported by a kernel that does not check it"); return ZZ_E_HIP; }
        if (c->pend.warm) {
            set_err("warm window: this device's LDS served equal addresses out of lane order during the call (checked in the kernel); "
                    "the stream is valid DEFLATE but not the defined one -- warm windows are refused on this device from now on");
            return ZZ_E_UNSUPPORTED;
        }
        const auto q = c->pend;
        return encode_common(c, q.d_src, q.n, q.halo, q.last_is_final, q.d_dst, q.cap, q.format, q.cks_kind, q.whole, q.level_asked, q.P,
                             q.st, host_res, true);
    }
    if (c->h_err[0] & 8u) { set_err("internal: a wavefront of the level-2 kernel waited for its neighbour longer than a packet can take (zz_level2p.h, l2p_wait_ge)"); return ZZ_E_HIP; }
    if (c->h_err[0]) { set_err("internal: packet slot overflow"); return ZZ_E_NOSPACE; }
    if (host_res->err) { set_err("destination too small for the compressed stream"); return ZZ_E_NOSPACE; }
    if (c->pend.npk) { c->last.stream_bytes = host_res->stream_bytes; c->have_last = true; }
    return ZZ_OK;
}

// inflate every packet of the stream the last packet-mode call on this context produced (device
// entry points; its source and destination must still be in place) and compare with that call's input.
extern "C" int zz_verify_last_device(zz_ctx* c, uint64_t* bad_packets, uint64_t* first_bad_packet, void* hip_stream)
{
    if (!c || !bad_packets) { set_err("null argument"); return ZZ_E_ARG; }
    if (!c->have_last) { set_err("no packet-mode call to verify on this context"); return ZZ_E_ARG; }
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)hip_stream;
    if (!c->d_verify) HIPCHK(hipMalloc(&c->d_verify, 2 * sizeof(unsigned long long)));
    const unsigned long long init[2] = { 0ull, ~0ull };
    HIPCHK(hipMemcpyAsync(c->d_verify, init, sizeof init, hipMemcpyHostToDevice, st));
    zz_verify_params v = c->last;
    v.out = c->d_verify;
    hipLaunchKernelGGL(k_verify_packets, dim3((v.npk + 63) / 64), dim3(64), 0, st, v);
    HIPCHK(hipGetLastError());
    unsigned long long res[2];
    HIPCHK(hipMemcpyAsync(res, c->d_verify, sizeof res, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    *bad_packets = res[0];
    if (first_bad_packet) *first_bad_packet = res[1];
    return ZZ_OK;
}

// Where packet k of the stream the last packet-mode call on this context produced lies: *offset counts from the first
// byte of the DEFLATE stream (behind the container header), *bytes is the packet's length. Packets are independent
// (cold table, byte-aligned: zzflate.cpp:101-125), so this is a random-access index into the stream -- and what the
// full-size tests use to compare sampled packets of a multi-GiB call with the oracle.
extern "C" int zz_packet_extent_device(zz_ctx* c, uint64_t packet, uint64_t* offset, uint64_t* bytes, void* hip_stream)
{
    if (!c || !offset || !bytes) { set_err("null argument"); return ZZ_E_ARG; }
    if (!c->have_last) { set_err("no packet-mode call on this context"); return ZZ_E_ARG; }
    const zz_verify_params& v = c->last;
    if (packet >= v.npk) { set_err("packet index out of range"); return ZZ_E_ARG; }
    HIPCHK(hipSetDevice(c->device));
    if (!v.offsets) {                                         // level 0: every packet has its known place
        *offset = packet * v.l0_stride;
        *bytes = packet + 1 < v.npk ? v.l0_stride : v.stream_bytes - packet * v.l0_stride;
        return ZZ_OK;
    }
    uint64_t o = 0; uint32_t sz = 0;
    hipStream_t st = (hipStream_t)hip_stream;
    HIPCHK(hipMemcpyAsync(&o, v.offsets + packet, sizeof o, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&sz, v.sizes + packet, sizeof sz, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    *offset = o; *bytes = sz;
    return ZZ_OK;
}

static int cks_kind_for(int format) { return format == ZZ_ZLIB ? ZZ_CKS_ADLER : format == ZZ_GZIP ? ZZ_CKS_CRC : ZZ_CKS_NONE; }

// The reference's sequential whole-buffer stream (threaded=false, zzflate.cpp:84-95) on the device. Level 0 is
// parallel (stored blocks of 65535 bytes have known places); level 1 is a chain of fixed-Huffman blocks whose lengths
// follow from the room in the output buffer (encoder.cpp:331-337: ONE block when the destination is roomy), levels
// 2,3 a chain of dynamic blocks, each chain produced by a single wavefront (k_stream_l1 / k_stream_l2) -- a
// compatibility mode, bit-identical to the reference, not a fast one.
//   chunked == false: ZzFlateEncode's caller-owned buffer of `cap` bytes (zzflate.cpp:225-242);
//   chunked == true : ZzFlateEncodeToCallback's library-owned chunks of 1,000,000 bytes (zzflate.cpp:197-222,
//                     outputbitstream.h:171-201); *chunk_sizes receives the byte counts the callback would see between the
//                     header call and the trailer call, `cap` only has to hold the stream.
static int encode_stream(zz_ctx* c, const uint8_t* d_src, uint64_t n, uint8_t* d_dst, uint64_t cap, int format, int level,
                         bool chunked, std::vector<uint64_t>* chunk_sizes, hipStream_t st, zz_result* host_res)
{
    if (c->pend.active) { set_err("a call enqueued with zz_encode_device_async has not been finished on this context"); return ZZ_E_ARG; }
    if (level < 0 || level > 3) { set_err("level must be 0..3 (zzflate.cpp:201,230)"); return ZZ_E_LEVEL; }
    if (!d_dst || (!d_src && n)) { set_err("null buffer"); return ZZ_E_ARG; }
    if (chunk_sizes) chunk_sizes->clear();
    if (n == 0) {
        int rc = encode_common(c, d_src, 0, 0, true, d_dst, cap, format, cks_kind_for(format), true, level, ZZ_DEFAULT_PACKET, st, host_res);
        if (!rc && chunk_sizes && host_res->stream_bytes) chunk_sizes->push_back(host_res->stream_bytes);
        return rc;
    }
    if (n >= (1ull << 31)) { set_err("sequential stream: input must be < 2 GiB (the reference funnels lengths through int)"); return ZZ_E_ARG; }
    HIPCHK(hipSetDevice(c->device));
    const int hl = header_len(format), tl = trailer_len(format);
    if (cap < (uint64_t)hl) { set_err("destination smaller than the container header"); return ZZ_E_NOSPACE; }
    const int cks_kind = cks_kind_for(format);
    c->have_time = false;
    c->have_last = false;
    HIPCHK(hipMemsetAsync(c->d_res, 0, sizeof(zz_result), st));
    HIPCHK(hipMemsetAsync(c->d_cks_total, 0, sizeof(zz_cks_total), st));
    HIPCHK(hipMemsetAsync(c->d_err, 0, 4 * sizeof(uint32_t), st));
    zz_packet_params pp;
    memset(&pp, 0, sizeof pp);
    pp.src = d_src; pp.n = n; pp.halo = 0; pp.last_is_final = 1; pp.err = c->d_err; pp.prof = c->d_prof; pp.tail = c->d_tail;
    std::vector<uint64_t> log;           // (bytes stored, length asked for) per EnsureOutputLength call, chunked form
    uint32_t log_cap = 0;
    if (level == 0) {
        const uint32_t B = 0xFFFF;                                     // encoder.cpp:484
        const uint32_t npk = (uint32_t)((n + B - 1) / B);
        int rc = ensure_workspace(c, 0, npk, 0);
        if (rc) return rc;
        const uint64_t total = (uint64_t)(npk - 1) * (B + 5) + 5 + (n - (uint64_t)(npk - 1) * B);
        if ((uint64_t)hl + total + tl > cap) { set_err("destination too small"); return ZZ_E_NOSPACE; }
        pp.packet_size = B; pp.npk = npk; pp.cks_kind = cks_kind; pp.sizes = c->sizes; pp.cks = c->cks;
        if (cks_kind == ZZ_CKS_CRC) {
            hipLaunchKernelGGL(k_crc32_packets, dim3(npk < 2048 ? npk : 2048), dim3(ZZ_CRC_THREADS), 0, st, pp);
            pp.cks_kind = ZZ_CKS_NONE;
        }
        zz_l0_params q; q.pk = pp; q.dst = d_dst + hl; q.stream_mode = 1;
        hipLaunchKernelGGL(k_encode_l0, dim3(npk < 16384 ? npk : 16384), dim3(256), 0, st, q);
        hipLaunchKernelGGL(k_put_small, dim3(1), dim3(1), 0, st, (uint8_t*)nullptr, 0ull, 0u, c->d_res, total);
        if (cks_kind != ZZ_CKS_NONE)
            hipLaunchKernelGGL(k_cks_reduce, dim3(1), dim3(ZZ_RED_THREADS), 0, st, c->cks, npk, B, n, cks_kind, c->d_cks_total);
        if (chunked && chunk_sizes)                                     // WriteUncompressedBlock asks for 6 + length (encoder.cpp:488)
            for (uint32_t k = 0; k < npk; ++k) {
                log.push_back((uint64_t)k * (B + 5));
                log.push_back(6 + (k + 1 < npk ? (uint64_t)B : n - (uint64_t)(npk - 1) * B));
            }
    } else {
        // worst cases. Level 1: nine bits per byte plus ten per block; blocks of the chunked form hold at least
        // (2^18 - 1) * 8 / 9 - 8 bytes each, and a caller-owned buffer is never overrun (the block lengths see to it).
        // Level >= 2: every block falls back to stored blocks of <= 65535 bytes.
        const uint64_t avail = cap - hl;
        uint64_t bound = level == 1 ? ((uint64_t)9 * n + 17) / 8 + n / 65536 + 128 : n + (n / 65535 + 2) * 5 + (n / 400000 + 2) * 8 + 64;
        if (level == 1 && !chunked && avail + 64 < bound) bound = avail + 64;
        const uint32_t P = 32768, npk_c = (uint32_t)((n + P - 1) / P);   // checksum chunks
        int rc = ensure_workspace(c, 0, npk_c, 0);
        if (rc) return rc;
        if (bound > c->slots_cap) {
            (void)hipFree(c->slots); c->slots = nullptr; c->slots_cap = 0;
            HIPCHK(hipMalloc(&c->slots, bound));
            c->slots_cap = bound;
        }
        zz_stream_ctl ctl;
        memset(&ctl, 0, sizeof ctl);
        ctl.cap = avail; ctl.chunked = chunked ? 1 : 0; ctl.truncated = c->d_err + 1; ctl.log_n = c->d_err + 2;
        if (chunked) {
            log_cap = (uint32_t)(n / 32768 + 64);                        // far more than the blocks a stream can have
            if ((uint64_t)log_cap * 16 > c->log_cap_bytes) {
                (void)hipFree(c->d_log); c->d_log = nullptr; c->log_cap_bytes = 0;
                HIPCHK(hipMalloc(&c->d_log, (uint64_t)log_cap * 16));
                c->log_cap_bytes = (uint64_t)log_cap * 16;
            }
            ctl.log = c->d_log; ctl.log_cap = log_cap;
        }
        pp.packet_size = P; pp.npk = npk_c; pp.cks_kind = cks_kind; pp.cks = c->cks; pp.sizes = c->sizes;
        if (cks_kind == ZZ_CKS_CRC) hipLaunchKernelGGL(k_crc32_packets, dim3(npk_c < 2048 ? npk_c : 2048), dim3(ZZ_CRC_THREADS), 0, st, pp);
        else if (cks_kind == ZZ_CKS_ADLER) hipLaunchKernelGGL(k_adler_packets, dim3(npk_c < 4096 ? npk_c : 4096), dim3(ZZ_WAVE), 0, st, pp);
        zz_packet_params ps = pp;
        ps.npk = 1; ps.slots = c->slots; ps.slot_stride = (uint32_t)(bound > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : bound); ps.cks_kind = ZZ_CKS_NONE;
        if (level >= 2 && (uint64_t)ZZ_ST_SCRATCH_BYTES > c->l2_scratch_cap) {
            (void)hipFree(c->l2_scratch); c->l2_scratch = nullptr; c->l2_scratch_cap = 0;
            HIPCHK(hipMalloc(&c->l2_scratch, ZZ_ST_SCRATCH_BYTES));
            c->l2_scratch_cap = ZZ_ST_SCRATCH_BYTES;
        }
        if (c->timing) HIPCHK(hipEventRecord(c->ev0, st));
        if (level == 1) hipLaunchKernelGGL(k_stream_l1, dim3(1), dim3(ZZ_WAVE), 0, st, ps, ctl);
        else { zz_st_params q; q.pk = ps; q.scratch = c->l2_scratch; q.ctl = ctl; hipLaunchKernelGGL(k_stream_l2, dim3(1), dim3(ZZ_WAVE), 0, st, q); }
        if (c->timing) { HIPCHK(hipEventRecord(c->ev1, st)); c->have_time = true; }
        hipLaunchKernelGGL(k_copy_stream, dim3(1024), dim3(256), 0, st, c->slots, c->sizes, d_dst + hl,
                           cap >= (uint64_t)(hl + tl) ? cap - hl - tl : 0, c->d_res);
        if (cks_kind != ZZ_CKS_NONE)
            hipLaunchKernelGGL(k_cks_reduce, dim3(1), dim3(ZZ_RED_THREADS), 0, st, c->cks, npk_c, P, n, cks_kind, c->d_cks_total);
    }
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(1), 0, st, d_dst, cap, format, c->d_cks_total, n, c->d_res);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->h_res, c->d_res, sizeof(zz_result), hipMemcpyDeviceToHost, st));
    // (into pinned memory: an asynchronous copy into pageable memory is staged by the runtime and makes concurrent callers on
    // other streams take turns)
    HIPCHK(hipMemcpyAsync(c->h_err, c->d_err, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const uint32_t kerr[4] = { c->h_err[0], c->h_err[1], c->h_err[2], c->h_err[3] };
    *host_res = *c->h_res;
    if (kerr[0]) { set_err("internal: output slot overflow"); return ZZ_E_NOSPACE; }
    if (kerr[1]) {
        set_err("destination too small: the reference's level-1 stream stops early here (encoder.cpp:331-337,546-548)");
        return ZZ_E_NOSPACE;
    }
    if (host_res->err) { set_err("destination too small for the compressed stream"); return ZZ_E_NOSPACE; }
    if (chunked && chunk_sizes) {
        if (level != 0) {
            const uint32_t nlog = kerr[2];
            if (nlog > log_cap) { set_err("internal: chunk log overflow"); return ZZ_E_HIP; }
            log.resize((size_t)nlog * 2);
            if (nlog) HIPCHK(hipMemcpy(log.data(), c->d_log, (size_t)nlog * 16, hipMemcpyDeviceToHost));
        }
        // replay EnsureOutputLength (outputbitstream.h:171-190) over the log: a chunk starts where the rule opens one
        zz_chunker ck = { 0, 0 };
        std::vector<uint64_t> starts;
        for (size_t k = 0; k + 1 < log.size(); k += 2) {
            bool opened;
            const int64_t room = zz_chunk_ensure(ck, log[k], (int64_t)log[k + 1], &opened);
            if (level >= 2 && !opened && room < (int64_t)log[k + 1]) {
                // a dynamic block that needs more than the > 2^18 bytes still free: the reference gives up here
                // (encoder.cpp:277-278) and leaves an undecodable stream; the block is in ours, so it gets a chunk of
                // its own. (At level 1 the length asked for is only a wish: the block is cut to what fits.)
                ck.chunk_start = log[k]; ck.nchunks++; opened = true;
            }
            if (opened) starts.push_back(log[k]);
        }
        for (size_t k = 0; k < starts.size(); ++k) {
            const uint64_t e = k + 1 < starts.size() ? starts[k + 1] : host_res->stream_bytes;
            chunk_sizes->push_back(e - starts[k]);
        }
    }
    return ZZ_OK;
}

// The reference's OWN threaded=true split (zzflate.cpp:67-78 divideInRanges, :97-155): `count` ranges of ceil(n / count) bytes
// -- count is std::thread::hardware_concurrency() there, the caller's choice here --, every range a fresh encoder, joined in
// order. Packet mode replaces this split by fixed ranges of <= 32 KiB because a GPU wants thousands of ranges;
// this entry point reproduces the reference's bytes for a given count: one WAVEFRONT per range (k_stream_l2: a range is one
// dependency chain, and longer than the packet kernels' 16-bit positions allow), so it is a compatibility mode like the
// sequential stream. Levels 0, 2, 3: at level 1 the reference's threaded stream is invalid (the block
// lengths follow from the room, destLen / count per range, and the joined stream does not inflate), so there is nothing to equal.
// every range of the reference's split starts inside the input: (count - 1) * ceil(n / count) < n
static bool ranges_split_ok(uint64_t n, uint32_t count)
{
    if (count == 0) return false;
    const uint64_t step = (n + count - 1) / count;
    return (uint64_t)(count - 1) * step < n;
}
static int encode_ranges(zz_ctx* c, const uint8_t* d_src, uint64_t n, uint8_t* d_dst, uint64_t cap, int format, int level,
                         uint32_t count, hipStream_t st, zz_result* host_res)
{
    if (count == 0 || count > 4096) { set_err("ranges: count must be 1..4096"); return ZZ_E_ARG; }
    if (level == 1) { set_err("ranges: the reference's threaded level-1 stream is invalid (block lengths follow from destLen / count); use packet mode"); return ZZ_E_LEVEL; }
    if (n < 100ull * count)                                               // zzflate.cpp:84: the single encoder
        return encode_stream(c, d_src, n, d_dst, cap, format, level, false, nullptr, st, host_res);
    if (c->pend.active) { set_err("a call enqueued with zz_encode_device_async has not been finished on this context"); return ZZ_E_ARG; }
    if (level < 0 || level > 3) { set_err("level must be 0..3 (zzflate.cpp:201,230)"); return ZZ_E_LEVEL; }
    if (!d_dst || !d_src) { set_err("null buffer"); return ZZ_E_ARG; }
    if (n >= (1ull << 31)) { set_err("ranges: input must be < 2 GiB (the reference funnels lengths through int)"); return ZZ_E_ARG; }
    HIPCHK(hipSetDevice(c->device));
    const int hl = header_len(format), tl = trailer_len(format);
    if (cap < (uint64_t)hl) { set_err("destination smaller than the container header"); return ZZ_E_NOSPACE; }
    const int cks_kind = cks_kind_for(format);
    const uint64_t step = (n + count - 1) / count;                        // zzflate.cpp:70
    // divideInRanges (zzflate.cpp:67-78) puts boundary i at step * i and only the last one at n: with count near sqrt(n) or
    // above, the trailing boundaries lie at or past n (the reference then reads past its input and joins
    // a stream that does not inflate -- undefined behaviour, nothing to equal). Refused here, before anything is launched.
    if (!ranges_split_ok(n, count)) {
        set_err("ranges: count too large for this input: (count - 1) * ceil(n / count) must be < n (zzflate.cpp:67-78 would cut ranges past the input's end)");
        return ZZ_E_ARG;
    }
    c->have_time = false;
    c->have_last = false;
    HIPCHK(hipMemsetAsync(c->d_res, 0, sizeof(zz_result), st));
    HIPCHK(hipMemsetAsync(c->d_cks_total, 0, sizeof(zz_cks_total), st));
    HIPCHK(hipMemsetAsync(c->d_err, 0, 4 * sizeof(uint32_t), st));
    const uint32_t P = 32768, npk_c = (uint32_t)((n + P - 1) / P);         // checksum chunks
    int rc = ensure_workspace(c, 0, npk_c > count ? npk_c : count, 0);
    if (rc) return rc;
    zz_packet_params pp;
    memset(&pp, 0, sizeof pp);
    pp.src = d_src; pp.n = n; pp.halo = 0; pp.last_is_final = 1; pp.err = c->d_err; pp.prof = c->d_prof; pp.tail = c->d_tail;
    pp.packet_size = P; pp.npk = npk_c; pp.cks_kind = cks_kind; pp.cks = c->cks; pp.sizes = c->sizes;
    if (cks_kind == ZZ_CKS_CRC) hipLaunchKernelGGL(k_crc32_packets, dim3(npk_c < 2048 ? npk_c : 2048), dim3(ZZ_CRC_THREADS), 0, st, pp);
    else if (cks_kind == ZZ_CKS_ADLER) hipLaunchKernelGGL(k_adler_packets, dim3(npk_c < 4096 ? npk_c : 4096), dim3(ZZ_WAVE), 0, st, pp);
    if (level == 0) {
        const uint64_t total = (uint64_t)(count - 1) * l0_range_bytes(step, false) + l0_range_bytes(n - (uint64_t)(count - 1) * step, true);
        if ((uint64_t)hl + total + tl > cap) { set_err("destination too small"); return ZZ_E_NOSPACE; }
        hipLaunchKernelGGL(k_ranges_l0, dim3(count), dim3(256), 0, st, d_src, n, step, d_dst + hl);
        hipLaunchKernelGGL(k_put_small, dim3(1), dim3(1), 0, st, (uint8_t*)nullptr, 0ull, 0u, c->d_res, total);
    } else {
        // per range: every block may fall back to stored blocks of <= 65535 bytes, plus the closing stored byte
        const uint64_t bound = (step + (step / 65535 + 2) * 5 + (step / 400000 + 2) * 8 + 64 + 15) & ~15ull;
        if (bound > 0xFFFFFFF0ull) { set_err("ranges: range too large"); return ZZ_E_ARG; }
        if (bound * count > c->slots_cap) {
            (void)hipFree(c->slots); c->slots = nullptr; c->slots_cap = 0;
            HIPCHK(hipMalloc(&c->slots, bound * count));
            c->slots_cap = bound * count;
        }
        if ((uint64_t)ZZ_ST_SCRATCH_BYTES * count > c->l2_scratch_cap) {
            (void)hipFree(c->l2_scratch); c->l2_scratch = nullptr; c->l2_scratch_cap = 0;
            HIPCHK(hipMalloc(&c->l2_scratch, (uint64_t)ZZ_ST_SCRATCH_BYTES * count));
            c->l2_scratch_cap = (uint64_t)ZZ_ST_SCRATCH_BYTES * count;
        }
        zz_st_params q;
        memset(&q, 0, sizeof q);
        q.pk = pp; q.pk.npk = count; q.pk.slots = c->slots; q.pk.slot_stride = (uint32_t)bound; q.pk.cks_kind = ZZ_CKS_NONE;
        q.scratch = c->l2_scratch; q.range_step = step;
        q.ctl.cap = ~0ull;
        if (c->timing) HIPCHK(hipEventRecord(c->ev0, st));
        hipLaunchKernelGGL(k_stream_l2, dim3(count), dim3(ZZ_WAVE), 0, st, q);
        if (c->timing) { HIPCHK(hipEventRecord(c->ev1, st)); c->have_time = true; }
        hipLaunchKernelGGL(k_scan_sizes, dim3(1), dim3(ZZ_SCAN_THREADS), 0, st, c->sizes, count, c->offsets, c->d_res);
        hipLaunchKernelGGL(k_compact, dim3(count), dim3(256), 0, st, c->slots, (uint32_t)bound, c->sizes, c->offsets, count,
                           d_dst + hl, cap >= (uint64_t)(hl + tl) ? cap - hl - tl : 0, c->d_res);
    }
    if (cks_kind != ZZ_CKS_NONE)
        hipLaunchKernelGGL(k_cks_reduce, dim3(1), dim3(ZZ_RED_THREADS), 0, st, c->cks, npk_c, P, n, cks_kind, c->d_cks_total);
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(1), 0, st, d_dst, cap, format, c->d_cks_total, n, c->d_res);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->h_res, c->d_res, sizeof(zz_result), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(c->h_err, c->d_err, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    *host_res = *c->h_res;
    if (c->h_err[0]) { set_err("internal: output slot overflow"); return ZZ_E_NOSPACE; }
    if (host_res->err) { set_err("destination too small for the compressed stream"); return ZZ_E_NOSPACE; }
    return ZZ_OK;
}

extern "C" int zz_encode_ranges_device(zz_ctx* c, const void* d_src, uint64_t n, void* d_dst, uint64_t cap, uint64_t* out_len,
                                       int format, int level, uint32_t count, void* hip_stream)
{
    if (out_len) *out_len = ~0ull;
    if (!c || !out_len) { set_err("null argument"); return ZZ_E_ARG; }
    if (format < 0 || format > 2) format = ZZ_DEFLATE;
    zz_result r;
    int rc = encode_ranges(c, (const uint8_t*)d_src, n, (uint8_t*)d_dst, cap, format, level, count, (hipStream_t)hip_stream, &r);
    if (rc) return rc;
    *out_len = r.total_bytes;
    return ZZ_OK;
}

extern "C" int zz_encode_stream_device(zz_ctx* c, const void* d_src, uint64_t n, void* d_dst, uint64_t cap, uint64_t* out_len,
                                       int format, int level, void* hip_stream)
{
    if (out_len) *out_len = ~0ull;
    if (!c || !out_len) { set_err("null ctx/out_len"); return ZZ_E_ARG; }
    if (format < 0 || format > 2) format = ZZ_DEFLATE;
    zz_result r;
    int rc = encode_stream(c, (const uint8_t*)d_src, n, (uint8_t*)d_dst, cap, format, level, false, nullptr, (hipStream_t)hip_stream, &r);
    if (rc) return rc;
    *out_len = r.total_bytes;
    return ZZ_OK;
}

extern "C" int zz_encode_stream_chunks_device(zz_ctx* c, const void* d_src, uint64_t n, void* d_dst, uint64_t cap, uint64_t* out_len,
                                              int format, int level, uint64_t* chunk_sizes, uint32_t max_chunks,
                                              uint32_t* nchunks, void* hip_stream)
{
    if (out_len) *out_len = ~0ull;
    if (nchunks) *nchunks = 0;
    if (!c || !out_len) { set_err("null ctx/out_len"); return ZZ_E_ARG; }
    if (format < 0 || format > 2) format = ZZ_DEFLATE;
    zz_result r;
    std::vector<uint64_t> sizes;
    int rc = encode_stream(c, (const uint8_t*)d_src, n, (uint8_t*)d_dst, cap, format, level, true, &sizes, (hipStream_t)hip_stream, &r);
    if (rc) return rc;
    *out_len = r.total_bytes;
    if (nchunks) *nchunks = (uint32_t)sizes.size();
    if (chunk_sizes) for (size_t k = 0; k < sizes.size() && k < max_chunks; ++k) chunk_sizes[k] = sizes[k];
    return ZZ_OK;
}


extern "C" int zz_encode_device(zz_ctx* c, const void* d_src, uint64_t n, void* d_dst, uint64_t cap, uint64_t* out_len,
                                int format, int level, uint32_t P, void* hip_stream)
{
    if (out_len) *out_len = ~0ull;
    if (!c || !out_len) { set_err("null ctx/out_len"); return ZZ_E_ARG; }
    if (format < 0 || format > 2) format = ZZ_DEFLATE;   // zzflate.cpp:39-48 default branch
    if (P == 0) P = ZZ_DEFAULT_PACKET;
    zz_result r;
    int rc = encode_common(c, (const uint8_t*)d_src, n, 0, true, (uint8_t*)d_dst, cap, format, cks_kind_for(format),
                           true, level, P, (hipStream_t)hip_stream, &r);
    if (rc) return rc;
    *out_len = r.total_bytes;
    return ZZ_OK;
}

// The same call in two halves, so that several calls -- on several contexts, each with its own stream -- can be in flight:
// the next call's encode kernel then fills the CUs the previous one's last packets leave idle, and runs under its
// compaction, checksum fold and result copy. zz_encode_device_async enqueues everything and returns; zz_encode_finish
// waits for it and hands out the length (or the error). One call per context at a time; source, destination and the
// stream must stay alive in between.
extern "C" int zz_encode_device_async(zz_ctx* c, const void* d_src, uint64_t n, void* d_dst, uint64_t cap, int format, int level,
                                      uint32_t P, void* hip_stream)
{
    if (!c) { set_err("null ctx"); return ZZ_E_ARG; }
    if (format < 0 || format > 2) format = ZZ_DEFLATE;
    if (P == 0) P = ZZ_DEFAULT_PACKET;
    return encode_common(c, (const uint8_t*)d_src, n, 0, true, (uint8_t*)d_dst, cap, format, cks_kind_for(format), true, level, P,
                         (hipStream_t)hip_stream, nullptr);
}
extern "C" int zz_encode_finish(zz_ctx* c, uint64_t* out_len)
{
    if (out_len) *out_len = ~0ull;
    if (!c || !out_len) { set_err("null ctx/out_len"); return ZZ_E_ARG; }
    zz_result r;
    int rc = encode_finish(c, &r);
    if (rc) return rc;
    *out_len = r.total_bytes;
    return ZZ_OK;
}

extern "C" int zz_encode_shard_device(zz_ctx* c, const void* d_src, uint64_t n, uint64_t halo, int is_last, void* d_dst,
                                      uint64_t cap, uint64_t* out_len, uint32_t* cks, int checksum, int level,
                                      uint32_t P, void* hip_stream)
{
    if (out_len) *out_len = ~0ull;
    if (!c || !out_len) { set_err("null ctx/out_len"); return ZZ_E_ARG; }
    if (P == 0) P = ZZ_DEFAULT_PACKET;
    zz_result r;
    int rc = encode_common(c, (const uint8_t*)d_src, n, halo, is_last != 0, (uint8_t*)d_dst, cap, ZZ_DEFLATE,
                           cks_kind_for(checksum), false, level, P, (hipStream_t)hip_stream, &r);
    if (rc) return rc;
    *out_len = r.stream_bytes;
    if (cks) *cks = checksum == ZZ_ZLIB ? ((r.cks_b << 16) | r.cks_a) : r.cks_a;
    return ZZ_OK;
}

// The shard call in two halves (as zz_encode_device_async / zz_encode_finish): a rank can enqueue the next step's shard
// before it has exchanged the previous one's size and checksum with its peers, so the GPU never waits for the host.
extern "C" int zz_encode_shard_device_async(zz_ctx* c, const void* d_src, uint64_t n, uint64_t halo, int is_last, void* d_dst,
                                            uint64_t cap, int checksum, int level, uint32_t P, void* hip_stream)
{
    if (!c) { set_err("null ctx"); return ZZ_E_ARG; }
    if (P == 0) P = ZZ_DEFAULT_PACKET;
    return encode_common(c, (const uint8_t*)d_src, n, halo, is_last != 0, (uint8_t*)d_dst, cap, ZZ_DEFLATE, cks_kind_for(checksum),
                         false, level, P, (hipStream_t)hip_stream, nullptr);
}
extern "C" int zz_encode_shard_finish(zz_ctx* c, uint64_t* out_len, uint32_t* cks, int checksum)
{
    if (out_len) *out_len = ~0ull;
    if (!c || !out_len) { set_err("null ctx/out_len"); return ZZ_E_ARG; }
    zz_result r;
    int rc = encode_finish(c, &r);
    if (rc) return rc;
    *out_len = r.stream_bytes;
    if (cks) *cks = checksum == ZZ_ZLIB ? ((r.cks_b << 16) | r.cks_a) : r.cks_a;
    return ZZ_OK;
}

static int ensure_stage(zz_ctx* c, uint64_t in_bytes, uint64_t out_bytes);
extern "C" int zz_header(int format, uint8_t out[10]);
extern "C" int zz_trailer(int format, uint32_t v, uint64_t n, uint8_t out[8]);
// The reference's fan-out and join (WriteDeflateStream, zzflate.cpp:97-155: ranges -> std::async encoders -> in-order
// memmove) for data that is ALREADY RESIDENT on several GPUs of one process -- the north star's dataflow without
// torch.distributed. Shard i (contiguous ranges of one stream, in order, every shard but the last a whole number of
// packets) lives on the device of ctxs[i]; all shards are encoded concurrently, each on its own device and stream; as
// shard i finishes, its compressed bytes are pulled over xGMI (hipMemcpyPeerAsync) straight to their final offset in
// d_dst, which lives on the device of ctxs[0] (shard 0 is encoded in place there); the checksum partials are folded on
// the host (adler.cpp:5-15 / GF(2) shifts) and header and trailer are written around the stream. No bulk collective.
// ---- peer access for the pulls of zz_encode_multi_device ------------------------------------------------------------------
// hipMemcpyPeerAsync works with or without peer access; without it the copy is staged through host memory instead of going
// over xGMI. So: once per ordered device pair (src -> dst), ask hipDeviceCanAccessPeer and enable access both ways; the answer
// is kept, "already enabled" counts as enabled, and zz_debug_peer_state tells a caller (and the tests) what a pull will use.
static std::mutex g_peer_mu;
static signed char g_peer[64][64];          // [dst][src]: 0 unknown, 1 peer access enabled, -1 not available (copies are staged)
static int peer_prepare(int dst, int src)
{
    if (dst == src) return 1;
    if (dst < 0 || src < 0 || dst >= 64 || src >= 64) return -1;
    std::lock_guard<std::mutex> lk(g_peer_mu);
    if (g_peer[dst][src] == 0) {
        int prev = 0;
        (void)hipGetDevice(&prev);
        bool ok = true;
        for (int dir = 0; dir < 2 && ok; ++dir) {                    // dst reads/writes src's memory and the other way round
            const int a = dir ? src : dst, b = dir ? dst : src;
            int can = 0;
            ok = hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can != 0 && hipSetDevice(a) == hipSuccess;
            if (!ok) (void)hipGetLastError();
            if (ok) {
                const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
                if (e != hipSuccess) (void)hipGetLastError();    // whatever it was: the thread's last-error word must not leak into the next launch check
                ok = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;             // (idempotent)
            }
        }
        (void)hipSetDevice(prev);
        g_peer[dst][src] = g_peer[src][dst] = ok ? 1 : -1;
    }
    return g_peer[dst][src];
}
// (not part of the public header) 1: pulls from `src` to `dst` go peer to peer; -1: staged through the host; 0: not asked yet
extern "C" int zz_debug_peer_state(int dst, int src)
{
    if (dst == src) return 1;
    if (dst < 0 || src < 0 || dst >= 64 || src >= 64) return -1;
    std::lock_guard<std::mutex> lk(g_peer_mu);
    return g_peer[dst][src];
}
// (not part of the public header) how many of the last zz_encode_multi_device call's pulls (this thread) went peer to peer / were staged
static thread_local int tl_pulls_p2p = 0, tl_pulls_staged = 0;
extern "C" void zz_debug_last_pulls(int* p2p, int* staged) { if (p2p) *p2p = tl_pulls_p2p; if (staged) *staged = tl_pulls_staged; }

extern "C" int zz_encode_multi_device(zz_ctx* const* ctxs, int nshards, const void* const* d_src, const uint64_t* n,
                                      const uint64_t* halo, void* d_dst, uint64_t cap, uint64_t* out_len, int format, int level,
                                      uint32_t P)
{
    if (out_len) *out_len = ~0ull;
    if (!ctxs || nshards < 1 || !d_src || !n || !d_dst || !out_len) { set_err("null argument"); return ZZ_E_ARG; }
    if (format < 0 || format > 2) format = ZZ_DEFLATE;
    if (P == 0) P = ZZ_DEFAULT_PACKET;
    if (P > ZZ_MAX_PACKET_SIZE) { set_err("packet size must be 1..32768"); return ZZ_E_ARG; }
    for (int i = 0; i < nshards; ++i) {
        if (!ctxs[i]) { set_err("null context"); return ZZ_E_ARG; }
        for (int j = 0; j < i; ++j) if (ctxs[j] == ctxs[i]) { set_err("every shard needs a context of its own"); return ZZ_E_ARG; }
        if (i + 1 < nshards && n[i] % P) { set_err("every shard but the last must be a whole number of packets"); return ZZ_E_ARG; }
        if (i + 1 < nshards && n[i] == 0) { set_err("empty shard in front of the last one"); return ZZ_E_ARG; }
    }
    const int hl = header_len(format), tl = trailer_len(format);
    if (cap < (uint64_t)hl) { set_err("destination smaller than the container header"); return ZZ_E_NOSPACE; }
    uint8_t* dst = (uint8_t*)d_dst;
    zz_ctx* c0 = ctxs[0];
    // the pulls' way: peer access between the first device and every other one, asked for and enabled once per pair
    tl_pulls_p2p = tl_pulls_staged = 0;
    for (int i = 1; i < nshards; ++i) (void)peer_prepare(c0->device, ctxs[i]->device);
    // enqueue every shard on its own device and stream; nothing waits yet
    std::vector<uint64_t> bound(nshards);
    for (int i = 0; i < nshards; ++i) {
        zz_ctx* c = ctxs[i];
        HIPCHK(hipSetDevice(c->device));
        if (!c->s_enc) HIPCHK(hipStreamCreateWithFlags(&c->s_enc, hipStreamNonBlocking));
        bound[i] = zz_bound(n[i], ZZ_DEFLATE, level > 3 ? 3 : level, P);
        uint8_t* out = nullptr;
        uint64_t ocap = 0;
        if (i == 0) { out = dst + hl; ocap = cap - hl; }               // final place: offset known
        else { int rc = ensure_stage(c, 0, bound[i]); if (rc) return rc; out = c->stage_out; ocap = c->stage_out_cap; }
        int rc = zz_encode_shard_device_async(c, d_src[i], n[i], halo ? halo[i] : 0, i + 1 == nshards, out, ocap, format, level, P,
                                              (void*)c->s_enc);
        if (rc) {           // finish what was enqueued so far: the contexts stay usable
            for (int j = 0; j < i; ++j) { uint64_t w; (void)zz_encode_shard_finish(ctxs[j], &w, nullptr, format); }
            return rc;
        }
    }
    // in-order join: shard i's offset is the sum of the sizes in front of it
    uint64_t off = (uint64_t)hl;
    uint32_t acc = format == ZZ_ZLIB ? 1u : 0u;
    uint64_t total_in = 0;
    int err = ZZ_OK;
    std::string errmsg;
    for (int i = 0; i < nshards; ++i) {
        zz_ctx* c = ctxs[i];
        uint64_t w = 0; uint32_t part = 0;
        int rc = zz_encode_shard_finish(c, &w, &part, format);
        if (rc && !err) { err = rc; errmsg = g_err; }
        if (err) continue;                                            // keep finishing: no context is left with a pending call
        if (off + w + (uint64_t)tl > cap) { err = ZZ_E_NOSPACE; errmsg = "destination too small for the compressed stream"; continue; }
        if (i > 0 && w) {
            if (hipSetDevice(c->device) != hipSuccess ||
                hipMemcpyPeerAsync(dst + off, c0->device, c->stage_out, c->device, w, c->s_enc) != hipSuccess) {
                err = ZZ_E_HIP; errmsg = "hipMemcpyPeerAsync failed"; continue;
            }
            if (peer_prepare(c0->device, c->device) == 1) tl_pulls_p2p++; else tl_pulls_staged++;
        }
        acc = format == ZZ_ZLIB ? adler_combine(acc, part, n[i]) : format == ZZ_GZIP ? crc32_combine(acc, part, n[i]) : 0u;
        off += w;
        total_in += n[i];
    }
    for (int i = 1; i < nshards; ++i) {                               // the pulls have landed
        (void)hipSetDevice(ctxs[i]->device);
        if (hipStreamSynchronize(ctxs[i]->s_enc) != hipSuccess && !err) { err = ZZ_E_HIP; errmsg = "hipStreamSynchronize failed after the peer copies"; }
    }
    if (err) { set_err(errmsg); return err; }
    HIPCHK(hipSetDevice(c0->device));
    uint8_t hdr[10], trl[8];
    const int hn = zz_header(format, hdr), tn = zz_trailer(format, acc, total_in, trl);
    if (hn) HIPCHK(hipMemcpy(dst, hdr, (size_t)hn, hipMemcpyHostToDevice));
    if (tn) HIPCHK(hipMemcpy(dst + off, trl, (size_t)tn, hipMemcpyHostToDevice));
    *out_len = off + (uint64_t)tn;
    return ZZ_OK;
}

// ---- container pieces --------------------------------------------------------------------------------
extern "C" int zz_header(int format, uint8_t out[10])
{
    static const uint8_t gz[10] = { 0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xFF };   // zzflate.cpp:28
    if (format == ZZ_ZLIB) { out[0] = 0x78; out[1] = 0x01; return 2; }          // zzflate.cpp:30-36
    if (format == ZZ_GZIP) { memcpy(out, gz, 10); return 10; }
    return 0;
}
extern "C" int zz_trailer(int format, uint32_t v, uint64_t n, uint8_t out[8])   // zzflate.cpp:170-192
{
    if (format == ZZ_ZLIB) { out[0] = v >> 24; out[1] = v >> 16; out[2] = v >> 8; out[3] = v; return 4; }
    if (format == ZZ_GZIP) {
        uint32_t l = (uint32_t)n;
        for (int i = 0; i < 4; ++i) { out[i] = (uint8_t)(v >> (8 * i)); out[4 + i] = (uint8_t)(l >> (8 * i)); }
        return 8;
    }
    return 0;
}

// ---- host checksum utilities (adler.cpp / crc.cpp API; not on the encode path) ----------------------------
extern "C" uint32_t zz_adler32(uint32_t start, const uint8_t* p, uint64_t n)
{
    uint64_t a = start & 0xFFFF, b = start >> 16;
    while (n) {
        uint64_t k = n < 5552 ? n : 5552;
        for (uint64_t i = 0; i < k; ++i) { a += p[i]; b += a; }
        a %= ZZ_ADLER_MOD; b %= ZZ_ADLER_MOD;
        p += k; n -= k;
    }
    return (uint32_t)((b << 16) | a);
}
extern "C" uint32_t zz_adler32_combine(uint32_t first, uint32_t second, uint64_t len2) { return adler_combine(first, second, len2); }
extern "C" uint32_t zz_crc32(const uint8_t* p, uint64_t n, uint32_t start)
{
    static uint32_t tab[256];
    static std::once_flag once;
    std::call_once(once, [] {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int j = 0; j < 8; ++j) c = (c >> 1) ^ ((c & 1u) * ZZ_CRC_POLY);
            tab[i] = c;
        }
    });
    uint32_t c = ~start;
    for (uint64_t i = 0; i < n; ++i) c = (c >> 8) ^ tab[(c & 0xFF) ^ p[i]];
    return ~c;
}
extern "C" uint32_t zz_crc32_combine(uint32_t c1, uint32_t c2, uint64_t len2) { return crc32_combine(c1, c2, len2); }

// ---- host-buffer entry points: contexts, devices -----------------------------------------------------------------
// The reference's entry points are re-entrant and, with threaded=true, fan the input out over every core of the
// machine (zzflate.cpp:97-132). Here a call borrows one context per device it uses from a small pool (created on
// demand, at most two per device, so concurrent callers overlap without unbounded memory) and fans the input's slabs
// out over every visible GPU. Which devices: env ZZFLATE_DEVICES ("0,1,2" -- an index may repeat, which gives that GPU
// several pipelines -- or "all"), else env ZZFLATE_DEVICE (one index), else all of them. No lock is held while
// kernels run or callbacks are invoked, so a callback may call back into the library.
static std::mutex g_mu;
static std::condition_variable g_cv;
static uint32_t g_packet = 0;
struct pool_entry { zz_ctx* c; bool busy; };
// contexts the calling THREAD holds through host calls further up its stack (a callback that calls back into the
// library): such a call must never wait for the pool -- the contexts it would wait for may be its own, or belong to
// another thread whose callback is waiting the same way -- so it gets a temporary context beyond the cap instead.
static thread_local int tl_leases = 0;
static std::vector<pool_entry> g_pool;
static std::vector<int> g_devices;
#define ZZ_POOL_PER_DEVICE 2

extern "C" uint32_t zz_get_packet_size(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_packet == 0) {
        const char* e = getenv("ZZFLATE_PACKET_SIZE");
        long v = e ? atol(e) : 0;
        g_packet = (v >= 1 && v <= (long)ZZ_MAX_PACKET_SIZE) ? (uint32_t)v : ZZ_DEFAULT_PACKET;
    }
    return g_packet;
}
extern "C" int zz_set_packet_size(uint32_t P)
{
    if (P == 0 || P > ZZ_MAX_PACKET_SIZE) { set_err("packet size must be 1..32768"); return ZZ_E_ARG; }
    std::lock_guard<std::mutex> lk(g_mu);
    g_packet = P;
    return ZZ_OK;
}

// the device list of the host entry points (read once)
static int host_devices(std::vector<int>& out)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_devices.empty()) {
        int count = 0;
        hipError_t e = hipGetDeviceCount(&count);
        if (e != hipSuccess || count == 0) {
            set_err(std::string("no HIP device available (hipGetDeviceCount: ") + hipGetErrorString(e) + ", count " +
                    std::to_string(count) + "): this library has no CPU encode path");
            return ZZ_E_HIP;
        }
        const char* list = getenv("ZZFLATE_DEVICES");
        const char* one = getenv("ZZFLATE_DEVICE");
        std::vector<int> v;
        if (list && *list && strcmp(list, "all") != 0) {
            for (const char* q = list; *q;) {
                char* endp = nullptr;
                const long d = strtol(q, &endp, 10);
                if (endp == q) break;
                if (d < 0 || d >= count) { set_err("ZZFLATE_DEVICES names a device that does not exist"); return ZZ_E_ARG; }
                v.push_back((int)d);
                q = *endp == ',' ? endp + 1 : endp;
            }
        } else if (!list && one && *one) {
            const int d = atoi(one);
            if (d < 0 || d >= count) { set_err("ZZFLATE_DEVICE names a device that does not exist"); return ZZ_E_ARG; }
            v.push_back(d);
        }
        if (v.empty()) for (int d = 0; d < count; ++d) v.push_back(d);
        g_devices = v;
    }
    out = g_devices;
    return ZZ_OK;
}
// test hook (not in the public header): forget the device list so that the environment is read again
extern "C" void zz_debug_reset_devices(void) { std::lock_guard<std::mutex> lk(g_mu); g_devices.clear(); }

// `held`: contexts of this device the calling host call already holds (ZZFLATE_DEVICES may name a device several
// times). A call only ever waits for its first context of a device, devices are taken in one global order, and a
// `nested` call (its thread holds contexts through calls further up its stack) never waits at all: concurrent and
// nested calls cannot block each other for good.
// *temp = the context was created beyond the cap for a nested call and is destroyed on release.
static int pool_acquire(int device, int held, bool nested, zz_ctx** out, bool* temp)
{
    *temp = false;
    std::unique_lock<std::mutex> lk(g_mu);
    for (;;) {
        int have = 0;
        static const uint32_t warm = [] {               // env ZZFLATE_WARM_WINDOW: warm window of the host entry points
            const char* e = getenv("ZZFLATE_WARM_WINDOW");
            const long v = e ? atol(e) : 0;
            return (uint32_t)(v < 0 ? 0 : v > 32768 ? 32768 : v);
        }();
        static const bool ext = [] { const char* e = getenv("ZZFLATE_EXTENDED_LEVELS"); return e && atoi(e) != 0; }();
        if ((warm || ext) && !lds_order_ok(device)) {   // (the same refusal as zz_ctx_set_warm_window / zz_ctx_set_extended_levels)
            set_err("ZZFLATE_WARM_WINDOW / ZZFLATE_EXTENDED_LEVELS refused: this device's LDS does not serve equal addresses in lane order");
            return ZZ_E_UNSUPPORTED;
        }
        for (auto& e : g_pool)
            if (e.c->device == device) {
                if (!e.busy) { e.busy = true; e.c->warm = warm; e.c->extended = ext; *out = e.c; return ZZ_OK; }
                have++;
            }
        if (have < ZZ_POOL_PER_DEVICE + held || nested) {
            zz_ctx* c = nullptr;
            int rc = zz_ctx_create(device, &c);
            if (rc) return rc;
            c->warm = warm; c->extended = ext;
            if (have < ZZ_POOL_PER_DEVICE + held) g_pool.push_back({ c, true });
            else *temp = true;
            *out = c;
            return ZZ_OK;
        }
        g_cv.wait(lk);
    }
}
static void pool_release(zz_ctx* c)
{
    { std::lock_guard<std::mutex> lk(g_mu); for (auto& e : g_pool) if (e.c == c) e.busy = false; }
    g_cv.notify_all();
}
struct ctx_lease {                       // contexts borrowed for one host call
    std::vector<zz_ctx*> v;
    std::vector<bool> temp;
    const int outer = tl_leases;         // contexts held by host calls further up this thread's stack
    ~ctx_lease()
    {
        for (size_t i = 0; i < v.size(); ++i) { if (temp[i]) zz_ctx_destroy(v[i]); else pool_release(v[i]); }
        tl_leases -= (int)v.size();
    }
    int take(int device)
    {
        int held = 0;
        for (zz_ctx* h : v) held += h->device == device;
        zz_ctx* c = nullptr;
        bool t = false;
        int rc = pool_acquire(device, held, outer > 0, &c, &t);
        if (!rc) { v.push_back(c); temp.push_back(t); tl_leases++; }
        return rc;
    }
};

static int ensure_stage(zz_ctx* c, uint64_t in_bytes, uint64_t out_bytes)
{
    if (in_bytes > c->stage_in_cap) {
        (void)hipFree(c->stage_in); c->stage_in = nullptr; c->stage_in_cap = 0;
        HIPCHK(hipMalloc(&c->stage_in, in_bytes + 64));
        c->stage_in_cap = in_bytes + 64;
    }
    if (out_bytes > c->stage_out_cap) {
        (void)hipFree(c->stage_out); c->stage_out = nullptr; c->stage_out_cap = 0;
        HIPCHK(hipMalloc(&c->stage_out, out_bytes + 64));
        c->stage_out_cap = out_bytes + 64;
    }
    return ZZ_OK;
}

// ---- host-buffer entry points ------------------------------------------------------------------------------------
// Where the compressed bytes go: straight into the caller's destination (ZzFlateEncode, zzflate.cpp:225-242) or
// through the callback in library-owned chunks (ZzFlateEncodeToCallback, zzflate.cpp:197-222): at most 1,000,000 bytes
// each (outputbitstream.h:183); header and trailer are calls of their own, as in the reference.
struct host_sink {
    // destination form
    uint8_t* dest = nullptr; uint64_t cap = 0, pos = 0; bool overflow = false;
    // callback form
    zz_callback cb = nullptr; void* user = nullptr; std::vector<uint8_t> chunk;
    void raw(const uint8_t* p, uint64_t n)            // header / trailer / a chunk with the reference's own boundaries
    {
        if (cb) { flush(); cb(user, p, n); return; }          // also for 0 bytes (raw deflate: zzflate.cpp:205,221 call regardless)
        put(p, n);
    }
    void put(const uint8_t* p, uint64_t n)            // stream bytes
    {
        if (!cb) {
            if (overflow || n > cap - pos) { overflow = true; return; }
            memcpy_mt(dest + pos, p, n);
            pos += n;
            return;
        }
        while (n) {
            const uint64_t room = 1000000 - chunk.size();
            const uint64_t k = n < room ? n : room;
            chunk.insert(chunk.end(), p, p + k);
            p += k; n -= k;
            if (chunk.size() == 1000000) flush();
        }
    }
    void flush() { if (cb && !chunk.empty()) { cb(user, chunk.data(), chunk.size()); chunk.clear(); } }
    static void memcpy_mt(uint8_t* d, const uint8_t* s, uint64_t n)
    {
        // pageable <-> pinned copies of whole slabs: a few threads reach the memory bandwidth one thread cannot
        const uint64_t piece = 8ull << 20;
        if (n < 2 * piece) { memcpy(d, s, n); return; }
        const unsigned nt = (unsigned)((n + piece - 1) / piece < 8 ? (n + piece - 1) / piece : 8);
        const uint64_t per = ((n + nt - 1) / nt + 4095) & ~4095ull;
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; ++t) {
            const uint64_t o = (uint64_t)t * per;
            if (o >= n) break;
            th.emplace_back([=] { memcpy(d + o, s + o, n - o < per ? n - o : per); });
        }
        memcpy(d, s, per < n ? per : n);
        for (auto& x : th) x.join();
    }
};

static uint64_t slab_bytes(uint32_t P)
{
    uint64_t mib = 64;
    if (const char* e = getenv("ZZFLATE_SLAB_MIB")) { const long v = atol(e); if (v >= 1 && v <= 4096) mib = (uint64_t)v; }
    const uint64_t s = mib << 20;
    return s < P ? P : s / P * P;          // slabs are cut at packet boundaries
}

// bytes of input kept in front of a slab on the device: level >= 2 extends matches backward over at most 258 bytes
// (encoder.cpp:92-102,404, capped as D11 says) and compares eight at a time; a warm window reaches back 32768
#define ZZ_SLAB_HALO 36864ull

static int ensure_pipe(zz_ctx* c, uint64_t slab, uint64_t slab_bound)
{
    if (!c->s_in) {
        HIPCHK(hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&c->s_enc, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&c->s_out, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            HIPCHK(hipEventCreateWithFlags(&c->ev_in[i], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&c->ev_out[i], hipEventDisableTiming));
        }
    }
    const uint64_t in_bytes = ZZ_SLAB_HALO + slab;
    if (in_bytes > c->pin_in_cap) {
        for (int i = 0; i < 2; ++i) { (void)hipHostFree(c->pin_in[i]); c->pin_in[i] = nullptr; (void)hipFree(c->slab_in[i]); c->slab_in[i] = nullptr; }
        c->pin_in_cap = 0;
        for (int i = 0; i < 2; ++i) {
            HIPCHK(hipHostMalloc((void**)&c->pin_in[i], in_bytes, hipHostMallocDefault));
            HIPCHK(hipMalloc(&c->slab_in[i], in_bytes + 64));
        }
        c->pin_in_cap = in_bytes;
    }
    if (slab_bound > c->pin_out_cap) {
        for (int i = 0; i < 2; ++i) { (void)hipHostFree(c->pin_out[i]); c->pin_out[i] = nullptr; (void)hipFree(c->slab_out[i]); c->slab_out[i] = nullptr; }
        c->pin_out_cap = c->slab_out_cap = 0;
        for (int i = 0; i < 2; ++i) {
            HIPCHK(hipHostMalloc((void**)&c->pin_out[i], slab_bound, hipHostMallocDefault));
            HIPCHK(hipMalloc(&c->slab_out[i], slab_bound + 64));
        }
        c->pin_out_cap = c->slab_out_cap = slab_bound;
    }
    return ZZ_OK;
}

// Packet mode on a host buffer (SURVEY.md 8f.1 and the device analogue of the std::async fan-out, zzflate.cpp:97-155).
// The input is cut into slabs of whole packets; slab i belongs to device i mod D. Every device runs its own pipeline
// on its own host thread: the slab crosses PCIe (through a pinned buffer) while the device's previous slab is encoded
// as a shard of the stream (no container; checksum partial returned) and the one before that travels back. Only two
// slabs of input -- each with the 4 KiB in front of it, for level >= 2's backward match extension -- and two of output
// live on a device at any time, so the input may be far larger than HBM. The calling thread is the in-order join: it
// hands the slabs' bytes to the sink as they arrive and folds the checksum partials (adler.cpp:5-15 / GF(2) shifts).
static int encode_host_slabs(const std::vector<zz_ctx*>& ctxs, const uint8_t* src, uint64_t n, int format, int level, uint32_t P,
                             host_sink& sink)
{
    const uint64_t slab = slab_bytes(P);
    const uint64_t nslab = (n + slab - 1) / slab;
    const uint64_t sb = zz_bound(slab, ZZ_DEFLATE, level, P);
    const int D = (int)ctxs.size();
    const int ck = format == ZZ_ZLIB ? ZZ_ZLIB : format == ZZ_GZIP ? ZZ_GZIP : ZZ_DEFLATE;

    std::mutex mu;
    std::condition_variable cv;
    std::vector<uint64_t> out_len(nslab, 0);
    std::vector<uint32_t> part(nslab, 0);
    std::vector<uint8_t> ready(nslab, 0), consumed(nslab, 0);
    int err = 0;
    std::string errmsg;
    auto fail = [&](int rc) {
        std::lock_guard<std::mutex> lk(mu);
        if (!err) { err = rc; errmsg = g_err; }
        cv.notify_all();
    };

    auto worker = [&](int d) {
        zz_ctx* c = ctxs[d];
        auto run = [&]() -> int {
            HIPCHK(hipSetDevice(c->device));
            int rc = ensure_pipe(c, slab, sb);
            if (rc) return rc;
            auto stage_in = [&](uint64_t j) -> int {                       // j: this device's j-th slab
                const uint64_t i = (uint64_t)d + j * D, off = i * slab, len = n - off < slab ? n - off : slab;
                const uint64_t h = off < ZZ_SLAB_HALO ? off : ZZ_SLAB_HALO;
                const int b = (int)(j & 1);
                // pin_in[b] / slab_in[b] were last used by slab j-2, whose encode has returned
                host_sink::memcpy_mt(c->pin_in[b] + ZZ_SLAB_HALO - h, src + off - h, h + len);
                HIPCHK(hipMemcpyAsync(c->slab_in[b] + ZZ_SLAB_HALO - h, c->pin_in[b] + ZZ_SLAB_HALO - h, h + len, hipMemcpyHostToDevice, c->s_in));
                HIPCHK(hipEventRecord(c->ev_in[b], c->s_in));
                return ZZ_OK;
            };
            const uint64_t mine = nslab > (uint64_t)d ? (nslab - d + D - 1) / D : 0;
            if (mine == 0) return ZZ_OK;
            rc = stage_in(0);
            if (rc) return rc;
            for (uint64_t j = 0; j < mine; ++j) {
                const uint64_t i = (uint64_t)d + j * D, off = i * slab, len = n - off < slab ? n - off : slab;
                const uint64_t h = off < ZZ_SLAB_HALO ? off : ZZ_SLAB_HALO;
                const int b = (int)(j & 1);
                if (j + 1 < mine) { rc = stage_in(j + 1); if (rc) return rc; }     // the next slab's copy runs under this slab's encode
                HIPCHK(hipStreamWaitEvent(c->s_enc, c->ev_in[b], 0));
                if (j >= 2) {
                    // the output buffers of slab j-2: its copy back must be over, and the join must have taken the bytes
                    HIPCHK(hipStreamWaitEvent(c->s_enc, c->ev_out[b], 0));
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return err || consumed[i - 2 * (uint64_t)D]; });
                    if (err) return ZZ_OK;
                }
                uint64_t w = 0; uint32_t pc = 0;
                rc = zz_encode_shard_device(c, c->slab_in[b] + ZZ_SLAB_HALO, len, h, i + 1 == nslab, c->slab_out[b], c->slab_out_cap, &w, &pc,
                                            ck, level, P, (void*)c->s_enc);     // returns when the slab is encoded
                if (rc) return rc;
                HIPCHK(hipMemcpyAsync(c->pin_out[b], c->slab_out[b], w, hipMemcpyDeviceToHost, c->s_out));
                HIPCHK(hipEventRecord(c->ev_out[b], c->s_out));              // the join waits for it; this thread moves on
                {
                    std::lock_guard<std::mutex> lk(mu);
                    out_len[i] = w; part[i] = pc; ready[i] = 1;
                }
                cv.notify_all();
                { std::lock_guard<std::mutex> lk(mu); if (err) return ZZ_OK; }
            }
            return ZZ_OK;
        };
        const int rc = run();
        if (rc) fail(rc);
    };

    std::vector<std::thread> threads;
    for (int d = 0; d < D; ++d) threads.emplace_back(worker, d);
    uint8_t hdr[10], trl[8];
    uint32_t acc = format == ZZ_ZLIB ? 1u : 0u;          // running checksum of everything in front of the slab
    for (uint64_t i = 0; i < nslab; ++i) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return err || ready[i]; });
            if (err) break;
        }
        if (i == 0) sink.raw(hdr, (uint64_t)zz_header(format, hdr));    // nothing is delivered before a slab has encoded
        const uint64_t off = i * slab, len = n - off < slab ? n - off : slab;
        zz_ctx* c = ctxs[i % D];
        const int b = (int)((i / D) & 1);
        if (hipEventSynchronize(c->ev_out[b]) != hipSuccess) { set_err("hipEventSynchronize failed in the slab join"); fail(ZZ_E_HIP); break; }
        sink.put(c->pin_out[b], out_len[i]);
        acc = format == ZZ_ZLIB ? adler_combine(acc, part[i], len) : format == ZZ_GZIP ? crc32_combine(acc, part[i], len) : 0u;
        { std::lock_guard<std::mutex> lk(mu); consumed[i] = 1; }
        cv.notify_all();
    }
    for (auto& t : threads) t.join();
    if (err) { set_err(errmsg); return err; }
    sink.raw(trl, (uint64_t)zz_trailer(format, acc, n, trl));
    sink.flush();
    return ZZ_OK;
}

// one slab or less: one copy in, one call, one copy out
static int encode_host_simple(zz_ctx* c, const uint8_t* src, uint64_t n, int format, int level, uint32_t P, host_sink& sink)
{
    const uint64_t bound = zz_bound(n, format, level, P);
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure_stage(c, n, bound);
    if (rc) return rc;
    if (n) HIPCHK(hipMemcpy(c->stage_in, src, n, hipMemcpyHostToDevice));
    uint64_t total = 0;
    rc = zz_encode_device(c, c->stage_in, n, c->stage_out, bound, &total, format, level, P, nullptr);
    if (rc) return rc;
    std::vector<uint8_t> host(total);
    if (total) HIPCHK(hipMemcpy(host.data(), c->stage_out, total, hipMemcpyDeviceToHost));
    const uint64_t hl = header_len(format), tl = trailer_len(format);
    sink.raw(host.data(), hl);
    sink.put(host.data() + hl, total - hl - tl);
    sink.raw(host.data() + total - tl, tl);
    sink.flush();
    return ZZ_OK;
}

// threaded == false: the reference's single Encoder over the whole input (zzflate.cpp:84-95). Its output depends on
// where it goes: into the caller's buffer the level-1 block lengths follow from the buffer's size; through the callback
// they follow from the library's own 1,000,000-byte chunks, and the callback sees exactly those chunks.
static int encode_host_sequential(zz_ctx* c, const uint8_t* src, uint64_t n, int format, int level, host_sink& sink)
{
    const bool chunked = sink.cb != nullptr;
    const uint64_t hl = header_len(format), tl = trailer_len(format);
    // device buffer for the stream: level 1 never needs more than nine bits per byte plus ten per block
    uint64_t bound = zz_bound(n, format, level, ZZ_DEFAULT_PACKET) + n / 65536 + 256;
    uint64_t cap = bound;
    if (!chunked && level == 1 && sink.cap < cap) cap = sink.cap;      // the caller's capacity decides the block lengths
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure_stage(c, n, cap > bound ? cap : bound);
    if (rc) return rc;
    if (n) HIPCHK(hipMemcpy(c->stage_in, src, n, hipMemcpyHostToDevice));
    zz_result r;
    std::vector<uint64_t> chunks;
    rc = encode_stream(c, c->stage_in, n, c->stage_out, cap, format, level, chunked, chunked ? &chunks : nullptr, nullptr, &r);
    if (rc) return rc;
    const uint64_t total = r.total_bytes;
    std::vector<uint8_t> host(total);
    if (total) HIPCHK(hipMemcpy(host.data(), c->stage_out, total, hipMemcpyDeviceToHost));
    sink.raw(host.data(), hl);
    if (chunked) {
        uint64_t o = hl;
        for (uint64_t k : chunks) { sink.raw(host.data() + o, k); o += k; }   // one call per chunk (zzflate.cpp:207-215)
    } else {
        sink.put(host.data() + hl, total - hl - tl);
    }
    sink.raw(host.data() + total - tl, tl);
    sink.flush();
    return ZZ_OK;
}

// ZZFLATE_RANGES=<count>, threaded != 0, levels 0, 2, 3: the reference's own split (encode_ranges) instead of packets
static int encode_host_ranges(zz_ctx* c, const uint8_t* src, uint64_t n, int format, int level, uint32_t count, host_sink& sink)
{
    const uint64_t hl = header_len(format), tl = trailer_len(format);
    const uint64_t bound = n + (n / 65535 + 2 * (uint64_t)count + 2) * 5 + (n / 400000 + 2 * (uint64_t)count + 2) * 8 + 6ull * count + 64 + hl + tl;
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure_stage(c, n, bound);
    if (rc) return rc;
    if (n) HIPCHK(hipMemcpy(c->stage_in, src, n, hipMemcpyHostToDevice));
    zz_result r;
    rc = encode_ranges(c, c->stage_in, n, c->stage_out, bound, format, level, count, nullptr, &r);
    if (rc) return rc;
    const uint64_t total = r.total_bytes;
    std::vector<uint8_t> host(total);
    if (total) HIPCHK(hipMemcpy(host.data(), c->stage_out, total, hipMemcpyDeviceToHost));
    sink.raw(host.data(), hl);
    sink.put(host.data() + hl, total - hl - tl);
    sink.raw(host.data() + total - tl, tl);
    sink.flush();
    return ZZ_OK;
}

static int encode_host(const uint8_t* src, uint64_t n, const zz_config* cfg, host_sink& sink)
{
    const int level = cfg->level;
    static const bool ext = [] { const char* e = getenv("ZZFLATE_EXTENDED_LEVELS"); return e && atoi(e) != 0; }();
    if (level < 0 || level > (ext ? 6 : 3)) { set_err("level must be 0..3"); return ZZ_E_LEVEL; }
    if (level > 3 && !cfg->threaded) { set_err("levels 4..6 exist in packet mode only (threaded != 0)"); return ZZ_E_LEVEL; }
    if (!src && n) { set_err("null source"); return ZZ_E_ARG; }
    std::vector<int> devs;
    int rc = host_devices(devs);
    if (rc) return rc;
    const uint32_t P = zz_get_packet_size();
    int format = cfg->format;
    if (format < 0 || format > 2) format = ZZ_DEFLATE;
    ctx_lease lease;
    if (!cfg->threaded && n > 0) {
        rc = lease.take(devs[0]);
        if (rc) return rc;
        return encode_host_sequential(lease.v[0], src, n, format, level, sink);
    }
    {
        const char* e = getenv("ZZFLATE_RANGES");                         // (read per call: tests switch it)
        const int count = e ? atoi(e) : 0;
        // (only where the split's preconditions hold -- count <= 4096, n < 2 GiB, every range inside the input; otherwise the
        // call is served in packet mode like any other: the environment switch must not turn working calls into errors)
        if (count > 0 && count <= 4096 && n > 0 && n < (1ull << 31) && level != 1 && level <= 3 &&
            (n < 100ull * (uint64_t)count || ranges_split_ok(n, (uint32_t)count))) {
            rc = lease.take(devs[0]);
            if (rc) return rc;
            return encode_host_ranges(lease.v[0], src, n, format, level, (uint32_t)count, sink);
        }
    }
    const uint64_t slab = slab_bytes(P);
    const uint64_t nslab = (n + slab - 1) / slab;
    if (nslab <= 1) {
        rc = lease.take(devs[0]);
        if (rc) return rc;
        return encode_host_simple(lease.v[0], src, n, format, level, P, sink);
    }
    const size_t D = devs.size() < nslab ? devs.size() : (size_t)nslab;
    for (size_t d = 0; d < D; ++d) { rc = lease.take(devs[d]); if (rc) return rc; }
    return encode_host_slabs(lease.v, src, n, format, level, P, sink);
}

extern "C" int zz_encode(uint8_t* dest, uint64_t* dest_len, const uint8_t* src, uint64_t n, const zz_config* cfg)
{
    if (!dest_len) { set_err("null dest_len"); return ZZ_E_ARG; }
    const uint64_t cap = *dest_len;
    *dest_len = ~0ull;
    if (!cfg || !dest) { set_err("null argument"); return ZZ_E_ARG; }
    if (cap < (uint64_t)header_len(cfg->format)) { set_err("destination smaller than the container header"); return ZZ_E_NOSPACE; }
    host_sink sink;
    sink.dest = dest; sink.cap = cap;
    const int rc = encode_host(src, n, cfg, sink);
    if (rc) return rc;
    if (sink.overflow) { set_err("destination too small for the compressed stream"); return ZZ_E_NOSPACE; }
    *dest_len = sink.pos;
    return ZZ_OK;
}

extern "C" int zz_encode_callback(const uint8_t* src, uint64_t n, const zz_config* cfg, zz_callback cb, void* user)
{
    if (!cfg || !cb) { set_err("null argument"); return ZZ_E_ARG; }
    host_sink sink;
    sink.cb = cb; sink.user = user;
    sink.chunk.reserve(1000000);
    return encode_host(src, n, cfg, sink);
}

// diagnostic (not in the public header): device bytes the host entry points' contexts hold for input/output staging
extern "C" uint64_t zz_debug_host_staging_bytes(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    uint64_t t = 0;
    for (auto& e : g_pool) t += e.c->stage_in_cap + e.c->stage_out_cap + 2 * (e.c->pin_in_cap ? e.c->pin_in_cap + 64 : 0) + 2 * (e.c->slab_out_cap ? e.c->slab_out_cap + 64 : 0);
    return t;
}

// ---- synthetic inputs -----------------------------------------------------------------------------------------
__global__ void k_generate(int kind, uint64_t seed, uint64_t first_block, uint8_t* buf, uint64_t n)
{
    uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t off = b * ZZ_GEN_BLOCK;
    if (off >= n) return;
    uint32_t cap = (uint32_t)(n - off < ZZ_GEN_BLOCK ? n - off : ZZ_GEN_BLOCK);
    zzgen::gen_block(kind, seed, first_block + b, buf + off, cap);
}

extern "C" int zz_generate_device(zz_ctx* c, int kind, uint64_t seed, uint64_t first_byte, void* d_buf, uint64_t n,
                                  void* hip_stream)
{
    if (!c || !d_buf) { set_err("null argument"); return ZZ_E_ARG; }
    if (first_byte % ZZ_GEN_BLOCK) { set_err("first_byte must be a multiple of 65536"); return ZZ_E_ARG; }
    if (n == 0) return ZZ_OK;
    HIPCHK(hipSetDevice(c->device));
    uint64_t nb = (n + ZZ_GEN_BLOCK - 1) / ZZ_GEN_BLOCK;
    hipLaunchKernelGGL(k_generate, dim3((uint32_t)((nb + 63) / 64)), dim3(64), 0, (hipStream_t)hip_stream, kind, seed,
                       first_byte / ZZ_GEN_BLOCK, (uint8_t*)d_buf, n);
    HIPCHK(hipGetLastError());
    return ZZ_OK;
}
extern "C" int zz_generate_host(int kind, uint64_t seed, uint64_t first_byte, uint8_t* buf, uint64_t n)
{
    if (!buf) { set_err("null argument"); return ZZ_E_ARG; }
    if (first_byte % ZZ_GEN_BLOCK) { set_err("first_byte must be a multiple of 65536"); return ZZ_E_ARG; }
    for (uint64_t off = 0; off < n; off += ZZ_GEN_BLOCK) {
        uint32_t cap = (uint32_t)(n - off < ZZ_GEN_BLOCK ? n - off : ZZ_GEN_BLOCK);
        zzgen::gen_block(kind, seed, (first_byte + off) / ZZ_GEN_BLOCK, buf + off, cap);
    }
    return ZZ_OK;
}
