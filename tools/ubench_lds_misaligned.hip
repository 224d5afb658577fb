// Probe: what does a byte-addressed (misaligned) LDS read cost on gfx950? The compiler emits ds_read_b64 / ds_read_b32 for align-1
// accesses, and the results are right (k_l6_matches with -DZZ_L6_WORDS4=2 is bit-exact) -- but level 6 lost 22 % with them
// (profiles/r05_ab_l6_unaligned_reads_and_packed_key.txt). Here: 64 lanes gather at pseudo-random offsets of a 64 KiB image, as
// the candidates of k_l6_matches do, with the offset's low bits forced to a given misalignment; throughput form (32 independent
// reads in flight per measurement), one wavefront and sixteen wavefronts per workgroup; cycles per read instruction from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ub tools/ubench_lds_misaligned.hip && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint64_t __attribute__((aligned(1))) u64u;
typedef uint32_t __attribute__((aligned(1))) u32u;
typedef __attribute__((address_space(3))) const uint8_t* lds_bytes;

// KIND 0: aligned ds_read_b64 (offset & ~7)        1: ds_read_b64 at offset & ~7 | 4 (four-byte aligned)
//      2: ds_read_b64 at any byte offset           3: ds_read2_b32 + ds_read_b32 at offset & ~3 (k_l6_matches' default: three dwords)
//      4: aligned ds_read_b32                      5: ds_read_b32 at any byte offset
template <int KIND>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, uint32_t seed, int slot)
{
    __shared__ __attribute__((aligned(16))) uint8_t img[65536 + 64];
    const int tid = threadIdx.x;
    for (int i = tid; i < (65536 + 64) / 4; i += blockDim.x) ((uint32_t*)img)[i] = i * 2654435761u;
    __syncthreads();
    const lds_bytes L = (lds_bytes)(const uint8_t*)img;
    unsigned long long t0, t1, acc = 0;
    uint32_t x = seed + tid * 747796405u;
    uint64_t sink = 0;
    for (int r = 0; r < 128; ++r) {
        uint32_t off[32];
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            x = x * 1664525u + 1013904223u;
            uint32_t o = (x >> 16) & 65535u;
            if (KIND == 0) o &= ~7u;
            else if (KIND == 1) o = (o & ~7u) | 4u;
            else if (KIND == 3 || KIND == 4) o &= ~3u;
            off[s] = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            if (KIND <= 2) sink ^= *(__attribute__((address_space(3))) const u64u*)(L + off[s]);
            else if (KIND == 3) {
                const __attribute__((address_space(3))) uint32_t* p = (const __attribute__((address_space(3))) uint32_t*)(L + off[s]);
                sink ^= ((uint64_t)p[1] << 32 | p[0]) + p[2];
            }
            else sink ^= *(__attribute__((address_space(3))) const u32u*)(L + off[s]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        acc += t1 - t0;
    }
    if (tid == 0) out[slot] = acc;
    if ((uint32_t)sink == 0x89abcdefu) out[63] = 1;
}

template <int KIND>
static void run(unsigned long long* d, const char* name, int& slot)
{
    for (int threads : { 64, 1024 }) {
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, d, 1u, slot);
        unsigned long long h = 0; hipMemcpy(&h, d + slot, 8, hipMemcpyDeviceToHost);
        printf("%-58s %2d wavefront(s): %7.1f ticks per read instruction of one wavefront, %6.2f per instruction of the workgroup\n", name, threads / 64,
               (double)h / (128.0 * 32.0), (double)h / (128.0 * 32.0) / (threads / 64));
        ++slot;
    }
}

// The reads the compiler will not emit at four-byte alignment (it splits them), forced by inline asm: eight in flight per block, four blocks
// per measurement. W = 64: ds_read_b64, 96: ds_read_b96, 128: ds_read_b128; ALIGN = the alignment the offsets are forced to (4, 8, 16).
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int W> struct vec_of;
template <> struct vec_of<64> { typedef u32x2 t; };
template <> struct vec_of<96> { typedef u32x3 t; };
template <> struct vec_of<128> { typedef u32x4 t; };
#define ZZ_RD8(INSN) asm volatile(INSN " %0, %8\n\t" INSN " %1, %9\n\t" INSN " %2, %10\n\t" INSN " %3, %11\n\t" INSN " %4, %12\n\t" INSN " %5, %13\n\t" \
                                  INSN " %6, %14\n\t" INSN " %7, %15\n\ts_waitcnt lgkmcnt(0)" \
                                  : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) \
                                  : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "memory")
template <int W, int ALIGN>
__global__ __launch_bounds__(1024) void ka(unsigned long long* out, uint32_t seed, int slot)
{
    __shared__ __attribute__((aligned(16))) uint8_t img[65536 + 64];
    const int tid = threadIdx.x;
    for (int i = tid; i < (65536 + 64) / 4; i += blockDim.x) ((uint32_t*)img)[i] = i * 2654435761u;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(lds_bytes)(const uint8_t*)img;
    unsigned long long t0, t1, acc = 0;
    uint32_t x = seed + tid * 747796405u, sink = 0;
    for (int r = 0; r < 128; ++r) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            uint32_t a[8];
            typename vec_of<W>::t v[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                x = x * 1664525u + 1013904223u;
                uint32_t o = ((x >> 16) & 65535u) & ~(uint32_t)(ALIGN - 1);
                if (ALIGN == 4) o |= 4u;                     // never better aligned than asked
                if (ALIGN == 8) o |= 8u;
                a[s] = base + o;
            }
            if (W == 64) ZZ_RD8("ds_read_b64"); else if (W == 96) ZZ_RD8("ds_read_b96"); else ZZ_RD8("ds_read_b128");
#pragma unroll
            for (int s = 0; s < 8; ++s) sink ^= v[s].x + v[s].y;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        acc += t1 - t0;
    }
    if (tid == 0) out[slot] = acc;
    if (sink == 0x89abcdefu) out[63] = 1;
}
template <int W, int ALIGN>
static void runa(unsigned long long* d, const char* name, int& slot)
{
    for (int threads : { 64, 1024 }) {
        hipLaunchKernelGGL((ka<W, ALIGN>), dim3(1), dim3(threads), 0, 0, d, 1u, slot);
        unsigned long long h = 0; hipMemcpy(&h, d + slot, 8, hipMemcpyDeviceToHost);
        printf("%-58s %2d wavefront(s): %7.1f ticks per read instruction of one wavefront, %6.2f per instruction of the workgroup\n", name, threads / 64,
               (double)h / (128.0 * 32.0), (double)h / (128.0 * 32.0) / (threads / 64));
        ++slot;
    }
}

int main()
{
    unsigned long long* d; hipMalloc(&d, 64 * 8); hipMemset(d, 0, 64 * 8);
    int slot = 0;
    run<0>(d, "ds_read_b64, eight-byte aligned", slot);
    run<1>(d, "eight bytes, four-byte aligned (compiles to ds_read2_b32)", slot);
    run<2>(d, "ds_read_b64, any byte offset", slot);
    run<3>(d, "ds_read2_b32 + ds_read_b32, four-byte aligned (3 dwords)", slot);
    run<4>(d, "ds_read_b32, aligned", slot);
    run<5>(d, "ds_read_b32, any byte offset", slot);
    printf("-- forced by inline asm, eight reads in flight (the address arithmetic is inside the timed region: compare these rows with each other) --\n");
    runa<64, 8>(d, "asm ds_read_b64, eight-byte aligned", slot);
    runa<64, 4>(d, "asm ds_read_b64, four-byte aligned", slot);
    runa<96, 16>(d, "asm ds_read_b96, sixteen-byte aligned", slot);
    runa<96, 4>(d, "asm ds_read_b96, four-byte aligned", slot);
    runa<128, 16>(d, "asm ds_read_b128, sixteen-byte aligned", slot);
    runa<128, 8>(d, "asm ds_read_b128, eight-byte aligned", slot);
    runa<128, 4>(d, "asm ds_read_b128, four-byte aligned", slot);
    printf("(s_memtime counts at a fixed rate; compare the rows with each other. Offsets are pseudo-random over 64 KiB: bank conflicts as in a gather)\n");
    return 0;
}
