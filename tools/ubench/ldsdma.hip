// Micro-benchmark: how fast can one CU pull GEMM operand tiles from L2 into LDS / registers on gfx950?
// 512-thread workgroups (8 waves), one per CU (LDS 128 KiB), each step stages 64 KiB (a 256 x 64 bf16 A tile + a 256 x 64 W tile)
// the way csrc/vit.hip and csrc/conv.hip do (8 wave-instructions per wave and step), two-stage ring, one barrier per step.
//   dma128   : global_load_lds_dwordx4, a wave-instruction = 8 rows x 128 B (row stride ld bytes)
//   dma1k    : global_load_lds_dwordx4, a wave-instruction = 1 KiB contiguous
//   reg128   : global_load_dwordx4 to registers (no LDS write), 8 rows x 128 B
//   reg128w  : the same + ds_write_b128 into LDS
// Operands come from a buffer of `ws` MiB read round-robin (4 MiB = an XCD's L2 share of weights; 64 MiB = activations).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int STEPS = 96;

template <int MODE>
__global__ __launch_bounds__(512, 1) void fill_kernel(const unsigned char *__restrict__ src, size_t span, int ld, unsigned *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // 2 x 64 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // tile base of this workgroup: rows of `ld` bytes; 512 rows (A + W) of 128 B per step
    size_t base = ((size_t)blockIdx.x * 9973 * 4096) % span;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int s = 0; s < STEPS; ++s) {
        unsigned char *st = lds + (s & 1) * 65536;
        const size_t kofs = (size_t)(s % 36) * 128;  // walk along K like a GEMM
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int grp = wave + 8 * j;  // 64 groups of 8 rows
            size_t off;
            if (MODE == 1) off = (base + (size_t)(s * 64 + grp) * 1024 + lane * 16) % span;               // 1 KiB contiguous
            else off = (base + (size_t)(grp * 8 + (lane >> 3)) * ld + kofs + (lane & 7) * 16) % span;      // 8 rows x 128 B
            const unsigned char *g = src + (off & ~(size_t)15);
            if (MODE <= 1) {
                __builtin_amdgcn_global_load_lds((const void *)g, (__attribute__((address_space(3))) void *)(st + grp * 1024), 16, 0, 0);
            } else {
                const uint4 v = *reinterpret_cast<const uint4 *>(g);
                if (MODE == 3) *reinterpret_cast<uint4 *>(st + grp * 1024 + lane * 16) = v;
                else { acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = acc.x;
    if (lds[threadIdx.x] == 0x7f && threadIdx.x == 511) out[1] = 1;
}

template <int MODE>
static void run(const char *name, const unsigned char *src, size_t span, int ld, unsigned *out, int wgs) {
    CK(hipFuncSetAttribute((const void *)fill_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(fill_kernel<MODE>, dim3(wgs), dim3(512), 131072, 0, src, span, ld, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(fill_kernel<MODE>, dim3(wgs), dim3(512), 131072, 0, src, span, ld, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    const double bytes = (double)wgs * STEPS * 65536.0;
    printf("%-8s span %4zu MiB ld %5d: %7.3f ms  %6.1f GB/s per CU  %5.2f TB/s chip  (%5.0f cycles @2.4GHz per 64 KiB step)\n", name, span >> 20, ld, ms,
           bytes / wgs / ms / 1e6, bytes / ms / 1e9, ms * 1e-3 / STEPS * 2.4e9);
}

int main() {
    unsigned char *buf; unsigned *out;
    const size_t cap = (size_t)256 << 20;
    CK(hipMalloc(&buf, cap + 4096)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 1, cap + 4096));
    for (size_t span_mb : {4, 64, 256}) {
        const size_t span = span_mb << 20;
        for (int ld : {1536, 4608}) {
            run<0>("dma128", buf, span, ld, out, 256);
            run<1>("dma1k", buf, span, ld, out, 256);
            run<2>("reg128", buf, span, ld, out, 256);
            run<3>("reg128w", buf, span, ld, out, 256);
        }
    }
    return 0;
}
