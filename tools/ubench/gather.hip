// Micro-benchmark: what does one vector-memory wave-instruction cost a CU on gfx950, by kind?
//   stream4   : coalesced global_load_dwordx4 (16 B / lane) from a large buffer
//   gather1/2 : per-lane random global_load_dword / dwordx2 from a 2.4 MB table (L2 resident), the texel gather of integrate
//   gatherloc : dwordx2 gather whose lanes hit neighbouring texels (lane i -> base + 3 * i): the realistic pattern
//   lds2      : ds_read_b64 random gather from a 32 KB LDS window
// Each kernel issues ITER instructions per wave with 8 independent chains; prints ns per wave-instruction per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITER = 256;
__device__ __forceinline__ unsigned lcg(unsigned s) { return s * 1664525u + 1013904223u; }

template <int W>  // W = 1: dword, 2: dwordx2
__global__ __launch_bounds__(256) void gather_kernel(const unsigned *__restrict__ tab, unsigned n_tex, int local, unsigned *out) {
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    unsigned acc = 0;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < ITER; it += 4) {
        unsigned idx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s = lcg(s);
            unsigned base = local ? (__builtin_amdgcn_readfirstlane(s) >> 8) % (n_tex - 400) + 3 * lane + (k * 640 % 1200) : (s >> 8) % n_tex;
            idx[k] = base;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (W == 1) acc += tab[2 * (size_t)idx[k]];
            else { uint2 v = *reinterpret_cast<const uint2 *>(tab + 2 * (size_t)idx[k]); acc += v.x ^ v.y; }
        }
    }
    if (acc == 0x12345) out[0] = acc;
}

__global__ __launch_bounds__(256) void stream_kernel(const uint4 *__restrict__ buf, size_t n16, unsigned *out) {
    unsigned acc = 0;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (int it = 0; it < ITER; ++it, i += stride) {
        if (i >= n16) i -= n16;
        uint4 v = buf[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345) out[0] = acc;
}

__global__ __launch_bounds__(256) void lds_kernel(const unsigned *__restrict__ tab, int local, unsigned *out) {
    __shared__ uint2 win[4096];  // 32 KB
    for (int i = threadIdx.x; i < 4096; i += 256) win[i] = make_uint2(tab[2 * i], tab[2 * i + 1]);
    __syncthreads();
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    unsigned acc = 0;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < ITER * 4; it += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s = lcg(s);
            unsigned idx = local ? ((__builtin_amdgcn_readfirstlane(s) >> 8) % 3500 + 3 * lane + k * 97) & 4095 : (s >> 8) & 4095;
            uint2 v = win[idx];
            acc += v.x ^ v.y;
        }
    }
    if (acc == 0x12345) out[0] = acc;
}

template <typename F>
static float time_ms(F launch) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / 5;
}

int main() {
    const unsigned n_tex = 640 * 480;
    std::vector<unsigned> h(2 * n_tex);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u);
    unsigned *tab, *out; uint4 *big;
    const size_t big_bytes = (size_t)1 << 31;
    CK(hipMalloc(&tab, h.size() * 4)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&big, big_bytes));
    CK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(big, 1, big_bytes));
    const int cus = 256;
    for (int wg_per_cu : {2, 4, 8}) {
        const int blocks = cus * wg_per_cu;
        const double instr_per_cu = (double)wg_per_cu * 4 * ITER;  // wave-instructions per CU
        auto rep = [&](const char *name, float ms, double per_cu, double bytes) {
            printf("%-10s wg/cu %d: %8.3f ms  %7.1f ns per wave-instr per CU (%5.1f cycles @2.4GHz)  %7.1f GB/s\n", name, wg_per_cu, ms,
                   ms * 1e6 / per_cu, ms * 1e6 / per_cu * 2.4, bytes / ms / 1e6);
        };
        float ms;
        ms = time_ms([&] { hipLaunchKernelGGL(stream_kernel, dim3(blocks), dim3(256), 0, 0, big, big_bytes / 16, out); });
        rep("stream4", ms, instr_per_cu, (double)blocks * 256 * ITER * 16);
        ms = time_ms([&] { hipLaunchKernelGGL(gather_kernel<1>, dim3(blocks), dim3(256), 0, 0, tab, n_tex, 0, out); });
        rep("gather1", ms, instr_per_cu, (double)blocks * 256 * ITER * 4);
        ms = time_ms([&] { hipLaunchKernelGGL(gather_kernel<2>, dim3(blocks), dim3(256), 0, 0, tab, n_tex, 0, out); });
        rep("gather2", ms, instr_per_cu, (double)blocks * 256 * ITER * 8);
        ms = time_ms([&] { hipLaunchKernelGGL(gather_kernel<1>, dim3(blocks), dim3(256), 0, 0, tab, n_tex, 1, out); });
        rep("gather1loc", ms, instr_per_cu, (double)blocks * 256 * ITER * 4);
        ms = time_ms([&] { hipLaunchKernelGGL(gather_kernel<2>, dim3(blocks), dim3(256), 0, 0, tab, n_tex, 1, out); });
        rep("gather2loc", ms, instr_per_cu, (double)blocks * 256 * ITER * 8);
        ms = time_ms([&] { hipLaunchKernelGGL(lds_kernel, dim3(blocks), dim3(256), 0, 0, tab, 0, out); });
        rep("lds2", ms, instr_per_cu * 4, (double)blocks * 256 * ITER * 4 * 8);
        ms = time_ms([&] { hipLaunchKernelGGL(lds_kernel, dim3(blocks), dim3(256), 0, 0, tab, 1, out); });
        rep("lds2loc", ms, instr_per_cu * 4, (double)blocks * 256 * ITER * 4 * 8);
    }
    return 0;
}
