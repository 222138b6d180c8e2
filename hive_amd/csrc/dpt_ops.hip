// Memory-bound glue of the DPT-Hybrid convolutional parts, fused for channels-last 16-bit tensors on gfx950:
//   hive_nhwc_group_norm   GroupNorm(32) [+ residual add] [+ ReLU] of the ResNetV2 stem / bottlenecks
//   hive_nhwc_upsample2x   bilinear x2, align_corners=True, of the RefineNet fusion blocks and the depth head
// (timm 0.5.4 `GroupNormAct`, isl-org/DPT `FeatureFusionBlock_custom` / `Interpolate`, reached from
// dpt.models.DPTDepthModel.forward -- /root/reference/hive/dataset_adaptors.py:1419).  The convolutions
// themselves stay with MIOpen this round; the second half of the depth head is dpt_head.hip.
//
// Both are HBM-bound: every lane moves 16 bytes (8 channels) per access, channels fastest.
//   group norm: 2 reads + 1 write of the tensor (statistics pass, apply pass) + 1 read of the residual;
//   upsample  : 1 read (mostly L2 hits, each input pixel feeds 4 outputs) + 1 write of 4x the input.
#include "hive_internal.hpp"

#include <algorithm>
#include <cstdlib>

typedef __bf16 bf16;

template <typename T>
struct Vec8 {
    T v[8];
};

template <typename T>
__device__ __forceinline__ void load8(const T *p, float (&f)[8]) {
    const uint4 raw = *reinterpret_cast<const uint4 *>(p);
    const T *t = reinterpret_cast<const T *>(&raw);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (float)t[j];
}

template <typename T>
__device__ __forceinline__ void store8(T *p, const float (&f)[8]) {
    Vec8<T> o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = (T)f[j];
    *reinterpret_cast<uint4 *>(p) = *reinterpret_cast<const uint4 *>(&o);
}
template <typename T>
__device__ __forceinline__ void store8_nt(T *p, const float (&f)[8]) {  // streaming store (the x2 upsampling's 4 GB output)
    Vec8<T> o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = (T)f[j];
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(*reinterpret_cast<const u4 *>(&o), reinterpret_cast<u4 *>(p));
}

// ------------------------------------------------------------------------------------------------
// GroupNorm statistics, stage 1: per (sample, slab of pixels, channel) sum and sum of squares.
// grid (slabs, N); 256 threads = (256 / VC) pixels x VC channel vectors, VC = C / 8 (a power of two <= 256).
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T *__restrict__ x, int HW, int C, int slabs, float *__restrict__ partial) {
    __shared__ float red[256 * 16];
    const int VC = C >> 3, PP = 256 / VC;
    const int n = blockIdx.y, slab = blockIdx.x;
    const int v = threadIdx.x % VC, pp = threadIdx.x / VC;
    const int per = (HW + slabs - 1) / slabs;
    const int p0 = slab * per, p1 = min(p0 + per, HW);
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const T *base = x + (size_t)n * HW * C + v * 8;
    // four loads in flight per thread (one per trip left the kernel latency-bound at 3.6 TB/s); the adds keep the pixel order
    int p = p0 + pp;
    for (; p + 3 * PP < p1; p += 4 * PP) {
        float f[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) load8(base + (size_t)(p + u * PP) * C, f[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s[j] += f[u][j];
                q[j] += f[u][j] * f[u][j];
            }
    }
    for (; p < p1; p += PP) {
        float f[8];
        load8(base + (size_t)p * C, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s[j] += f[j];
            q[j] += f[j] * f[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[threadIdx.x * 16 + j] = s[j];
        red[threadIdx.x * 16 + 8 + j] = q[j];
    }
    __syncthreads();
    // thread t < 2 * C: t -> (which = sum / sumsq, channel c); add the PP pixel lanes in a fixed order
    for (int t = threadIdx.x; t < 2 * C; t += 256) {
        const int which = t / C, c = t % C, vv = c >> 3, j = c & 7;
        float acc = 0.f;
        for (int r = 0; r < PP; ++r) acc += red[(r * VC + vv) * 16 + which * 8 + j];
        partial[(((size_t)n * slabs + slab) * 2 + which) * C + c] = acc;
    }
}

// stage 2: per (sample, group) mean and rstd from the slab partials, summed in a fixed order (deterministic):
// one 256-thread workgroup per (sample, group) -- threads stride over the slabs x channels-of-the-group partials (two
// dependent-load rounds instead of eight with one wave), fixed butterfly per wave, the four wave sums added in order
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float *__restrict__ partial, int C, int G, int slabs, int HW, float eps,
                                                          float *__restrict__ stats, int total) {
    __shared__ double red[8];
    const int i = blockIdx.x;  // i = n * G + g
    if (i >= total) return;
    const int n = i / G, g = i % G, cpg = C / G;
    double s = 0.0, q = 0.0;
    for (int e = threadIdx.x; e < slabs * cpg; e += 256) {
        const int sl = e / cpg, c = e % cpg;
        const float *ps = partial + (((size_t)n * slabs + sl) * 2) * C + g * cpg + c;
        s += (double)ps[0];
        q += (double)ps[C];
    }
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_xor(s, off);
        q += __shfl_xor(q, off);
    }
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = s;
        red[4 + (threadIdx.x >> 6)] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = ((red[0] + red[1]) + red[2]) + red[3];
        q = ((red[4] + red[5]) + red[6]) + red[7];
        const double cnt = (double)HW * cpg;
        const double mean = s / cnt;
        const double var = fmax(q / cnt - mean * mean, 0.0);
        stats[2 * i] = (float)mean;
        stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// stage 2 when the producing convolution's epilogue already summed its tiles (csrc/conv.hip, GN): partial[tile][h][sum, sq][C] with
// h = 0 for the rows of the tile's first image, 1 for the rows past that image's end (TM <= HW: two images at most).  Same
// reduction as above over the tiles that overlap sample n.
__global__ __launch_bounds__(256) void gn_finalize_tiles_kernel(const float *__restrict__ partial, int C, int G, int TM, int HW, float eps,
                                                                float *__restrict__ stats, int total) {
    __shared__ double red[8];
    const int i = blockIdx.x;  // i = n * G + g
    if (i >= total) return;
    const int n = i / G, g = i % G, cpg = C / G;
    const long long r0 = (long long)n * HW, r1 = r0 + HW - 1;
    const int t0 = (int)(r0 / TM), nt = (int)(r1 / TM) - t0 + 1;
    double s = 0.0, q = 0.0;
    for (int e = threadIdx.x; e < nt * cpg; e += 256) {
        const int t = t0 + e / cpg, c = e % cpg;
        const int h = ((long long)t * TM) / HW == n ? 0 : 1;  // is n the image of the tile's first row?
        const float *ps = partial + (((size_t)t * 2 + h) * 2) * C + g * cpg + c;
        s += (double)ps[0];
        q += (double)ps[C];
    }
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_xor(s, off);
        q += __shfl_xor(q, off);
    }
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = s;
        red[4 + (threadIdx.x >> 6)] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = ((red[0] + red[1]) + red[2]) + red[3];
        q = ((red[4] + red[5]) + red[6]) + red[7];
        const double cnt = (double)HW * cpg;
        const double mean = s / cnt;
        const double var = fmax(q / cnt - mean * mean, 0.0);
        stats[2 * i] = (float)mean;
        stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// stage 3: y = (x - mean) * rstd * gamma + beta  [+ residual]  [ReLU].  grid (blocks, N)
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T *__restrict__ x, const T *__restrict__ gamma, const T *__restrict__ beta,
                                                       const float *__restrict__ stats, const T *__restrict__ residual, T *__restrict__ out,
                                                       int HW, int C, int G, int relu) {
    const int VC = C >> 3, PP = 256 / VC;
    const int n = blockIdx.y;
    const int v = threadIdx.x % VC, pp = threadIdx.x / VC;
    const int cpg = C / G;
    float a[8], b[8];
    {
        float gm[8], bt[8];
        load8(gamma + v * 8, gm);
        load8(beta + v * 8, bt);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int g = (v * 8 + j) / cpg;
            const float mean = stats[2 * (n * G + g)], rstd = stats[2 * (n * G + g) + 1];
            a[j] = rstd * gm[j];
            b[j] = bt[j] - mean * a[j];
        }
    }
    const size_t off = (size_t)n * HW * C + v * 8;
    auto finish = [&](int p, float (&f)[8], const float (&r)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = f[j] * a[j] + b[j];
        if (residual) {
            // the reference rounds the normalised value to the tensor type before the add (separate ops)
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = (float)(T)f[j] + r[j];
        }
        if (relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j], 0.f);
        }
        store8(out + off + (size_t)p * C, f);
    };
    // four pixels per trip: the loads of all four are in flight together (one per trip: 3.5 TB/s, latency-bound)
    const int step = gridDim.x * PP;
    int p = blockIdx.x * PP + pp;
    for (; p + 3 * step < HW; p += 4 * step) {
        float f[4][8], r[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) load8(x + off + (size_t)(p + u * step) * C, f[u]);
        if (residual) {
#pragma unroll
            for (int u = 0; u < 4; ++u) load8(residual + off + (size_t)(p + u * step) * C, r[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) finish(p + u * step, f[u], r[u]);
    }
    for (; p < HW; p += step) {
        float f[8], r[8];
        load8(x + off + (size_t)p * C, f);
        if (residual) load8(residual + off + (size_t)p * C, r);
        finish(p, f, r);
    }
}

// ------------------------------------------------------------------------------------------------
// bilinear x2 upsampling, align_corners=True (PyTorch's formula, evaluated in float):
//   src = dst * (in - 1) / (out - 1);  i0 = floor(src), i1 = min(i0 + 1, in - 1), l1 = src - i0, l0 = 1 - l1
//   y = h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11)
// `bias` (optional, [C]): added to every input value on load and rounded to T first -- the bias pass of the
// convolution that produced `in`, folded in (x + b rounded to the tensor type, as the separate add rounds it).
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_kernel(const T *__restrict__ in, const T *__restrict__ bias, T *__restrict__ out, int N, int H,
                                                         int W, int C) {
    // A thread produces a 2 x 2 block of output pixels (8 channels) from the 3 x 3 input pixels it can touch: 9 loads per 4 stores
    // instead of 16 -- the kernel is bound by the L1 path of its gathers, not by HBM (a fill of its output alone runs twice as fast).
    // Source rows of output rows 2k, 2k + 1: y0(2k) = ya and y0(2k + 1) = ya + dy with dy in {0, 1} (the scale is below 1/2), so both
    // pairs (y0, y1 = min(y0 + 1, H - 1)) lie in r_i = min(ya + i, H - 1), i = 0..2; columns likewise.  Every output is evaluated with
    // the formula above on the same operands as the one-pixel form: bit-identical.
    // grid (x-blocks over W * C / 8, H, N)
    const int VC = C >> 3;
    const int OH = 2 * H, OW = 2 * W;
    const float sh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
    const float sw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
    const int k = blockIdx.y, n = blockIdx.z;
    const float fy0 = sh * (float)(2 * k), fy1 = sh * (float)(2 * k + 1);
    const int ya = (int)fy0, yb = (int)fy1;
    const bool dy = yb != ya;  // (uniform)
    const float hy1[2] = {fy0 - (float)ya, fy1 - (float)yb};
    const int r1 = min(ya + 1, H - 1), r2 = min(ya + 2, H - 1);
    float bb[8];
    for (int t = blockIdx.x * 256 + threadIdx.x; t < W * VC; t += gridDim.x * 256) {
        const int j = t / VC, v = t - j * VC;
        const float fx0 = sw * (float)(2 * j), fx1 = sw * (float)(2 * j + 1);
        const int xa = (int)fx0, xb = (int)fx1;
        const bool dx = xb != xa;
        const float wx1[2] = {fx0 - (float)xa, fx1 - (float)xb};
        const int c1 = min(xa + 1, W - 1), c2 = min(xa + 2, W - 1);
        const T *base = in + (size_t)n * H * W * C + v * 8;
        const int rows[3] = {ya, r1, r2}, cols[3] = {xa, c1, c2};
        float B[3][3][8];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int q = 0; q < 3; ++q) load8(base + ((size_t)rows[i] * W + cols[q]) * C, B[i][q]);
        if (bias) {
            load8(bias + v * 8, bb);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int e = 0; e < 8; ++e) B[i][q][e] = (float)(T)(B[i][q][e] + bb[e]);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const float h1 = hy1[a], h0 = 1.f - h1;
            const bool sy = a == 1 && dy;  // rows (sy, sy + 1) of the block
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float w1 = wx1[b], w0 = 1.f - w1;
                const bool sx = b == 1 && dx;
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t0 = sy ? B[1][0][e] : B[0][0][e], t1 = sy ? B[1][1][e] : B[0][1][e], t2 = sy ? B[1][2][e] : B[0][2][e];
                    const float u0 = sy ? B[2][0][e] : B[1][0][e], u1 = sy ? B[2][1][e] : B[1][1][e], u2 = sy ? B[2][2][e] : B[1][2][e];
                    const float v00 = sx ? t1 : t0, v01 = sx ? t2 : t1, v10 = sx ? u1 : u0, v11 = sx ? u2 : u1;
                    o[e] = h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11);
                }
                store8(out + ((((size_t)n * OH + 2 * k + a) * OW) + 2 * j + b) * C + v * 8, o);
            }
        }
    }
}

// The same operation through an LDS tile (round 3; used for C <= 512): a workgroup produces UP_TY x UP_TX output 2 x 2 blocks (8 x 16 output
// pixels, all channels) from the (UP_TY + 2) x (UP_TX + 2) input pixels they can touch.  The gather form above issues 9 loads per 4 stores
// (13 vector-memory instructions per 64 bytes of output: it is bound by the CU's texture-address path, 3.7 TB/s of useful bytes); here every
// input pixel is loaded ONCE per tile (0.47 loaded bytes per output byte instead of 2.25), the bias is added once per input element, and
// the four taps of an output come from LDS without selects.  Same operands, same formula: bit-identical (tests/test_vit_gpu.py).
constexpr int UP_TY = 4, UP_TX = 8;
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_lds_kernel(const T *__restrict__ in, const T *__restrict__ bias, T *__restrict__ out, int N, int H,
                                                             int W, int C) {
    extern __shared__ __attribute__((aligned(16))) unsigned char up_lds[];
    T *tile = reinterpret_cast<T *>(up_lds);  // [(UP_TY + 2)][(UP_TX + 2)][C]
    constexpr int WR = UP_TY + 2, WC = UP_TX + 2;
    const int VC = C >> 3;
    const int OH = 2 * H, OW = 2 * W;
    const float sh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
    const float sw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
    const int k0 = blockIdx.y * UP_TY, j0 = blockIdx.x * UP_TX, n = blockIdx.z;
    const int ry0 = (int)(sh * (float)(2 * k0)), cx0 = (int)(sw * (float)(2 * j0));  // first source row / column of the tile
    const T *src = in + (size_t)n * H * W * C;
    // stage the window: row l holds source row min(ry0 + l, H - 1), column likewise (the clamps of the formula)
    const int n_vec = WR * WC * VC;
    for (int base = 0; base < n_vec; base += 256 * 4) {
        uint4 raw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * 256 + (int)threadIdx.x;
            if (idx < n_vec) {
                const int v = idx % VC, px = idx / VC, lc = px % WC, lr = px / WC;
                const int gy = min(ry0 + lr, H - 1), gx = min(cx0 + lc, W - 1);
                raw[u] = *reinterpret_cast<const uint4 *>(src + ((size_t)gy * W + gx) * C + v * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * 256 + (int)threadIdx.x;
            if (idx < n_vec) {
                const int v = idx % VC;
                if (bias) {  // (x + b) rounded to T, as the separate bias pass rounds it
                    float f[8], bb[8];
                    const T *t = reinterpret_cast<const T *>(&raw[u]);
                    load8(bias + v * 8, bb);
                    T r[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = (float)t[e], r[e] = (T)(f[e] + bb[e]);
                    raw[u] = *reinterpret_cast<const uint4 *>(r);
                }
                *reinterpret_cast<uint4 *>(tile + (size_t)idx * 8) = raw[u];
            }
        }
    }
    __syncthreads();
    for (int item = threadIdx.x; item < UP_TY * UP_TX * VC; item += 256) {
        const int v = item % VC, pair = item / VC, jl = pair % UP_TX, kl = pair / UP_TX;
        const int k = k0 + kl, j = j0 + jl;
        if (k >= H || j >= W) continue;
        const float fy[2] = {sh * (float)(2 * k), sh * (float)(2 * k + 1)}, fx[2] = {sw * (float)(2 * j), sw * (float)(2 * j + 1)};
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int y0 = (int)fy[a];
            const float h1 = fy[a] - (float)y0, h0 = 1.f - h1;
            const int r0 = y0 - ry0;  // rows r0, r0 + 1 of the window: source rows y0, min(y0 + 1, H - 1)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int x0 = (int)fx[b];
                const float w1 = fx[b] - (float)x0, w0 = 1.f - w1;
                const int c0 = x0 - cx0;
                float v00[8], v01[8], v10[8], v11[8], o[8];
                load8(tile + ((size_t)(r0 * WC + c0) * VC + v) * 8, v00);
                load8(tile + ((size_t)(r0 * WC + c0 + 1) * VC + v) * 8, v01);
                load8(tile + ((size_t)((r0 + 1) * WC + c0) * VC + v) * 8, v10);
                load8(tile + ((size_t)((r0 + 1) * WC + c0 + 1) * VC + v) * 8, v11);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = h0 * (w0 * v00[e] + w1 * v01[e]) + h1 * (w0 * v10[e] + w1 * v11[e]);
                store8_nt(out + ((((size_t)n * OH + 2 * k + a) * OW) + 2 * j + b) * C + v * 8, o);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// out = relu?( (x + bias[c]) (+ residual) ) on [n_px][C]: the convolution bias, the ReLU between the two convs of a
// RefineNet residual unit and its skip add, as one pass (PyTorch runs them as 2-3 elementwise kernels after MIOpen's
// bias-less convolution).  Rounding follows the separate ops: x + bias is rounded to the tensor type first.
// `out_relu` (optional): additionally relu(out), the input of the next residual unit's first convolution, so that
// unit does not need its own ReLU pass.
template <typename T>
__global__ __launch_bounds__(256) void bias_act_kernel(const T *__restrict__ x, const T *__restrict__ bias, const T *__restrict__ residual,
                                                       const T *__restrict__ residual2, T *__restrict__ out, T *__restrict__ out_relu, size_t n_vec,
                                                       int VC, int relu) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * 256) {
        const int v = (int)(i % VC);
        float f[8], b[8];
        load8(x + i * 8, f);
        load8(bias + v * 8, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (float)(T)(f[j] + b[j]);
        if (residual) {
            float r[8];
            load8(residual + i * 8, r);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] += r[j];
        }
        if (residual2) {  // second skip (fusion block: output + rcu(x)); the first sum is rounded first, as separate ops would
            float r[8];
            load8(residual2 + i * 8, r);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = (float)(T)f[j] + r[j];
        }
        if (relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j], 0.f);
        }
        store8(out + i * 8, f);
        if (out_relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fmaxf((float)(T)f[j], 0.f);
            store8(out_relu + i * 8, f);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Patch embedding of ViT-L/16 (timm PatchEmbed: Conv2d(3, D, 16, 16)) as a GEMM: the P x P x C patch of output token (n, gy, gx) is
// copied into row m of a [tokens][P P C] matrix in (ky, kx, c) order -- the order of the weight tensor in channels-last memory
// format -- and hive_vit_linear does the rest.  A kernel row of a patch is P C contiguous values of the channels-last frame.
template <typename T>
__global__ __launch_bounds__(256) void patch_rows_kernel(const T *__restrict__ x, T *__restrict__ out, int N, int H, int W, int C, int P) {
    const int gh = H / P, gw = W / P, row_vals = P * C;        // values per kernel row of a patch
    const long long total = (long long)N * gh * gw * P;         // one (token, ky) pair per work item
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ky = (int)(i % P);
        const long long m = i / P;
        const int gx = (int)(m % gw), gy = (int)((m / gw) % gh), n = (int)(m / ((long long)gw * gh));
        const T *src = x + (((long long)n * H + gy * P + ky) * W + (long long)gx * P) * C;
        T *dst = out + (m * P + ky) * row_vals;
        if (row_vals % 8 == 0 && ((uintptr_t)src % 16 == 0)) {
            for (int v = 0; v < row_vals; v += 8) *reinterpret_cast<uint4 *>(dst + v) = *reinterpret_cast<const uint4 *>(src + v);
        } else {
            for (int v = 0; v < row_vals; ++v) dst[v] = src[v];
        }
    }
}

// ConvTranspose2d with kernel == stride == s (DPT-Large's reassemble stages 1 and 2): the transposed convolution is a 1 x 1 convolution
// to s s C channels ((dy, dx, co) order) followed by this scatter: in [N H W][s s C] -> out [N][s H][s W][C] (+ bias[co]).
template <typename T>
__global__ __launch_bounds__(256) void pixel_shuffle_bias_kernel(const T *__restrict__ in, const T *__restrict__ bias, T *__restrict__ out, int N, int H,
                                                                  int W, int C, int s) {
    const int VC = C >> 3;
    const long long total = (long long)N * H * W * s * s * VC;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int v = (int)(i % VC);
        long long r = i / VC;
        const int dx = (int)(r % s);
        r /= s;
        const int dy = (int)(r % s);
        r /= s;  // r = input pixel (n, y, x)
        const int xx = (int)(r % W), yy = (int)((r / W) % H), n = (int)(r / ((long long)W * H));
        float f[8], b[8];
        load8(in + i * 8, f);  // in is [pixel][dy][dx][C]: exactly the order of i
        if (bias) {
            load8(bias + v * 8, b);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] += b[j];
        }
        store8(out + ((((long long)n * H * s + (long long)yy * s + dy) * W * s) + (long long)xx * s + dx) * C + v * 8, f);
    }
}

// MaxPool2dSame(3, 2)(relu(GroupNorm(x))) in one pass (the ResNetV2 stem behind its convolution): out[n][oy][ox][c] = the maximum over the
// 3 x 3 / 2 window (TensorFlow "SAME" padding: padded positions never win) of the normalised, rectified and ROUNDED values -- rounding and
// ReLU are monotonic, so max(round(relu(y))) == round(relu(max(y))): bit-identical to gn_apply_kernel followed by the pooling kernel, without
// the normalised map (1.05 GB at 107 frames of 480 x 640) ever reaching memory.  A thread: one output pixel x 8 channels.
template <typename T>
__global__ __launch_bounds__(256) void gn_relu_maxpool_kernel(const T *__restrict__ x, const T *__restrict__ gamma, const T *__restrict__ beta,
                                                              const float *__restrict__ stats, T *__restrict__ out, int H, int W, int C, int G, int Ho,
                                                              int Wo, int pad_t, int pad_l) {
    const int VC = C >> 3, cpg = C / G;
    const int n = blockIdx.y;
    const long long total = (long long)Ho * Wo * VC;
    const int v = threadIdx.x % VC;  // VC (a power of two <= 256) divides the stride of the loop below: a thread keeps its channels
    float a[8], b[8];
    {
        float gm[8], bt[8];
        load8(gamma + v * 8, gm);
        load8(beta + v * 8, bt);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int g = (v * 8 + j) / cpg;
            const float mean = stats[2 * (n * G + g)], rstd = stats[2 * (n * G + g) + 1];
            a[j] = rstd * gm[j];
            b[j] = bt[j] - mean * a[j];
        }
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int pix = (int)(i / VC), ox = pix % Wo, oy = pix / Wo;
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = 0.f;  // the ReLU's floor: every window holds at least one real pixel
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = 2 * oy + ky - pad_t, ix = 2 * ox + kx - pad_l;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                    float f[8];
                    load8(x + (((size_t)n * H + iy) * W + ix) * C + v * 8, f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], f[j] * a[j] + b[j]);
                }
            }
        store8(out + (((size_t)n * Ho + oy) * Wo + ox) * C + v * 8, m);
    }
}

static bool pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

int hive_gn_finalize_tiles(hive_ctx *ctx, const float *d_partial, int N, int HW, int C, int G, int tile_rows, float eps, float *d_stats) {
    hipLaunchKernelGGL(gn_finalize_tiles_kernel, dim3(N * G), dim3(256), 0, ctx->stream, d_partial, C, G, tile_rows, HW, eps, d_stats, N * G);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

extern "C" {

static int group_norm_impl(hive_ctx *ctx, const void *d_x, int dtype, int N, int HW, int C, int G, const void *d_gamma, const void *d_beta, float eps,
                           const void *d_residual, int relu, void *d_out, const void *d_gn_partial, int gn_tile_rows) {
    HIVE_REQUIRE(ctx, d_x && d_gamma && d_beta && d_out, "group_norm: NULL argument");
    HIVE_REQUIRE(ctx, N > 0 && HW > 0 && C >= 8 && C <= 2048 && pow2(C) && G > 0 && C % G == 0,
                 "group_norm: need C a power of two in [8, 2048] and C %% G == 0 (N=%d HW=%d C=%d G=%d)", N, HW, C, G);
    HIVE_REQUIRE(ctx, dtype == HIVE_BF16 || dtype == HIVE_F16, "group_norm: dtype must be HIVE_F16 or HIVE_BF16");
    const bool from_tiles = d_gn_partial && gn_tile_rows > 0;
    HIVE_REQUIRE(ctx, !from_tiles || gn_tile_rows <= HW, "group_norm_stats: tiles of %d rows on samples of %d", gn_tile_rows, HW);
    const int VC = C / 8, PP = 256 / VC;
    const int slabs = std::max(1, std::min(64, std::min(HW / (4 * PP) + 1, (ctx->num_cus * 8 + N - 1) / N)));
    const size_t partial_floats = from_tiles ? 0 : (size_t)N * slabs * 2 * C, stats_floats = (size_t)N * G * 2;
    int rc = hive_reserve_device(ctx, &ctx->d_scratch, &ctx->scratch_bytes, (partial_floats + stats_floats) * sizeof(float));
    if (rc) return rc;
    float *partial = (float *)ctx->d_scratch, *stats = partial + partial_floats;
    const dim3 g1(slabs, N), g3(std::max(1, std::min((HW + PP - 1) / PP, (ctx->num_cus * 8 + N - 1) / N)), N);
    if (from_tiles)
        hipLaunchKernelGGL(gn_finalize_tiles_kernel, dim3(N * G), dim3(256), 0, ctx->stream, (const float *)d_gn_partial, C, G, gn_tile_rows, HW, eps, stats, N * G);
    if (dtype == HIVE_BF16) {
        if (!from_tiles) {
            hipLaunchKernelGGL(gn_partial_kernel<bf16>, g1, dim3(256), 0, ctx->stream, (const bf16 *)d_x, HW, C, slabs, partial);
            hipLaunchKernelGGL(gn_finalize_kernel, dim3(N * G), dim3(256), 0, ctx->stream, partial, C, G, slabs, HW, eps, stats, N * G);
        }
        hipLaunchKernelGGL(gn_apply_kernel<bf16>, g3, dim3(256), 0, ctx->stream, (const bf16 *)d_x, (const bf16 *)d_gamma, (const bf16 *)d_beta,
                           stats, (const bf16 *)d_residual, (bf16 *)d_out, HW, C, G, relu);
    } else {
        if (!from_tiles) {
            hipLaunchKernelGGL(gn_partial_kernel<_Float16>, g1, dim3(256), 0, ctx->stream, (const _Float16 *)d_x, HW, C, slabs, partial);
            hipLaunchKernelGGL(gn_finalize_kernel, dim3(N * G), dim3(256), 0, ctx->stream, partial, C, G, slabs, HW, eps, stats, N * G);
        }
        hipLaunchKernelGGL(gn_apply_kernel<_Float16>, g3, dim3(256), 0, ctx->stream, (const _Float16 *)d_x, (const _Float16 *)d_gamma,
                           (const _Float16 *)d_beta, stats, (const _Float16 *)d_residual, (_Float16 *)d_out, HW, C, G, relu);
    }
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

int hive_nhwc_group_norm(hive_ctx *ctx, const void *d_x, int dtype, int N, int HW, int C, int G, const void *d_gamma,
                         const void *d_beta, float eps, const void *d_residual, int relu, void *d_out) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    return group_norm_impl(ctx, d_x, dtype, N, HW, C, G, d_gamma, d_beta, eps, d_residual, relu, d_out, nullptr, 0);
}

int hive_nhwc_group_norm_stats(hive_ctx *ctx, const void *d_x, int dtype, int N, int HW, int C, int G, const void *d_gamma, const void *d_beta,
                               float eps, const void *d_residual, int relu, void *d_out, const void *d_gn_partial, int gn_tile_rows) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    return group_norm_impl(ctx, d_x, dtype, N, HW, C, G, d_gamma, d_beta, eps, d_residual, relu, d_out, d_gn_partial, gn_tile_rows);
}

int hive_nhwc_group_norm_relu_maxpool(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, int C, int G, const void *d_gamma, const void *d_beta,
                                      float eps, void *d_out, const void *d_gn_partial, int gn_tile_rows) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, d_x && d_gamma && d_beta && d_out && d_out != d_x, "group_norm_relu_maxpool: bad pointers");
    HIVE_REQUIRE(ctx, N > 0 && N <= 65535 && H > 0 && W > 0 && C >= 8 && C <= 2048 && pow2(C) && G > 0 && C % G == 0,
                 "group_norm_relu_maxpool: need C a power of two in [8, 2048], C %% G == 0, N <= 65535 (N=%d H=%d W=%d C=%d G=%d)", N, H, W, C, G);
    HIVE_REQUIRE(ctx, dtype == HIVE_BF16 || dtype == HIVE_F16, "group_norm_relu_maxpool: dtype must be HIVE_F16 or HIVE_BF16");
    const int HW = H * W;
    const bool from_tiles = d_gn_partial && gn_tile_rows > 0;
    HIVE_REQUIRE(ctx, !from_tiles || gn_tile_rows <= HW, "group_norm_relu_maxpool: tiles of %d rows on samples of %d", gn_tile_rows, HW);
    const int VC = C / 8, PP = 256 / VC;
    const int slabs = std::max(1, std::min(64, std::min(HW / (4 * PP) + 1, (ctx->num_cus * 8 + N - 1) / N)));
    const size_t partial_floats = from_tiles ? 0 : (size_t)N * slabs * 2 * C, stats_floats = (size_t)N * G * 2;
    int rc = hive_reserve_device(ctx, &ctx->d_scratch, &ctx->scratch_bytes, (partial_floats + stats_floats) * sizeof(float));
    if (rc) return rc;
    float *partial = (float *)ctx->d_scratch, *stats = partial + partial_floats;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int pad_t = std::max((Ho - 1) * 2 + 3 - H, 0) / 2, pad_l = std::max((Wo - 1) * 2 + 3 - W, 0) / 2;
    const long long per_img = (long long)Ho * Wo * VC;
    const dim3 g1(slabs, N), gp((unsigned)std::max<long long>(1, std::min<long long>((per_img + 255) / 256, ((long long)ctx->num_cus * 32 + N - 1) / N)), N);
    if (from_tiles)
        hipLaunchKernelGGL(gn_finalize_tiles_kernel, dim3(N * G), dim3(256), 0, ctx->stream, (const float *)d_gn_partial, C, G, gn_tile_rows, HW, eps, stats, N * G);
    if (dtype == HIVE_BF16) {
        if (!from_tiles) {
            hipLaunchKernelGGL(gn_partial_kernel<bf16>, g1, dim3(256), 0, ctx->stream, (const bf16 *)d_x, HW, C, slabs, partial);
            hipLaunchKernelGGL(gn_finalize_kernel, dim3(N * G), dim3(256), 0, ctx->stream, partial, C, G, slabs, HW, eps, stats, N * G);
        }
        hipLaunchKernelGGL(gn_relu_maxpool_kernel<bf16>, gp, dim3(256), 0, ctx->stream, (const bf16 *)d_x, (const bf16 *)d_gamma, (const bf16 *)d_beta, stats,
                           (bf16 *)d_out, H, W, C, G, Ho, Wo, pad_t, pad_l);
    } else {
        if (!from_tiles) {
            hipLaunchKernelGGL(gn_partial_kernel<_Float16>, g1, dim3(256), 0, ctx->stream, (const _Float16 *)d_x, HW, C, slabs, partial);
            hipLaunchKernelGGL(gn_finalize_kernel, dim3(N * G), dim3(256), 0, ctx->stream, partial, C, G, slabs, HW, eps, stats, N * G);
        }
        hipLaunchKernelGGL(gn_relu_maxpool_kernel<_Float16>, gp, dim3(256), 0, ctx->stream, (const _Float16 *)d_x, (const _Float16 *)d_gamma,
                           (const _Float16 *)d_beta, stats, (_Float16 *)d_out, H, W, C, G, Ho, Wo, pad_t, pad_l);
    }
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

int hive_nhwc_bias_act(hive_ctx *ctx, const void *d_x, int dtype, int64_t n_px, int C, const void *d_bias, int relu,
                       const void *d_residual, const void *d_residual2, void *d_out, void *d_out_relu) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, d_x && d_bias && d_out, "bias_act: NULL argument");
    HIVE_REQUIRE(ctx, n_px > 0 && C > 0 && C % 8 == 0, "bias_act: need C %% 8 == 0 (n_px=%lld C=%d)", (long long)n_px, C);
    const size_t n_vec = (size_t)n_px * (C / 8);
    const dim3 grid((unsigned)std::min<size_t>((n_vec + 255) / 256, (size_t)ctx->num_cus * 32));
    if (dtype == HIVE_BF16)
        hipLaunchKernelGGL(bias_act_kernel<bf16>, grid, dim3(256), 0, ctx->stream, (const bf16 *)d_x, (const bf16 *)d_bias, (const bf16 *)d_residual,
                           (const bf16 *)d_residual2, (bf16 *)d_out, (bf16 *)d_out_relu, n_vec, C / 8, relu);
    else if (dtype == HIVE_F16)
        hipLaunchKernelGGL(bias_act_kernel<_Float16>, grid, dim3(256), 0, ctx->stream, (const _Float16 *)d_x, (const _Float16 *)d_bias,
                           (const _Float16 *)d_residual, (const _Float16 *)d_residual2, (_Float16 *)d_out, (_Float16 *)d_out_relu, n_vec, C / 8, relu);
    else
        return hive_fail(ctx, HIVE_ERR_INVALID, "bias_act: dtype must be HIVE_F16 or HIVE_BF16");
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

int hive_nhwc_upsample2x(hive_ctx *ctx, const void *d_in, const void *d_bias, int dtype, int N, int H, int W, int C, void *d_out) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, d_in && d_out, "upsample2x: NULL argument");
    HIVE_REQUIRE(ctx, N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "upsample2x: need C %% 8 == 0 (N=%d H=%d W=%d C=%d)", N, H, W, C);
    HIVE_REQUIRE(ctx, H <= 65535 && N <= 65535, "upsample2x: H %d / N %d too large for the launch grid", H, N);
    const bool gather_only = getenv("HIVE_UPSAMPLE_GATHER") != nullptr;  // (A / B runs and the bit-identity test)
    const size_t lds = (size_t)(UP_TY + 2) * (UP_TX + 2) * C * 2;
    if (C <= 512 && !gather_only && (H + UP_TY - 1) / UP_TY <= 65535) {  // through an LDS tile
        const dim3 tiles((unsigned)((W + UP_TX - 1) / UP_TX), (unsigned)((H + UP_TY - 1) / UP_TY), (unsigned)N);
        if (dtype == HIVE_BF16) {
            HIVE_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)upsample2x_lds_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            hipLaunchKernelGGL(upsample2x_lds_kernel<bf16>, tiles, dim3(256), lds, ctx->stream, (const bf16 *)d_in, (const bf16 *)d_bias, (bf16 *)d_out, N, H, W, C);
        } else if (dtype == HIVE_F16) {
            HIVE_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)upsample2x_lds_kernel<_Float16>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            hipLaunchKernelGGL(upsample2x_lds_kernel<_Float16>, tiles, dim3(256), lds, ctx->stream, (const _Float16 *)d_in, (const _Float16 *)d_bias, (_Float16 *)d_out,
                               N, H, W, C);
        } else {
            return hive_fail(ctx, HIVE_ERR_INVALID, "upsample2x: dtype must be HIVE_F16 or HIVE_BF16");
        }
        HIVE_CHECK_HIP(ctx, hipGetLastError());
        return HIVE_OK;
    }
    const dim3 grid((unsigned)((W * (C / 8) + 255) / 256), (unsigned)H, (unsigned)N);  // a thread per 2 x 2 output pixels x 8 channels
    if (dtype == HIVE_BF16)
        hipLaunchKernelGGL(upsample2x_kernel<bf16>, grid, dim3(256), 0, ctx->stream, (const bf16 *)d_in, (const bf16 *)d_bias, (bf16 *)d_out, N, H, W, C);
    else if (dtype == HIVE_F16)
        hipLaunchKernelGGL(upsample2x_kernel<_Float16>, grid, dim3(256), 0, ctx->stream, (const _Float16 *)d_in, (const _Float16 *)d_bias, (_Float16 *)d_out, N, H, W, C);
    else
        return hive_fail(ctx, HIVE_ERR_INVALID, "upsample2x: dtype must be HIVE_F16 or HIVE_BF16");
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

int hive_patch_rows(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, int C, int patch, void *d_out) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, d_x && d_out, "patch_rows: NULL argument");
    HIVE_REQUIRE(ctx, N > 0 && C > 0 && patch > 0 && H >= patch && W >= patch && H % patch == 0 && W % patch == 0,
                 "patch_rows: need H, W multiples of the patch size (N=%d H=%d W=%d C=%d patch=%d)", N, H, W, C, patch);
    const long long total = (long long)N * (H / patch) * (W / patch) * patch;
    const dim3 grid((unsigned)std::min<long long>((total + 255) / 256, (long long)ctx->num_cus * 32));
    if (dtype == HIVE_BF16)
        hipLaunchKernelGGL(patch_rows_kernel<bf16>, grid, dim3(256), 0, ctx->stream, (const bf16 *)d_x, (bf16 *)d_out, N, H, W, C, patch);
    else if (dtype == HIVE_F16)
        hipLaunchKernelGGL(patch_rows_kernel<_Float16>, grid, dim3(256), 0, ctx->stream, (const _Float16 *)d_x, (_Float16 *)d_out, N, H, W, C, patch);
    else
        return hive_fail(ctx, HIVE_ERR_INVALID, "patch_rows: dtype must be HIVE_F16 or HIVE_BF16");
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

int hive_nhwc_pixel_shuffle_bias(hive_ctx *ctx, const void *d_in, const void *d_bias, int dtype, int N, int H, int W, int C, int s, void *d_out) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, d_in && d_out && d_in != d_out, "pixel_shuffle_bias: NULL or aliasing argument");
    HIVE_REQUIRE(ctx, N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && s > 0 && s <= 8, "pixel_shuffle_bias: need C %% 8 == 0, 1 <= s <= 8 (N=%d H=%d W=%d C=%d s=%d)", N,
                 H, W, C, s);
    const long long total = (long long)N * H * W * s * s * (C / 8);
    const dim3 grid((unsigned)std::min<long long>((total + 255) / 256, (long long)ctx->num_cus * 32));
    if (dtype == HIVE_BF16)
        hipLaunchKernelGGL(pixel_shuffle_bias_kernel<bf16>, grid, dim3(256), 0, ctx->stream, (const bf16 *)d_in, (const bf16 *)d_bias, (bf16 *)d_out, N, H, W, C, s);
    else if (dtype == HIVE_F16)
        hipLaunchKernelGGL(pixel_shuffle_bias_kernel<_Float16>, grid, dim3(256), 0, ctx->stream, (const _Float16 *)d_in, (const _Float16 *)d_bias, (_Float16 *)d_out,
                           N, H, W, C, s);
    else
        return hive_fail(ctx, HIVE_ERR_INVALID, "pixel_shuffle_bias: dtype must be HIVE_F16 or HIVE_BF16");
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

}  // extern "C"
