// GroupNorm statistics of a 1 x 1 convolution's output WITHOUT computing the convolution (the ResNetV2 bottlenecks' expanding
// convolutions conv3 / downsample, /root/reference's DPT-Hybrid backbone as called at hive/dataset_adaptors.py:1419).
//
// y = W x is linear, so over the pixels p of one sample and the channels c of one group g
//     sum   y_c(p)   = u_g . s          u_g = sum_{c in g} w_c (a C_in vector),        s = sum_p x_p
//     sum   y_c(p)^2 = <G_g, S>         G_g = sum_{c in g} w_c w_c^T (C_in x C_in),    S = sum_p x_p x_p^T  (the Gram matrix of the sample's input)
// u_g and G_g depend on the weights only (hive_gn_gram_prepare, once per network); S and s cost one read of the C_in-wide input and C_in^2 MACs per
// pixel on the matrix cores -- a quarter of the convolution's for C_out = 4 C_in, no 4 C-wide epilogue arithmetic at all.  The two-pass GroupNorm
// convolution (conv.hip, hive_nhwc_conv_gn_apply) took its statistics from a first pass that ran the whole convolution and reduced the (rounded)
// accumulators in its epilogue: 205 / 132 / 92 us per call at 120 x 160 / 60 x 80 / 30 x 40 x 107 frames against ~60 / 30 / 15 us of input reading.
// What changes numerically: these are the statistics of the EXACT products (float32 sums), not of the outputs after their rounding to 16 bits; the
// rounding noise of tens of thousands of values per group averages out of both moments (tests: mean and rstd agree to ~1e-4 relative).
//
// gram_kernel: one workgroup = 128 consecutive output pixels at a time of one (sample, pixel chunk); the tile sits in LDS as C_in / 64 panels of
// [128 pixels][64 channels] (128-byte rows, 16-byte chunk c of row r at slot c ^ (r & 6)), filled by LDS-DMA, double-buffered.  The MFMA wants both
// operands as [channel][8 consecutive pixels per lane]: ds_read_b64_tr_b16 delivers exactly that from the pixel-major rows (4 pixels x 16 channels per
// 16-lane group; two reads per fragment; the pixel order inside a fragment is the same for both operands, which is all a contraction needs).
// Waves: C_in = 64: four waves split the pixels (each its own partial, summed later); 128: wave w owns the 64 x 64 block (w >> 1, w & 1); 256: eight
// waves, wave w owns blocks (w >> 1, 2 (w & 1)) and (w >> 1, 2 (w & 1) + 1).
#include "hive_internal.hpp"
#include "mfma_pipe.hpp"

#include <algorithm>

using hive_mfma::f32x4;
using hive_mfma::vec;

namespace {

template <typename T>
struct GramParams {
    const T *x;      // [N][H][W][C_in]
    const T *zeros;  // >= 128 bytes of zeros (rows past the end of a chunk)
    float *S;        // [N][parts][C_in][C_in]
    float *s;        // [N][parts][C_in]
    int H, W, stride, Wo, HW;  // input size, stride, output width, output pixels per sample
    int chunk_px;              // output pixels per chunk (a multiple of 128)
    int parts;                 // partial results per sample (= chunks)
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char lds_byte;

// fragment [16 channels 16 i ..][32 pixels of sub-tile t] of a panel: lane (fr = channel, fq = pixel octet) gets 8 pixels of its channel
template <typename T>
__device__ __forceinline__ vec<T, 8> gram_frag(lds_byte *panel, int t, int i, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int r0 = 32 * t + 4 * g + q, r1 = r0 + 16;
    const int c = 2 * i + (pp >> 1), h8 = (pp & 1) * 8;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(panel + r0 * 128 + ((c ^ (r0 & 6)) << 4) + h8));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(panel + r1 * 128 + ((c ^ (r1 & 6)) << 4) + h8));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(vec<T, 8>, both);
}

template <typename T, int CIN, int NW>
__global__ __launch_bounds__(NW * 64) void gram_kernel(GramParams<T> p) {
    constexpr int PANELS = CIN / 64, TILE = PANELS * 128 * 128;  // bytes of a 128-pixel tile
    constexpr int NBJ = CIN == 256 ? 2 : 1;                       // column blocks per wave
    constexpr int PIECES = PANELS * 16, PER_WAVE = PIECES / NW;   // LDS-DMA pieces (8 rows x 128 B) per tile and per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // 2 tiles
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int px0 = chunk * p.chunk_px, px1 = min(px0 + p.chunk_px, p.HW);
    const int n_tiles = (px1 - px0 + 127) / 128;
    const T *xs = p.x + (size_t)n * p.H * p.W * CIN;

    auto issue_tile = [&](int tile, int buf) {
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) {
            const int piece = wave + j * NW, panel = piece >> 4, r8 = piece & 15;
            const int row = 8 * r8 + (lane >> 3), slot = lane & 7;
            const int q = px0 + tile * 128 + row;
            const T *g = p.zeros + slot * 8;
            if (q < px1) {
                const int oy = q / p.Wo, ox = q - oy * p.Wo;
                g = xs + ((size_t)(oy * p.stride) * p.W + ox * p.stride) * CIN + panel * 64 + (slot ^ (row & 6)) * 8;
            }
            __builtin_amdgcn_global_load_lds((const void *)g, (__attribute__((address_space(3))) void *)(lds + buf * TILE + panel * 16384 + r8 * 1024), 16, 0, 0);
        }
    };

    const int bi = CIN == 64 ? 0 : wave >> 1, bj0 = CIN == 64 ? 0 : (CIN == 128 ? (wave & 1) : 2 * (wave & 1));
    f32x4 acc[NBJ][4][4], sums[4];
#pragma unroll
    for (int b = 0; b < NBJ; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) sums[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    vec<T, 8> ones;
#pragma unroll
    for (int k = 0; k < 8; ++k) ones[k] = (T)1.0f;
    const bool do_sums = CIN == 64 || (wave & 1) == 0;  // (wave-uniform) one wave per row panel adds the panel's channel sums

    if (n_tiles > 0) issue_tile(0, 0);
    for (int tile = 0; tile < n_tiles; ++tile) {
        const int buf = tile & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // the tile has landed; everyone has finished reading the other buffer
        if (tile + 1 < n_tiles) issue_tile(tile + 1, buf ^ 1);
        lds_byte *tb = (lds_byte *)lds + buf * TILE;
#pragma unroll
        for (int tt = 0; tt < (CIN == 64 ? 1 : 4); ++tt) {
            const int t = CIN == 64 ? wave : tt;  // C_in = 64: the waves split the tile's four 32-pixel sub-tiles
            vec<T, 8> fa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = gram_frag<T>(tb + bi * 16384, t, i, lane);
#pragma unroll
            for (int b = 0; b < NBJ; ++b) {
                vec<T, 8> fb[4];
                // (read again even where the column block IS the row block: the transposed read needs every lane active, so it stays out of any branch)
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = CIN == 64 ? fa[j] : gram_frag<T>(tb + (bj0 + b) * 16384, t, j, lane);
                // acc[b][i][j][e] = S[64 bi + 16 i + 4 fq + e][64 bj + 16 j + fr]
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[b][i][j] = hive_mfma::mfma16(fa[i], fb[j], acc[b][i][j]);
            }
            if (do_sums) {
#pragma unroll
                for (int i = 0; i < 4; ++i) sums[i] = hive_mfma::mfma16(fa[i], ones, sums[i]);
            }
        }
    }
    // S is symmetric: the lane's 4 consecutive ROW indices go out as 4 consecutive columns of row (64 bj + 16 j + fr): 16-byte stores
    const int fr = lane & 15, fq = lane >> 4;
    float *So = p.S + ((size_t)n * p.parts + chunk) * CIN * CIN;
    float *so = p.s + ((size_t)n * p.parts + chunk) * CIN;
    if constexpr (CIN == 64) {
        // the four waves hold partial sums over different pixels: through LDS, added in wave order -- in two halves (rows 0-31 / 32-63 of S^T, the sums with
        // the second), so that the workgroup stays within the two tiles' 32 KiB + 1 KiB and four of them fit a CU
        constexpr int HALF = 2048 + 64;  // floats per wave and half
        float *mine = reinterpret_cast<float *>(lds) + wave * HALF;
        const float *all = reinterpret_cast<const float *>(lds);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            __syncthreads();  // everyone has finished reading the tiles / the first half
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4 *>(mine + (16 * j + fr) * 64 + 16 * i + 4 * fq) = acc[0][i][2 * half + j];
            if (half == 1 && fr == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4 *>(mine + 2048 + 16 * i + 4 * fq) = sums[i];
            }
            __syncthreads();
            for (int e = tid * 4; e < (half ? 2048 + 64 : 2048); e += 256 * 4) {
                f32x4 v = *reinterpret_cast<const f32x4 *>(all + e);
#pragma unroll
                for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4 *>(all + w * HALF + e);
                if (e < 2048)
                    *reinterpret_cast<f32x4 *>(So + half * 2048 + e) = v;
                else
                    *reinterpret_cast<f32x4 *>(so + e - 2048) = v;
            }
        }
    } else {
#pragma unroll
        for (int b = 0; b < NBJ; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<f32x4 *>(So + (size_t)(64 * (bj0 + b) + 16 * j + fr) * CIN + 64 * bi + 16 * i + 4 * fq) = acc[b][i][j];
        if (do_sums && fr == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4 *>(so + 64 * bi + 16 * i + 4 * fq) = sums[i];
        }
    }
}

// weights -> tables: Gt[g][k][l] = sum_{c in g} w[c][k] w[c][l] (float64 sums of the 16-bit weights' products), u[g][k] = sum_{c in g} w[c][k].  grid (C_in, G)
template <typename T>
__global__ __launch_bounds__(256) void gram_tables_kernel(const T *__restrict__ w, int Cin, int Cout, int G, float *__restrict__ Gt, float *__restrict__ u) {
    const int k = blockIdx.x, g = blockIdx.y, cpg = Cout / G;
    for (int l = threadIdx.x; l < Cin; l += 256) {
        double a = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) a += (double)(float)w[(size_t)c * Cin + k] * (double)(float)w[(size_t)c * Cin + l];
        Gt[((size_t)g * Cin + k) * Cin + l] = (float)a;
    }
    if (threadIdx.x == 0) {
        double a = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) a += (double)(float)w[(size_t)c * Cin + k];
        u[(size_t)g * Cin + k] = (float)a;
    }
}

// Q[n][g][kc] = sum over the 256 (k, l) pairs of chunk kc and over the sample's parts of S[n][part][kl] Gt[g][kl].  grid (C_in^2 / 256, ceil(N / 16)): a workgroup holds
// the chunk's values of every group (G x 256) and of 16 samples (their parts added in part order, four independent loads in flight) in LDS; thread t takes sample
// t & 15 and the groups (t >> 4), (t >> 4) + 16, ...
constexpr int QUAD_KL = 256, QUAD_NS = 16;
__global__ __launch_bounds__(256) void gram_quad_kernel(const float *__restrict__ S, const float *__restrict__ Gt, int N, int parts, int CC, int G, int KC, float *__restrict__ Q) {
    extern __shared__ __attribute__((aligned(16))) float gl[];  // [G][QUAD_KL] + [QUAD_NS][QUAD_KL + 1]
    float *vl = gl + G * QUAD_KL;
    const int kc = blockIdx.x, kl0 = kc * QUAD_KL, n0 = blockIdx.y * QUAD_NS, tid = threadIdx.x;
    const int nn = min(QUAD_NS, N - n0);
    for (int e = tid; e < G * QUAD_KL; e += 256) gl[e] = Gt[(size_t)(e / QUAD_KL) * CC + kl0 + (e % QUAD_KL)];
    // the samples' values: 16 samples x 64 float4 = 4 float4 per thread, the parts added in part order, two parts' loads in flight at a time
    {
        f32x4 v[4];
        const f32x4 *sp[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = tid + 256 * r, ns = min(e >> 6, nn - 1), q4 = e & 63;
            sp[r] = reinterpret_cast<const f32x4 *>(S + ((size_t)(n0 + ns) * parts) * CC + kl0) + q4;
            v[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const size_t step = (size_t)CC / 4;
        int part = 0;
        for (; part + 2 <= parts; part += 2) {
            f32x4 a[4], b[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = sp[r][(size_t)part * step], b[r] = sp[r][(size_t)(part + 1) * step];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (v[r] + a[r]) + b[r];
        }
        if (part < parts) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += sp[r][(size_t)part * step];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = tid + 256 * r, ns = e >> 6, q4 = e & 63;
#pragma unroll
            for (int k = 0; k < 4; ++k) vl[ns * (QUAD_KL + 1) + 4 * q4 + k] = v[r][k];
        }
    }
    __syncthreads();
    const int ns = tid & (QUAD_NS - 1);
    if (ns < nn) {
        const float *vv = vl + ns * (QUAD_KL + 1);
        for (int g = tid >> 4; g < G; g += 16) {
            const float *gv = gl + g * QUAD_KL;
            float a0 = 0.f, a1 = 0.f;
#pragma unroll 4
            for (int kl = 0; kl < QUAD_KL; kl += 2) a0 += vv[kl] * gv[kl], a1 += vv[kl + 1] * gv[kl + 1];
            Q[((size_t)(n0 + ns) * G + g) * KC + kc] = a0 + a1;
        }
    }
}

// (mean, rstd) of the groups of sample n: sum y = u_g . (sum over parts of s), sum y^2 = sum over chunks of Q; float64 from here on.  One workgroup per sample:
// thread t -> group t % G, slice t / G of the 256 / G slices of the chunk / channel ranges; the slices are added in slice order through LDS.
__global__ __launch_bounds__(256) void gram_finalize_kernel(const float *__restrict__ Q, const float *__restrict__ s, const float *__restrict__ u, int G, int Cin, int parts,
                                                            int KC, double count, float eps, float *__restrict__ stats) {
    __shared__ float ssum[256];
    __shared__ double red[2][256];
    const int n = blockIdx.x, tid = threadIdx.x;
    if (tid < Cin) {
        float a = 0.f;
        for (int part = 0; part < parts; ++part) a += s[((size_t)n * parts + part) * Cin + tid];
        ssum[tid] = a;
    }
    __syncthreads();
    const int slices = 256 / G, g = tid % G, sl = tid / G;
    double sq = 0.0, sm = 0.0;
    if (sl < slices) {
        for (int kc = sl; kc < KC; kc += slices) sq += (double)Q[((size_t)n * G + g) * KC + kc];
        for (int k = sl; k < Cin; k += slices) sm += (double)ssum[k] * (double)u[(size_t)g * Cin + k];
    }
    red[0][tid] = sq;
    red[1][tid] = sm;
    __syncthreads();
    if (tid < G) {
        sq = 0.0, sm = 0.0;
        for (int i = 0; i < slices; ++i) sq += red[0][i * G + tid], sm += red[1][i * G + tid];
        const double mean = sm / count, var = fmax(sq / count - mean * mean, 0.0);
        stats[2 * (n * G + tid)] = (float)mean;
        stats[2 * (n * G + tid) + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

template <typename T, int CIN, int NW>
int launch_gram(hive_ctx *ctx, const GramParams<T> &p, int chunks, int N) {
    constexpr int lds = CIN == 64 ? 4 * (2048 + 64) * 4 : 2 * (CIN / 64) * 128 * 128;  // two tiles; C_in = 64: also half of the four waves' partial sums (33 792 bytes)
    static bool set[64] = {};
    if (!(ctx->device < 64 && set[ctx->device])) {
        HIVE_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)gram_kernel<T, CIN, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        if (ctx->device < 64) set[ctx->device] = true;
    }
    hipLaunchKernelGGL((gram_kernel<T, CIN, NW>), dim3(chunks, N), dim3(NW * 64), lds, ctx->stream, p);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

// pixel chunks per sample: whole 128-pixel tiles, and as many as fill the chip's workgroup slots once (C_in = 64 / 128 / 256: 4 / 2 / 1 workgroups per CU by LDS) --
// N x chunks just under the slots, never just over (a second round that is nearly empty costs a whole round); every chunk leaves a partial Gram matrix to be read again
void gram_chunks(int num_cus, int N, int HW, int C_in, int *chunks, int *chunk_px) {
    const int slots = (C_in == 64 ? 4 : (C_in == 128 ? 2 : 1)) * num_cus;
    int c = std::max(1, std::min((HW + 255) / 256, slots / std::max(N, 1)));
    if (const char *e = getenv("HIVE_GRAM_CHUNKS")) c = std::max(1, std::min((HW + 255) / 256, atoi(e)));  // tuning override
    *chunk_px = ((HW + c - 1) / c + 127) / 128 * 128;
    *chunks = (HW + *chunk_px - 1) / *chunk_px;
}

template <typename T>
int gram_stats_t(hive_ctx *ctx, const void *d_x, int N, int H, int W, int C_in, int stride, int Ho, int Wo, int G, int cpg, const float *d_tables, float eps,
                 float *d_stats, float *d_S_out, float *d_s_out) {
    const int HW = Ho * Wo, CC = C_in * C_in;
    int chunks, chunk_px;
    gram_chunks(ctx->num_cus, N, HW, C_in, &chunks, &chunk_px);
    const int parts = chunks, KC = CC / QUAD_KL;
    const size_t floats = (size_t)N * parts * (CC + C_in) + (size_t)KC * N * G;
    int rc = hive_reserve_device(ctx, &ctx->d_gram, &ctx->gram_bytes, floats * sizeof(float));
    if (rc) return rc;
    GramParams<T> p{};
    p.x = (const T *)d_x;
    p.zeros = (const T *)ctx->d_zeros;
    p.S = (float *)ctx->d_gram;
    p.s = p.S + (size_t)N * parts * CC;
    float *Q = p.s + (size_t)N * parts * C_in;
    p.H = H, p.W = W, p.stride = stride, p.Wo = Wo, p.HW = HW, p.chunk_px = chunk_px, p.parts = parts;
    if (C_in == 64)
        rc = launch_gram<T, 64, 4>(ctx, p, chunks, N);
    else if (C_in == 128)
        rc = launch_gram<T, 128, 4>(ctx, p, chunks, N);
    else
        rc = launch_gram<T, 256, 8>(ctx, p, chunks, N);
    if (rc) return rc;
    if (d_S_out) {  // (tests: the first part's Gram matrix and sums are only meaningful with parts == 1; the caller sums the parts)
        HIVE_CHECK_HIP(ctx, hipMemcpyAsync(d_S_out, p.S, (size_t)N * parts * CC * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        HIVE_CHECK_HIP(ctx, hipMemcpyAsync(d_s_out, p.s, (size_t)N * parts * C_in * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    }
    const size_t quad_lds = ((size_t)G * QUAD_KL + QUAD_NS * (QUAD_KL + 1)) * sizeof(float);
    static bool qset[64] = {};
    if (!(ctx->device < 64 && qset[ctx->device])) {
        HIVE_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)gram_quad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(((size_t)64 * QUAD_KL + QUAD_NS * (QUAD_KL + 1)) * sizeof(float))));
        if (ctx->device < 64) qset[ctx->device] = true;
    }
    hipLaunchKernelGGL(gram_quad_kernel, dim3(KC, (N + QUAD_NS - 1) / QUAD_NS), dim3(256), quad_lds, ctx->stream, p.S, d_tables, N, parts, CC, G, KC, Q);
    hipLaunchKernelGGL(gram_finalize_kernel, dim3(N), dim3(256), 0, ctx->stream, Q, p.s, d_tables + (size_t)G * CC, G, C_in, parts, KC, (double)HW * cpg, eps, d_stats);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

}  // namespace

// (mean, rstd)[N][G] of GroupNorm(conv1x1(x, w)) from the Gram matrices of x and the tables hive_gn_gram_prepare made of w; see the header of this file
int hive_gram_gn_stats(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, int C_in, int C_out, int stride, int Ho, int Wo, int G, const float *d_tables,
                       float eps, float *d_stats, float *d_S_out, float *d_s_out) {
    HIVE_REQUIRE(ctx, C_in == 64 || C_in == 128 || C_in == 256, "gram statistics: C_in must be 64, 128 or 256 (got %d)", C_in);
    HIVE_REQUIRE(ctx, G >= 4 && G <= 64 && 256 % G == 0 && C_out % G == 0 && d_tables && d_stats, "gram statistics: groups must be 4 .. 64 and divide 256 (got %d) / NULL argument", G);
    if (dtype == HIVE_BF16) return gram_stats_t<__bf16>(ctx, d_x, N, H, W, C_in, stride, Ho, Wo, G, C_out / G, d_tables, eps, d_stats, d_S_out, d_s_out);
    if (dtype == HIVE_F16) return gram_stats_t<_Float16>(ctx, d_x, N, H, W, C_in, stride, Ho, Wo, G, C_out / G, d_tables, eps, d_stats, d_S_out, d_s_out);
    return hive_fail(ctx, HIVE_ERR_INVALID, "gram statistics: dtype must be HIVE_F16 or HIVE_BF16");
}

extern "C" int64_t hive_gn_gram_table_floats(int C_in, int G) { return (int64_t)G * C_in * C_in + (int64_t)G * C_in; }

extern "C" int hive_gn_gram_prepare(hive_ctx *ctx, const void *d_w, int dtype, int C_in, int C_out, int G, float *d_tables) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, d_w && d_tables && C_in > 0 && G > 0 && C_out % G == 0, "gn_gram_prepare: bad arguments");
    float *Gt = d_tables, *u = d_tables + (size_t)G * C_in * C_in;
    if (dtype == HIVE_BF16)
        hipLaunchKernelGGL(gram_tables_kernel<__bf16>, dim3(C_in, G), dim3(256), 0, ctx->stream, (const __bf16 *)d_w, C_in, C_out, G, Gt, u);
    else if (dtype == HIVE_F16)
        hipLaunchKernelGGL(gram_tables_kernel<_Float16>, dim3(C_in, G), dim3(256), 0, ctx->stream, (const _Float16 *)d_w, C_in, C_out, G, Gt, u);
    else
        return hive_fail(ctx, HIVE_ERR_INVALID, "gn_gram_prepare: dtype must be HIVE_F16 or HIVE_BF16");
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

// test / probe entry: the statistics alone, and optionally the partial Gram matrices [N][parts][C_in][C_in] and sums [N][parts][C_in] (parts returned)
extern "C" int hive_gn_gram_stats(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, int C_in, int C_out, int stride, int H_out, int W_out, int G,
                                  const float *d_tables, float eps, float *d_stats, float *d_S_out, float *d_s_out) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    return hive_gram_gn_stats(ctx, d_x, dtype, N, H, W, C_in, C_out, stride, H_out, W_out, G, d_tables, eps, d_stats, d_S_out, d_s_out);
}

extern "C" int hive_gn_gram_parts(hive_ctx *ctx, int N, int C_in, int H_out, int W_out) {  // partial results per sample of the call above (for sizing d_S_out)
    int chunks, chunk_px;
    gram_chunks(ctx ? ctx->num_cus : 256, N, H_out * W_out, C_in, &chunks, &chunk_px);
    return chunks;
}
