// The 3 x 3 convolution of a stage-1 ResNetV2 bottleneck (64 -> 64 channels, stride 1, TensorFlow "SAME" = one pixel of zeros all round) with
// the GroupNorm + ReLU in FRONT of it applied while its input is staged, for gfx950.  timm 0.5.4 `Bottleneck`:
//     x = norm1(conv1(x));  x = norm2(conv2(x))          (reached from DPTDepthModel.forward, /root/reference/hive/dataset_adaptors.py:1419)
// Here  t2 = conv2(relu(norm1(t)))  is ONE kernel: conv1's raw output t goes in, conv2's raw output t2 and the sums of norm2 come out.
//
// Why its own kernel: in the general implicit-GEMM kernel (conv.hip) a 64-channel 3 x 3 convolution re-reads its A tile once per tap by
// LDS-DMA and streams 72 KB of weights per tile -- 360 KB through the CU's vector-memory path for 32 KB of output, 362 us per call at
// 120 x 160 x 107 (that path moves ~27 B/clk/CU: tools/ubench) -- and the GroupNorm + ReLU between the two convolutions is a pass of its
// own (95 us).  With 64 channels everything fits the CU: the whole weight matrix (64 x 576, 73 KB) stays in LDS for the life of a persistent
// workgroup, an 18 x 34-pixel input patch (77 KB) feeds all nine taps of a 16 x 32 output tile, and because the patch passes through
// registers on its way to LDS (as in stem.hip) the normalisation costs a few VALU instructions per loaded value: relu(t * a + b) with
// a = rstd * gamma, b = beta - mean * a, every operation rounded on its own and the result rounded to T -- the arithmetic of
// gn_apply_kernel (dpt_ops.hip; this file is built with -ffp-contract=off as well), so the convolution sees bit for bit the tensor the
// separate pass would have written.  Padding positions are zeros of the NORMALISED tensor (not normalised zeros).
//
// MFMA: 8 waves, wave w owns output rows 2 w, 2 w + 1 of the tile (64 pixels x 64 channels: 4 x 4 accumulators of
// v_mfma_f32_16x16x32_{bf16,f16}); K order = tap outer, 32-channel half inner -- the order of conv.hip's K-steps for C_in = 64, so the
// outputs are bit-identical to hive_nhwc_conv's.  The GroupNorm sums of the output come from the accumulators (as conv.hip's
// gn_sums_from_acc), one row of partials per tile in the layout hive_nhwc_group_norm_stats reads.
#include "hive_internal.hpp"

#include <algorithm>

#include "mfma_pipe.hpp"

using hive_mfma::f32x4;
using hive_mfma::vec;

namespace {

constexpr int BN_TH = 16, BN_TW = 32;             // output tile
constexpr int BN_PH = BN_TH + 2, BN_PW = BN_TW + 2;  // input patch 18 x 34 pixels, 64 channels = 128 B per pixel
constexpr int BN_K = 9 * 64;                      // 576
constexpr int BN_WP = BN_K + 16;                  // weight row pitch: 1184 B = 74 chunks of 16 B.  A ds_read_b128 is served in four groups of 16 lanes that are NOT
                                                  // consecutive lanes ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS): a group mixes rows fr of chunk fq with rows of chunk
                                                  // fq + 1, so the 16 slots (fr pitch + fq) mod 16 must be distinct over THAT set -- true for pitch = 2 mod 4 chunks.  Round 3's
                                                  // 73 chunks (distinct over 16 consecutive lanes, the wrong set) collided 7 of 16 times: SQ_LDS_BANK_CONFLICT 0.40 of the LDS cycles.
constexpr int BN_CHUNKS = BN_PH * BN_PW * 8;      // 16-byte chunks of a patch (4896)
constexpr int BN_LOADS = (BN_CHUNKS + 511) / 512;  // per thread (10)
constexpr int BN_PATCH_BYTES = BN_PH * BN_PW * 128, BN_W_BYTES = 64 * BN_WP * 2, BN_SUM_BYTES = 8 * 2 * 64 * 4;
constexpr int BN_LDS = BN_PATCH_BYTES + BN_W_BYTES + BN_SUM_BYTES;  // 158 208 of the CU's 163 840

template <typename T>
struct BneckParams {
    const T *x;           // [N][H][W][64]: conv1's raw output
    const float *stats;   // [N][32][2]: (mean, rstd) of norm1 per (image, group)
    const T *gamma, *beta;  // norm1's affine parameters [64]
    const T *w;           // [64][3][3][64] = W[co][(ky, kx, ci)]: conv2's standardised weights
    T *out;               // [N][H][W][64]
    float *gn_partial;    // or nullptr: [tile][2][2][64] sums / sums of squares of the tile's (rounded) outputs
    int H, W, tiles_x, tiles_y, n_tiles;
};

// byte offset of 16-byte chunk c (0..7) of patch pixel q: the chunk index is XORed with q & 6.  A fragment read takes 16 consecutive pixels from ANY
// first pixel (the taps shift it by 0..2, the rows by 34) and its lane groups mix chunk fq with fq + 1 (see BN_WP): q & 6 is the XOR that keeps
// the 16 slots of the 256-byte bank row distinct for every first pixel (searched exhaustively; (q >> 1) & 7, right for first pixels that are
// multiples of 4 -- the GEMM tiles' case -- collides 2-4 times per group otherwise)
__device__ __forceinline__ int bn_swz(int q, int c) { return q * 128 + ((c ^ (q & 6)) << 4); }

__device__ __forceinline__ float bn_dot2(vec<__bf16, 2> a, vec<__bf16, 2> b, float c) { return __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false); }
__device__ __forceinline__ float bn_dot2(vec<_Float16, 2> a, vec<_Float16, 2> b, float c) { return __builtin_amdgcn_fdot2(a, b, c, false); }
__device__ __forceinline__ float bn_row_total(float v) {  // sum over the 16 lanes of a DPP row, left in every lane
#define HIVE_ROR_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
    HIVE_ROR_ADD(0x128);
    HIVE_ROR_ADD(0x124);
    HIVE_ROR_ADD(0x122);
    HIVE_ROR_ADD(0x121);
#undef HIVE_ROR_ADD
    return v;
}

template <typename T>
__global__ __launch_bounds__(512, 2) void bneck_conv3x3_kernel(BneckParams<T> p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *patch = lds;
    T *wl = reinterpret_cast<T *>(lds + BN_PATCH_BYTES);
    float(*wsum)[2][64] = reinterpret_cast<float(*)[2][64]>(lds + BN_PATCH_BYTES + BN_W_BYTES);
    // weights: 64 x 576, 16 bytes per thread and pass.  Output channel of accumulator row (nt, f4, e) -- row nt 16 + f4 4 + e of the tile in LDS:
    // (nt >> 1) 32 + f4 8 + (nt & 1) 4 + e, so that a lane's two accumulator columns are 8 consecutive channels (stem.hip): 16-byte stores on whole sectors
    for (int i = threadIdx.x; i < 64 * (BN_K / 8); i += 512) {
        const int row = i / (BN_K / 8), c8 = i - row * (BN_K / 8);
        const int nt = row >> 4, f4 = (row >> 2) & 3, e = row & 3;
        const int ch = (nt >> 1) * 32 + f4 * 8 + (nt & 1) * 4 + e;
        *reinterpret_cast<uint4 *>(wl + row * BN_WP + c8 * 8) = reinterpret_cast<const uint4 *>(p.w)[ch * (BN_K / 8) + c8];
    }
    const int per_img = p.tiles_x * p.tiles_y;
    const int my_c = threadIdx.x & 7;  // the 16-byte chunk (channels 8 my_c .. + 7) of every patch piece this thread stages (512 is a multiple of 8)
    float gam[8], bet[8];
    {
        const vec<T, 8> gv = *reinterpret_cast<const vec<T, 8> *>(p.gamma + my_c * 8), bv = *reinterpret_cast<const vec<T, 8> *>(p.beta + my_c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) gam[e] = (float)gv[e], bet[e] = (float)bv[e];
    }
    uint4 regs[BN_LOADS];
    auto tile_origin = [&](int tile, int &img, int &oy0, int &ox0) {
        img = tile / per_img;
        const int t = tile - img * per_img;
        oy0 = (t / p.tiles_x) * BN_TH;
        ox0 = (t % p.tiles_x) * BN_TW;
    };
    auto fetch = [&](int tile) {  // the tile's raw patch -> registers (pieces outside the image are not loaded)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));  // (opaque per call: no address arithmetic hoisted over the tile loop and spilled)
        int img, oy0, ox0;
        tile_origin(tile, img, oy0, ox0);
        const T *src = p.x + (size_t)img * p.H * p.W * 64;
#pragma unroll
        for (int k = 0; k < BN_LOADS; ++k) {
            const int i = tid + 512 * k, q = i >> 3, pr = q / BN_PW, pc = q - pr * BN_PW;
            const int gy = oy0 - 1 + pr, gx = ox0 - 1 + pc;
            regs[k] = make_uint4(0u, 0u, 0u, 0u);
            if (i < BN_CHUNKS && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
                regs[k] = *reinterpret_cast<const uint4 *>(src + ((size_t)gy * p.W + gx) * 64 + (i & 7) * 8);
        }
    };
    int tile = blockIdx.x;
    if (tile < p.n_tiles) fetch(tile);
    for (; tile < p.n_tiles; tile += gridDim.x) {
        int tix = threadIdx.x;
        asm volatile("" : "+v"(tix));  // (opaque per tile: see fetch)
        const int lane = tix & 63, wave = __builtin_amdgcn_readfirstlane(tix >> 6), fr = lane & 15, fq = lane >> 4;
        int img, oy0, ox0;
        tile_origin(tile, img, oy0, ox0);
        // norm1 for this image and this thread's 8 channels: y = t * a + b (gn_apply_kernel's operations, un-contracted)
        float a[8], b[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = (my_c * 8 + e) >> 1;  // 64 channels in 32 groups
            const float mean = p.stats[((size_t)img * 32 + g) * 2], rstd = p.stats[((size_t)img * 32 + g) * 2 + 1];
            a[e] = rstd * gam[e];
            b[e] = bet[e] - mean * a[e];
        }
        __syncthreads();  // everyone finished with the previous tile's patch (and, the first time, the weights are in LDS)
#pragma unroll
        for (int k = 0; k < BN_LOADS; ++k) {
            const int i = tix + 512 * k, q = i >> 3, pr = q / BN_PW, pc = q - pr * BN_PW;
            const int gy = oy0 - 1 + pr, gx = ox0 - 1 + pc;
            if (i < BN_CHUNKS) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);  // padding: zeros of the normalised tensor
                if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) {
                    const T *t = reinterpret_cast<const T *>(&regs[k]);
                    T r[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) r[e] = (T)fmaxf((float)t[e] * a[e] + b[e], 0.f);
                    v = *reinterpret_cast<const uint4 *>(r);
                }
                *reinterpret_cast<uint4 *>(patch + bn_swz(q, i & 7)) = v;
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < p.n_tiles) fetch(tile + gridDim.x);  // in flight during the MFMAs and the stores below
        // wave w: output rows 2 w, 2 w + 1 of the tile; m fragment mt: row 2 w + (mt >> 1), columns 16 (mt & 1) .. + 15
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                vec<T, 8> wf[4], af[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) wf[nt] = *reinterpret_cast<const vec<T, 8> *>(wl + (nt * 16 + fr) * BN_WP + tap * 64 + kc * 32 + fq * 8);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int q = (2 * wave + (mt >> 1) + ky) * BN_PW + 16 * (mt & 1) + fr + kx;
                    af[mt] = *reinterpret_cast<const vec<T, 8> *>(patch + bn_swz(q, kc * 4 + fq));
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[nt][mt] = hive_mfma::mfma16(wf[nt], af[mt], acc[nt][mt]);
            }
        }
        // a lane owns channels 32 a + 8 fq .. + 7 (a = 0, 1: accumulators nt = 2 a, 2 a + 1) of pixel (2 w + (mt >> 1), 16 (mt & 1) + fr)
        T *out = p.out + (size_t)img * p.H * p.W * 64;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int oy = oy0 + 2 * wave + (mt >> 1), ox = ox0 + 16 * (mt & 1) + fr;
            if (oy < p.H && ox < p.W) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    vec<T, 8> ov;
#pragma unroll
                    for (int j = 0; j < 4; ++j) ov[j] = (T)acc[2 * h][mt][j], ov[4 + j] = (T)acc[2 * h + 1][mt][j];
                    *reinterpret_cast<vec<T, 8> *>(out + ((size_t)oy * p.W + ox) * 64 + h * 32 + fq * 8) = ov;
                }
            }
        }
        if (p.gn_partial) {  // (maps whose width is whole tiles only: every column of the tile is a stored output; rows past the map are left out)
            vec<T, 2> ones, zero2;
            ones[0] = ones[1] = (T)1.0f;
            zero2[0] = zero2[1] = (T)0.0f;
            const bool row_ok[2] = {oy0 + 2 * wave < p.H, oy0 + 2 * wave + 1 < p.H};  // (wave-uniform)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                f32x4 sv = f32x4{0.f, 0.f, 0.f, 0.f}, qv = sv;
#pragma unroll
                for (int r = 0; r < 2; ++r)  // the two columns halves (mt = 2 r, 2 r + 1) of output row 2 w + r: two pixels of a channel per dot
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        vec<T, 2> y;
                        y[0] = (T)acc[nt][2 * r][e], y[1] = (T)acc[nt][2 * r + 1][e];
                        if (!row_ok[r]) y = zero2;
                        sv[e] = bn_dot2(y, ones, sv[e]);
                        qv[e] = bn_dot2(y, y, qv[e]);
                    }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sv[e] = bn_row_total(sv[e]);
                    qv[e] = bn_row_total(qv[e]);
                }
                if (fr == 0) {
                    const int ch = (nt >> 1) * 32 + fq * 8 + (nt & 1) * 4;
                    *reinterpret_cast<f32x4 *>(&wsum[wave][0][ch]) = sv;
                    *reinterpret_cast<f32x4 *>(&wsum[wave][1][ch]) = qv;
                }
            }
            __syncthreads();
            if (tix < 128) {
                const int which = tix >> 6, c = tix & 63;
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) s += wsum[w][which][c];
                p.gn_partial[((size_t)tile * 4 + which) * 64 + c] = s;        // h = 0: the tile's (only) image
                p.gn_partial[((size_t)tile * 4 + 2 + which) * 64 + c] = 0.f;  // h = 1
            }
        }
    }
}

template <typename T>
int launch_bneck(hive_ctx *ctx, const void *d_x, int N, int H, int W, const float *stats, const void *gamma, const void *beta, const void *d_w, void *d_out,
                 float *d_gn_partial, long long gn_partial_floats, int *gn_tile_rows) {
    static bool attr_set[64] = {};
    if (!(ctx->device < 64 && attr_set[ctx->device])) {
        HIVE_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)bneck_conv3x3_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, BN_LDS));
        if (ctx->device < 64) attr_set[ctx->device] = true;
    }
    BneckParams<T> p{};
    p.x = (const T *)d_x;
    p.stats = stats;
    p.gamma = (const T *)gamma;
    p.beta = (const T *)beta;
    p.w = (const T *)d_w;
    p.out = (T *)d_out;
    p.H = H;
    p.W = W;
    p.tiles_x = (W + BN_TW - 1) / BN_TW;
    p.tiles_y = (H + BN_TH - 1) / BN_TH;
    const long long tiles = (long long)N * p.tiles_x * p.tiles_y;
    HIVE_REQUIRE(ctx, tiles < (1ll << 31), "bneck_conv3x3: %lld tiles", tiles);
    p.n_tiles = (int)tiles;
    const int per_img = p.tiles_x * p.tiles_y;
    // sums: one row of partials per tile; hive_nhwc_group_norm_stats maps row t to image t * tile_rows / HW, so the "tile rows" reported are
    // HW / tiles-per-image (exact only when that divides; otherwise no sums and the GroupNorm behind makes its own pass)
    if (d_gn_partial && gn_tile_rows && W % BN_TW == 0 && (H * W) % per_img == 0 && tiles * 4 * 64 <= gn_partial_floats) {
        p.gn_partial = d_gn_partial;
        *gn_tile_rows = H * W / per_img;
    }
    const unsigned grid = (unsigned)std::min<long long>(tiles, (long long)ctx->num_cus);  // persistent, one workgroup (157 KB of LDS) per CU
    hipLaunchKernelGGL(bneck_conv3x3_kernel<T>, dim3(grid), dim3(512), BN_LDS, ctx->stream, p);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

}  // namespace

extern "C" int hive_bneck_gn_conv3x3(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, int C, const void *d_in_partial, int in_tile_rows,
                                     const void *d_gamma, const void *d_beta, float eps, const void *d_w, void *d_out, void *d_gn_partial,
                                     int64_t gn_partial_floats, int *gn_tile_rows, int *fused) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, d_x && d_gamma && d_beta && d_w && d_out && fused && d_out != d_x, "bneck_gn_conv3x3: bad pointers");
    HIVE_REQUIRE(ctx, dtype == HIVE_BF16 || dtype == HIVE_F16, "bneck_gn_conv3x3: dtype must be HIVE_F16 or HIVE_BF16");
    HIVE_REQUIRE(ctx, N > 0 && H > 0 && W > 0 && (long long)N * H * W < (1ll << 31), "bneck_gn_conv3x3: bad sizes %d x %d x %d", N, H, W);
    *fused = 0;
    if (gn_tile_rows) *gn_tile_rows = 0;
    // applies to the 64-channel bottlenecks whose first convolution left its GroupNorm sums (anything else: the caller runs the pair)
    if (C != 64 || !d_in_partial || in_tile_rows <= 0 || in_tile_rows > H * W) return HIVE_OK;
    int rc = hive_reserve_device(ctx, &ctx->d_scratch, &ctx->scratch_bytes, (size_t)N * 32 * 2 * sizeof(float));
    if (rc) return rc;
    float *stats = (float *)ctx->d_scratch;
    if ((rc = hive_gn_finalize_tiles(ctx, (const float *)d_in_partial, N, H * W, 64, 32, in_tile_rows, eps, stats))) return rc;
    rc = dtype == HIVE_BF16 ? launch_bneck<__bf16>(ctx, d_x, N, H, W, stats, d_gamma, d_beta, d_w, d_out, (float *)d_gn_partial, gn_partial_floats, gn_tile_rows)
                            : launch_bneck<_Float16>(ctx, d_x, N, H, W, stats, d_gamma, d_beta, d_w, d_out, (float *)d_gn_partial, gn_partial_floats, gn_tile_rows);
    if (rc) return rc;
    *fused = 1;
    return HIVE_OK;
}
