// ResNetV2 stem of the DPT-Hybrid backbone on gfx950: the 7 x 7 stride-2 convolution (3 -> 64 channels, weight-standardised,
// TensorFlow "SAME" padding) as an MFMA implicit GEMM straight from the channels-last bf16 frame, and the 3 x 3 stride-2
// "SAME" max pool.  timm 0.5.4 `ResNetV2.stem` (StdConv2dSame 7 x 7 / 2, GroupNormAct, MaxPool2dSame 3 x 3 / 2) reached from
// `DPTDepthModel.forward` (call site /root/reference/hive/dataset_adaptors.py:1419).
//
// Convolution: K = 7 x 7 x 3 = 147 does not tile into 64-channel slices like the other convolutions (csrc/conv.hip), but with
// channels-last input the 21 values (kx, c) of one kernel ROW are contiguous in the frame, so K is laid out as 7 k-steps of 32
// (21 taps + 11 zero weights): the A fragment of a pixel for k-step ky is 32 consecutive bf16 of input row 2 oy + ky - pad starting
// at column 2 ox - pad (the 11 extra values belong to the neighbouring pixels and meet zero weights).  A workgroup owns an 8 x 32
// output tile: its 21 x 69 x 3 input patch and the 64 x 224 weight matrix sit in LDS; four waves x (4 x 4) accumulators of
// v_mfma_f32_16x16x32_bf16.  HBM-bound: 1.8 MB in, 9.8 MB out per 480 x 640 frame.  hive_resnet_stem_conv_gn also leaves the sums the
// GroupNorm behind the convolution needs (per 256-pixel tile, from the accumulators: as conv.hip's gn_sums_from_acc).
#include "hive_internal.hpp"

#include <algorithm>

#include "mfma_pipe.hpp"

using hive_mfma::f32x4;
using hive_mfma::vec;  // T = __bf16 or _Float16 (the reference's model.half())

namespace {

#ifndef HIVE_STEM_ABLATE
#define HIVE_STEM_ABLATE 0  // tuning builds: 1 no output stores, 2 no patch loads, 4 no MFMAs
#endif
constexpr int ST_TH = 8, ST_TW = 32;                      // output tile
constexpr int ST_PH = 2 * ST_TH + 5, ST_PW = 2 * ST_TW + 5;  // input patch 21 x 69 pixels
constexpr int ST_ROW = 240;                               // patch row pitch in elements (69 x 3 = 207, + slack for the 32-wide k-steps; 480 B)
constexpr int ST_RD = 104;                                // dwords of a patch row that are loaded (208 elements >= 207)
constexpr int ST_K = 7 * 32;                              // padded K
constexpr int ST_WP = ST_K + 16;                          // weight row pitch in LDS: 480 B = 30 chunks of 16 B (= 2 mod 4: conflict-free over the lane groups a
                                                          // ds_read_b128 is really served in -- bneck.hip BN_WP; round 3's 29 chunks collided 5 of 16 times)
constexpr int ST_RQ = ST_RD / 4;                          // 16-byte quads of a patch row (26)
constexpr int ST_LOADS = (ST_PH * ST_RQ + 255) / 256;     // 16-byte loads per thread and patch (3)

template <typename T>
struct StemParams {
    const T *x;   // [N][H][W][3]
    const T *w;   // [64][7][32]: (ky, (kx, c) padded from 21 to 32 with zeros)
    T *out;       // [N][Ho][Wo][64]
    int H, W, Ho, Wo, pad_t, pad_l;
    int tiles_x, tiles_y, n_tiles;
    int run;            // consecutive tiles a workgroup takes at a time (and sums before it writes a row of gn_partial): divides tiles_x * tiles_y
    float *gn_partial;  // or nullptr: [tile][2][2][64] sums / sums of squares of the tile's (rounded) outputs -- the layout of conv.hip's
                        // GroupNorm partials with 256-pixel "tiles" (hive_nhwc_group_norm_stats reads them); only whole tiles (Ho % 8 == 0, Wo % 32 == 0)
};

__device__ __forceinline__ float stem_dot2(vec<__bf16, 2> a, vec<__bf16, 2> b, float c) { return __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false); }
__device__ __forceinline__ float stem_dot2(vec<_Float16, 2> a, vec<_Float16, 2> b, float c) { return __builtin_amdgcn_fdot2(a, b, c, false); }
__device__ __forceinline__ float stem_row_total(float v) {  // sum over the 16 lanes of a DPP row, left in every lane
#define HIVE_ROR_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
    HIVE_ROR_ADD(0x128);
    HIVE_ROR_ADD(0x124);
    HIVE_ROR_ADD(0x122);
    HIVE_ROR_ADD(0x121);
#undef HIVE_ROR_ADD
    return v;
}

// PERSISTENT workgroups (as many as fit: 3 per CU by LDS): the weights go to LDS once; per tile the patch of the NEXT tile is loaded into
// registers (dwords: a patch row is 414 contiguous bytes of the frame, 4-byte aligned when W and the left padding are even -- `fast`;
// otherwise, and for tiles that touch the frame's border, element by element with the bounds tests) while the MFMAs and the stores of the
// current one run.  Round 3: 729 -> see DESIGN 5.6 (one workgroup per tile, weights re-read per tile, 2-byte patch loads, 4-way LDS
// conflicts on the weight fragments).
template <typename T>
__global__ __launch_bounds__(256, 3) void stem_conv_kernel(StemParams<T> p) {  // 3 waves per SIMD: three workgroups per CU (41 KB of LDS each)
    __shared__ __attribute__((aligned(16))) T patch[ST_PH * ST_ROW + 64];
    __shared__ __attribute__((aligned(16))) T wl[64 * ST_WP];
    __shared__ float wsum[4][2][64];
    const int tid = threadIdx.x;
    // Output channel of accumulator row (nt, fq, e) -- row nt 16 + fq 4 + e of the weight tile in LDS: (nt >> 1) 32 + fq 8 + (nt & 1) 4 + e.
    // With this permutation of the weight rows a lane holds 8 CONSECUTIVE channels per fragment pair (nt = 2 a, 2 a + 1) and the four
    // lanes fq of a pixel 64 contiguous bytes: the tile leaves in 8 stores of 16 bytes per lane on whole 64-byte sectors instead of 16
    // of 8 bytes on half sectors (the kernel is bound by its vector-memory instructions: tools/probe_stem.py with the tuning builds).
    for (int i = tid; i < 64 * ST_K / 8; i += 256) {  // weights: 64 x 224 = 28 KiB, 16 bytes per thread and pass
        const int row = i / (ST_K / 8), c8 = i - row * (ST_K / 8);
        const int nt = row >> 4, f4 = (row >> 2) & 3, e = row & 3;
        const int ch = (nt >> 1) * 32 + f4 * 8 + (nt & 1) * 4 + e;
        *reinterpret_cast<uint4 *>(wl + row * ST_WP + c8 * 8) = reinterpret_cast<const uint4 *>(p.w)[ch * (ST_K / 8) + c8];
    }
    if (tid < 64) patch[ST_PH * ST_ROW + tid] = (T)0.0f;
    // elements 208 .. 239 of every patch row are never loaded but are read (against zero weights): zero, not whatever LDS held
    for (int i = tid; i < ST_PH * (ST_ROW / 2 - ST_RD); i += 256) {
        const int r = i / (ST_ROW / 2 - ST_RD), d = i - r * (ST_ROW / 2 - ST_RD);
        reinterpret_cast<uint32_t *>(patch)[r * (ST_ROW / 2) + ST_RD + d] = 0u;
    }
    const bool even = (p.W % 2 == 0) && (p.pad_l % 2 == 0);  // dword-aligned patch rows
    const int per_img = p.tiles_x * p.tiles_y;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u4u __attribute__((ext_vector_type(4), aligned(4)));  // global_load_dwordx4 needs dword alignment only
    u4 regs[ST_LOADS];
    auto fetch = [&](int tile) {  // the tile's patch -> registers: quad i = row i / 26, dwords 4 (i % 26) .. + 3 (elements 8 (i % 26) .. + 7)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));  // (opaque per call: see the tile loop)
        const int img = tile / per_img, t = tile - img * per_img;
        const int oy0 = (t / p.tiles_x) * ST_TH, ox0 = (t % p.tiles_x) * ST_TW;
        const int iy0 = 2 * oy0 - p.pad_t, ix0 = 2 * ox0 - p.pad_l;
        const T *xin = p.x + (size_t)img * p.H * p.W * 3;
        const bool inside = iy0 >= 0 && iy0 + ST_PH <= p.H && ix0 >= 0 && ix0 + (2 * ST_RD + 2) / 3 <= p.W;  // (workgroup-uniform)
        if (HIVE_STEM_ABLATE & 2) {
#pragma unroll
            for (int k = 0; k < ST_LOADS; ++k) regs[k] = u4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
        } else if (even && inside) {
            const uint32_t *base = reinterpret_cast<const uint32_t *>(xin + ((size_t)iy0 * p.W + ix0) * 3);
            const int row_dwords = p.W * 3 / 2;
#pragma unroll
            for (int k = 0; k < ST_LOADS; ++k) {
                const int i = tid + 256 * k, r = i / ST_RQ, d = i - r * ST_RQ;
                regs[k] = u4{0u, 0u, 0u, 0u};
                if (k + 1 < ST_LOADS || i < ST_PH * ST_RQ) regs[k] = *reinterpret_cast<const u4u *>(base + (size_t)r * row_dwords + 4 * d);
            }
        } else {
#pragma unroll
            for (int k = 0; k < ST_LOADS; ++k) {
                const int i = tid + 256 * k, r = i / ST_RQ, d = i - r * ST_RQ;
                T v[8];
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    const int e = 8 * d + h, col = e / 3, c = e - col * 3;
                    const int iy = iy0 + r, ix = ix0 + col;
                    v[h] = (T)0.0f;
                    if (r < ST_PH && e < ST_PW * 3 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) v[h] = xin[((size_t)iy * p.W + ix) * 3 + c];
                }
                regs[k] = *reinterpret_cast<const u4 *>(v);
            }
        }
    };
    auto flush_sums = [&](int run_idx) {  // the four waves' sums of a run of tiles (in wsum since its last tile's MFMAs) -> its row of gn_partial
        if (tid < 128) {
            const int which = tid >> 6, c = tid & 63;
            const float a = ((wsum[0][which][c] + wsum[1][which][c]) + wsum[2][which][c]) + wsum[3][which][c];
            p.gn_partial[((size_t)run_idx * 4 + which) * 64 + c] = a;       // h = 0: the run's (only) image
            p.gn_partial[((size_t)run_idx * 4 + 2 + which) * 64 + c] = 0.f;  // h = 1
        }
    };
    // tiles in runs of p.run consecutive ones (same image); the workgroup walks runs blockIdx.x, + gridDim.x, ...
    const int n_runs = p.n_tiles / p.run;
    int run_idx = blockIdx.x, in_run = 0, pending = -1;
    if (run_idx < n_runs) fetch(run_idx * p.run);
    f32x4 rs[4], rq[4];  // this lane's running sums over the run: channels of accumulator rows (nt, fq, 0..3), its pixels
    for (; run_idx < n_runs;) {
        const int tile = run_idx * p.run + in_run;
        const bool last_of_run = in_run + 1 == p.run;
        const int next_tile = last_of_run ? (run_idx + (int)gridDim.x) * p.run : tile + 1;
        // the thread index made opaque per tile: otherwise every lane-constant below (LDS and store addresses) is computed once in front of
        // the loop and spilled (40 registers at three waves per SIMD)
        int tix = tid;
        asm volatile("" : "+v"(tix));
        const int lane = tix & 63, wave = __builtin_amdgcn_readfirstlane(tix >> 6), fr = lane & 15, fq = lane >> 4;
        __syncthreads();  // everyone finished with the previous tile's patch (and, the first time, the weights are in LDS)
#pragma unroll
        for (int k = 0; k < ST_LOADS; ++k) {
            const int i = tix + 256 * k, r = i / ST_RQ, d = i - r * ST_RQ;
            if (k + 1 < ST_LOADS || i < ST_PH * ST_RQ) *reinterpret_cast<u4 *>(reinterpret_cast<uint32_t *>(patch) + r * (ST_ROW / 2) + 4 * d) = regs[k];
        }
        if (pending >= 0) {  // between the two barriers: wsum is complete and nobody writes it yet
            flush_sums(pending);
            pending = -1;
        }
        __syncthreads();
        if (next_tile < p.n_tiles) fetch(next_tile);  // in flight during the MFMAs and the stores below
        const int img = tile / per_img, t = tile - img * per_img;
        const int oy0 = (t / p.tiles_x) * ST_TH, ox0 = (t % p.tiles_x) * ST_TW;
        // wave w: output rows 2 w, 2 w + 1 of the tile; m fragment mt: row 2 w + (mt >> 1), columns 16 (mt & 1) .. + 15
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1  // (unrolled, the compiler reads all seven kernel rows' fragments ahead: 247 registers, or 82 spilled at three waves per SIMD)
        for (int ky = 0; ky < 7; ++ky) {
            vec<T, 8> wf[4], af[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) wf[nt] = *reinterpret_cast<const vec<T, 8> *>(wl + (nt * 16 + fr) * ST_WP + ky * 32 + fq * 8);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int oy = 2 * wave + (mt >> 1), ox = 16 * (mt & 1) + fr;
                const T *src = patch + (2 * oy + ky) * ST_ROW + 6 * ox + fq * 8;  // 4-byte aligned: 12 ox + 16 fq bytes
                uint32_t raw[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) raw[q] = reinterpret_cast<const uint32_t *>(src)[q];
                af[mt] = *reinterpret_cast<const vec<T, 8> *>(raw);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    if (HIVE_STEM_ABLATE & 4)
                        acc[nt][mt][0] += (float)wf[nt][0] + (float)af[mt][0];
                    else
                        acc[nt][mt] = hive_mfma::mfma16(wf[nt], af[mt], acc[nt][mt]);
                }
        }
        // a lane owns channels 32 a + 8 fq .. + 7 (a = 0, 1: accumulators nt = 2 a, 2 a + 1) of pixel (2 w + (mt >> 1), 16 (mt & 1) + fr)
        T *out = p.out + (size_t)img * p.Ho * p.Wo * 64;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int oy = oy0 + 2 * wave + (mt >> 1), ox = ox0 + 16 * (mt & 1) + fr;
            if (oy < p.Ho && ox < p.Wo && (!(HIVE_STEM_ABLATE & 1) || acc[0][mt][0] == 12345.678f)) {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    vec<T, 8> ov;
#pragma unroll
                    for (int j = 0; j < 4; ++j) ov[j] = (T)acc[2 * a][mt][j], ov[4 + j] = (T)acc[2 * a + 1][mt][j];
                    *reinterpret_cast<vec<T, 8> *>(out + ((size_t)oy * p.Wo + ox) * 64 + a * 32 + fq * 8) = ov;
                }
            }
        }
        if (p.gn_partial) {  // (whole tiles only: every accumulator is a stored output)
            vec<T, 2> ones;
            ones[0] = ones[1] = (T)1.0f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                if (in_run == 0) rs[nt] = rq[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int mt = 0; mt < 4; mt += 2)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        vec<T, 2> y;
                        y[0] = (T)acc[nt][mt][e], y[1] = (T)acc[nt][mt + 1][e];
                        rs[nt][e] = stem_dot2(y, ones, rs[nt][e]);
                        rq[nt][e] = stem_dot2(y, y, rq[nt][e]);
                    }
            }
            if (last_of_run) {  // (workgroup-uniform) the 16 pixel lanes of a row, then -- behind the next barrier -- the four waves
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    f32x4 sv, qv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        sv[e] = stem_row_total(rs[nt][e]);
                        qv[e] = stem_row_total(rq[nt][e]);
                    }
                    if (fr == 0) {
                        const int ch = (nt >> 1) * 32 + fq * 8 + (nt & 1) * 4;
                        *reinterpret_cast<f32x4 *>(&wsum[wave][0][ch]) = sv;
                        *reinterpret_cast<f32x4 *>(&wsum[wave][1][ch]) = qv;
                    }
                }
                pending = run_idx;
            }
        }
        if (last_of_run) {
            run_idx += gridDim.x;
            in_run = 0;
        } else {
            ++in_run;
        }
    }
    if (pending >= 0) {
        __syncthreads();
        flush_sums(pending);
    }
}

// 3 x 3 stride-2 max pool with explicit top / left padding (padded positions never win: -inf), channels-last, 8 channels per lane
template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const T *__restrict__ x, T *__restrict__ out, int N, int H, int W, int C, int Ho,
                                                           int Wo, int pad_t, int pad_l) {
    const int c8 = C / 8;
    const long long total = (long long)N * Ho * Wo * c8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cc = (int)(i % c8);
        const long long pix = i / c8;
        const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), n = (int)(pix / ((long long)Wo * Ho));
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = 2 * oy + ky - pad_t, ix = 2 * ox + kx - pad_l;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                    const vec<T, 8> v = *reinterpret_cast<const vec<T, 8> *>(x + (((size_t)n * H + iy) * W + ix) * C + cc * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], (float)v[j]);
                }
            }
        vec<T, 8> o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (T)m[j];
        *reinterpret_cast<vec<T, 8> *>(out + (size_t)pix * C + cc * 8) = o;
    }
}

template <typename T>
int launch_stem(hive_ctx *ctx, const void *d_x, int N, int H, int W, const void *d_w, void *d_out, float *d_gn_partial, long long gn_partial_floats,
                int *gn_tile_rows) {
    StemParams<T> p{};
    p.x = (const T *)d_x;
    p.w = (const T *)d_w;
    p.out = (T *)d_out;
    p.H = H;
    p.W = W;
    p.Ho = (H + 1) / 2;
    p.Wo = (W + 1) / 2;
    p.pad_t = std::max((p.Ho - 1) * 2 + 7 - H, 0) / 2;  // TensorFlow "SAME": the odd pixel goes to the bottom / right
    p.pad_l = std::max((p.Wo - 1) * 2 + 7 - W, 0) / 2;
    p.tiles_x = (p.Wo + ST_TW - 1) / ST_TW;
    p.tiles_y = (p.Ho + ST_TH - 1) / ST_TH;
    const long long tiles = (long long)N * p.tiles_x * p.tiles_y;
    HIVE_REQUIRE(ctx, tiles < (1ll << 31), "resnet_stem_conv: %lld tiles", tiles);
    p.n_tiles = (int)tiles;
    if (gn_tile_rows) *gn_tile_rows = 0;
    const int per_img = p.tiles_x * p.tiles_y;
    p.run = per_img % 4 == 0 ? 4 : (per_img % 2 == 0 ? 2 : 1);  // the sums' cross-lane reduction once per run: it cost a quarter of the kernel per tile
    if (d_gn_partial && gn_tile_rows && p.Ho % ST_TH == 0 && p.Wo % ST_TW == 0) {  // whole tiles: a run is 256 * run pixels of ONE image
        HIVE_REQUIRE(ctx, tiles / p.run * 4 * 64 <= gn_partial_floats, "resnet_stem_conv_gn: gn_partial holds %lld floats, %lld needed", gn_partial_floats,
                     tiles / p.run * 4 * 64);
        p.gn_partial = d_gn_partial;
        *gn_tile_rows = ST_TH * ST_TW * p.run;
    }
    const unsigned grid = (unsigned)std::min<long long>(tiles / p.run, (long long)ctx->num_cus * 3);  // persistent: 3 workgroups of 41 KB LDS per CU
    hipLaunchKernelGGL(stem_conv_kernel<T>, dim3(grid), dim3(256), 0, ctx->stream, p);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

}  // namespace

extern "C" {

static int stem_entry(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, const void *d_w, void *d_out, void *d_gn_partial, int64_t gn_partial_floats,
                      int *gn_tile_rows) {
    HIVE_REQUIRE(ctx, d_x && d_w && d_out, "resnet_stem_conv: NULL argument");
    HIVE_REQUIRE(ctx, dtype == HIVE_BF16 || dtype == HIVE_F16, "resnet_stem_conv: dtype must be HIVE_F16 or HIVE_BF16");
    HIVE_REQUIRE(ctx, N > 0 && H >= 7 && W >= 7 && (long long)N * H * W < (1ll << 31), "resnet_stem_conv: bad sizes %d x %d x %d", N, H, W);
    return dtype == HIVE_BF16 ? launch_stem<__bf16>(ctx, d_x, N, H, W, d_w, d_out, (float *)d_gn_partial, gn_partial_floats, gn_tile_rows)
                              : launch_stem<_Float16>(ctx, d_x, N, H, W, d_w, d_out, (float *)d_gn_partial, gn_partial_floats, gn_tile_rows);
}

int hive_resnet_stem_conv(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, const void *d_w, void *d_out) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    return stem_entry(ctx, d_x, dtype, N, H, W, d_w, d_out, nullptr, 0, nullptr);
}

int hive_resnet_stem_conv_gn(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, const void *d_w, void *d_out, void *d_gn_partial,
                             int64_t gn_partial_floats, int *gn_tile_rows) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, d_gn_partial && gn_tile_rows, "resnet_stem_conv_gn: NULL argument");
    return stem_entry(ctx, d_x, dtype, N, H, W, d_w, d_out, d_gn_partial, gn_partial_floats, gn_tile_rows);
}

int hive_nhwc_maxpool3x3s2(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, int C, void *d_out) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, d_x && d_out && d_x != d_out, "nhwc_maxpool3x3s2: bad pointers");
    HIVE_REQUIRE(ctx, dtype == HIVE_BF16 || dtype == HIVE_F16, "nhwc_maxpool3x3s2: dtype must be HIVE_F16 or HIVE_BF16");
    HIVE_REQUIRE(ctx, N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "nhwc_maxpool3x3s2: bad sizes %d x %d x %d x %d", N, H, W, C);
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int pad_t = std::max((Ho - 1) * 2 + 3 - H, 0) / 2, pad_l = std::max((Wo - 1) * 2 + 3 - W, 0) / 2;
    const long long total = (long long)N * Ho * Wo * (C / 8);
    const int blocks = (int)std::min<long long>((total + 255) / 256, (long long)ctx->num_cus * 32);
    if (dtype == HIVE_BF16)
        hipLaunchKernelGGL(maxpool3x3s2_kernel<__bf16>, dim3(blocks), dim3(256), 0, ctx->stream, (const __bf16 *)d_x, (__bf16 *)d_out, N, H, W, C, Ho, Wo, pad_t, pad_l);
    else
        hipLaunchKernelGGL(maxpool3x3s2_kernel<_Float16>, dim3(blocks), dim3(256), 0, ctx->stream, (const _Float16 *)d_x, (_Float16 *)d_out, N, H, W, C, Ho, Wo, pad_t, pad_l);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

}  // extern "C"
