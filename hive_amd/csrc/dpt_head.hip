// Depth head of DPT, second half, as ONE kernel (isl-org/DPT `DPTDepthModel.scratch.output_conv[1:]`, reached from
// dpt.models.DPTDepthModel.forward -- /root/reference/hive/dataset_adaptors.py:1419):
//
//   Interpolate(x2, bilinear, align_corners=True) -> Conv3x3(128 -> 32) + bias -> ReLU -> Conv1x1(32 -> 1) + bias
//   -> ReLU -> depth = 1 / max(scale * x + shift, 1e-8) -> uint16-mm hand-off
//
// Unfused this is the largest activation of the network: the upsampled [B][480][640][128] map (1.26 GB at
// B = 16) written once and read once, a 362-GFLOP convolution with only 32 output channels (MIOpen: 1.08 ms,
// 335 TFLOP/s) and a pass over its output.  Fused, the kernel reads the [B][240][320][128] input (79 MB) and
// writes 6 bytes per pixel; it is bound by MFMA issue / LDS operand bandwidth:
//
//   * workgroup = 8 x 16 output pixels, 4 waves, persistent over the tile list;
//   * the low-resolution patch (7 x 11 pixels) is staged in LDS by LDS-DMA, one tile AHEAD (it lands under the previous tile's MFMA
//     and epilogue phases); the 10 x 18 upsampled patch (halo for the 3x3 taps, zero outside the image) is built from it once, rounded
//     to the 16-bit type exactly as the stand-alone upsampling kernel rounds its output;
//   * D^T[oc][px] = sum_tap W_tap[oc][ic] X_tap[ic][px] on v_mfma_f32_32x32x16_bf16: output channels are the rows
//     (A operand), pixels the columns (B operand), so a lane ends up with 16 of the 32 channels of ONE pixel and
//     the 1x1 convolution is a register sum plus one cross-half shuffle;
//   * the 128 input channels are split over the 4 waves (32 each): a wave keeps its 18 weight fragments (9 taps
//     x 2 k-steps) in registers for the whole kernel and reads only pixel operands from LDS -- 1 KiB per MFMA,
//     which is the LDS bandwidth of a CU at full MFMA rate, so weights must not come from LDS as well;
//     the 4 partial sums per pixel tile are exchanged through LDS once per tile;
//   * tried in round 3 and taken out again: producer / consumer waves (one workgroup of eight waves per CU: four build the next tile's
//     patch into a second buffer while four multiply the current one, K split two ways, 160,000 bytes of LDS) -- correct, but 4.1 ms
//     against 3.7: the patch construction is bound by what ONE wave issues (conversions, un-contracted multiplies and adds), so four
//     producer waves (one per SIMD) need 3.5 ms for what eight waves of two co-resident workgroups do in 1.5; reading the 14 vectors of a
//     column pair up front instead of row by row changes nothing either (not LDS latency).  What did help: compiling this file without
//     SLP vectorisation (-fno-slp-vectorize, csrc/Makefile): the packed f32 forms the vectoriser makes of the interpolation arithmetic
//     issue slowly beside the other workgroup's MFMAs (3.69 -> 3.53 ms; same values, IEEE per element);
//   * LDS layout of the upsampled patch: 272 bytes per pixel (256 + 16) and a row pitch that is a multiple of 256 bytes,
//     so the 16-byte bank slot of a pixel depends on its column only; the lane groups in which ds_read_b128 is serviced
//     ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...) each hold 16 different columns of the 2 x 16 pixel block: conflict-free
//     (with the natural pitch 18 x 272 the two rows of a group overlapped: SQ_LDS_BANK_CONFLICT 26 % of the LDS cycles).
#include "hive_internal.hpp"
#include "mfma_pipe.hpp"

#include <algorithm>

using hive_mfma::f32x16;
using hive_mfma::vec;  // T = __bf16 or _Float16 (the reference's model.half()): v_mfma_f32_32x32x16_{bf16,f16}

// tuning builds only (make -C hive_amd/csrc ablate): bit mask of phases left out of head_conv_kernel, to read what each costs from
// tools/probe_head.py (1 = low-resolution patch load, 2 = upsampled patch, 4 = MFMA loop, 8 = partial-sum exchange).  0 in the library.
#ifndef HIVE_HEAD_ABLATE
#define HIVE_HEAD_ABLATE 0
#endif

namespace {

constexpr int TH = 8, TW = 16;                 // output tile
constexpr int UP_H = TH + 2, UP_W = TW + 2;    // upsampled patch with the 3x3 halo
constexpr int LO_H = 7, LO_W = 11;             // low-resolution patch that feeds it (scale < 0.5)
constexpr int CIN = 128, COUT = 32;
constexpr int PIX = 272;                       // bytes per pixel of the upsampled patch in LDS
constexpr int UP_PITCH = 5120;                 // bytes per row of the upsampled patch: >= UP_W * PIX (4896) and a multiple of 256
constexpr int UP_BYTES = UP_H * UP_PITCH;      // 51200, also >= the partial-sum exchange (4*3*16*64*4 = 49152)
constexpr int LO_PIX = 256;                    // bytes per pixel of the low-resolution patch: lane-linear, as the LDS-DMA deposits it
constexpr int LO_ITEMS = LO_H * LO_W * 16;     // its 16-byte pieces: 1232
constexpr int LO_DMA = (LO_ITEMS + 63) / 64;   // LDS-DMA wave-instructions per patch: 20 (the last one 16 lanes of payload)
constexpr int LO_BYTES = LO_DMA * 1024;        // 20480
static_assert(UP_W * PIX <= UP_PITCH && UP_PITCH % 256 == 0 && UP_BYTES >= 49152, "upsampled patch layout");

template <typename T>
struct HeadParams {
    const T *x;      // [N][H][W][128]
    const T *w3;     // [9][32][128]  (tap = ky * 3 + kx, output channel, input channel)
    float b3[COUT];     // bias of the 3x3 convolution
    float w1[COUT];     // 1x1 convolution
    const float *b0;    // device, [128]: bias of the convolution that produced x, added on load (or null)
    float b1, scale, shift;
    int non_negative, invert;
    int N, H, W;        // low-resolution input; output is 2H x 2W
    float depth_scale, max_depth;
    float *out_depth;   // [N][2H][2W] or null
    uint16_t *out_mm;   // or null
    float *out_m;       // or null
};

template <typename T>
__device__ __forceinline__ vec<T, 8> lds_read8(const unsigned char *base, int byte_off) {
    return *reinterpret_cast<const vec<T, 8> *>(base + byte_off);
}

template <typename T>
__global__ __launch_bounds__(256, 2) void head_conv_kernel(HeadParams<T> p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *up = smem;             // upsampled patch, later the partial-sum exchange
    unsigned char *lo = smem + UP_BYTES;  // low-resolution patch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nn = lane & 31, hh = lane >> 5;
    const int OH = 2 * p.H, OW = 2 * p.W;
    const int tiles_x = (OW + TW - 1) / TW, tiles_y = (OH + TH - 1) / TH;
    const int n_tiles = p.N * tiles_y * tiles_x;  // < 2^31: checked on the host
    const float sh = OH > 1 ? (float)(p.H - 1) / (float)(OH - 1) : 0.f;
    const float sw = OW > 1 ? (float)(p.W - 1) / (float)(OW - 1) : 0.f;

    // this wave's weight fragments: input channels [32 * wave, 32 * wave + 32), lane = (output channel nn, k half hh)
    vec<T, 8> wf[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            wf[t][ks] = *reinterpret_cast<const vec<T, 8> *>(p.w3 + ((size_t)(t * COUT + nn) * CIN + 32 * wave + 16 * ks + 8 * hh));
    // epilogue constants (bias of the 3x3 convolution, 1x1 weights) live in LDS: 32 registers less in the MFMA loop
    float *tab = reinterpret_cast<float *>(smem + UP_BYTES + LO_BYTES);
    if (tid < COUT) {
        tab[tid] = p.b3[tid];
        tab[COUT + tid] = p.w1[tid];
    }

    // (a) low-resolution patch -> LDS by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B = 4 pixels per wave-instruction, lane-linear:
    // pixel q = item >> 4 at q * 256 bytes).  Rows / columns past the image edge are clamped duplicates: exactly what align_corners'
    // i1 = min(i0 + 1, size - 1) reads.  The patch of the NEXT tile is issued right behind the barrier that ends phase (b) -- from there
    // on nobody reads `lo` -- and lands under the MFMA, exchange and epilogue phases: the ~1 ms (of 5.1 at 96 frames) the kernel used to
    // wait for these loads at the top of every tile is gone, at no cost in registers (a register prefetch spilled) or LDS.
    auto issue_lo = [&](int tile) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int lo_y0 = (int)(sh * (float)max(ty * TH - 1, 0)), lo_x0 = (int)(sw * (float)max(tx * TW - 1, 0));
        const T *img = p.x + (size_t)n * p.H * p.W * CIN;
#pragma unroll
        for (int j = 0; j < LO_DMA / 4; ++j) {
            const int piece = __builtin_amdgcn_readfirstlane(wave) + 4 * j;         // wave-uniform: the LDS base of the piece goes to M0
            const int item = min(piece * 64 + lane, LO_ITEMS - 1);                  // lanes past the patch: its last piece again (inside LO_BYTES)
            const int v = item & 15, q = item >> 4;
            const int gy = min(lo_y0 + q / LO_W, p.H - 1), gx = min(lo_x0 + q % LO_W, p.W - 1);
            __builtin_amdgcn_global_load_lds((const void *)(img + ((size_t)gy * p.W + gx) * CIN + v * 8), (__attribute__((address_space(3))) void *)(lo + piece * 1024), 16, 0,
                                             0);
        }
    };
    static_assert(LO_DMA % 4 == 0, "pieces per wave");
    if (!(HIVE_HEAD_ABLATE & 1) && (int)blockIdx.x < n_tiles) issue_lo(blockIdx.x);

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int tx = tile % tiles_x;
        const int ty = (tile / tiles_x) % tiles_y;
        const int n = tile / (tiles_x * tiles_y);
        const int R0 = ty * TH - 1, C0 = tx * TW - 1;  // output coordinates of the patch origin
        const int lo_y0 = (int)(sh * (float)max(R0, 0)), lo_x0 = (int)(sw * (float)max(C0, 0));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the patch have landed
        __syncthreads();                                   // ... and everyone's
        if (p.b0) {  // x + bias rounded to T, as a separate bias add rounds it (the network passes NULL: output_conv[0] adds its bias itself)
            for (int item = tid; item < LO_ITEMS; item += 256) {
                vec<T, 8> xv = lds_read8<T>(lo, item * 16);
                const int v = item & 15;
                const float4 ba = *reinterpret_cast<const float4 *>(p.b0 + v * 8), bb = *reinterpret_cast<const float4 *>(p.b0 + v * 8 + 4);
                const float bias8[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[j] = (T)((float)xv[j] + bias8[j]);
                *reinterpret_cast<vec<T, 8> *>(lo + item * 16) = xv;
            }
            __syncthreads();
        }
        // (b) upsampled patch: PyTorch's align_corners=True formula in float, evaluated as upsample2x_kernel does,
        //   y = h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11).
        // Pass 1, columns 0 .. 15 (all 256 threads): a thread owns one (patch column, 8-channel vector) and walks down the rows, so the
        // horizontal terms t(r) = w0 * v(r, x0) + w1 * v(r, x1) of a low-resolution row r are computed once and kept in registers for the
        // two or three output rows that use them (same values, 43 % fewer flops and LDS reads than per pixel).
        // Pass 2, the two halo columns 16, 17: 32 (column, vector) items -- walked like pass 1 they kept half a wave busy for as long as all
        // of pass 1 (this phase was HALF of the kernel's time: ablation builds, tools/probe_head.py); as 320 independent (item, row) units they
        // take 1 - 2 short steps per thread.
        const uint4 zero = make_uint4(0u, 0u, 0u, 0u);  // zero padding of the convolution
        if (!(HIVE_HEAD_ABLATE & 2)) {
            const int v = tid & 15, ux = tid >> 4, ox = C0 + ux;
            const bool col_ok = ox >= 0 && ox < OW;
            const float fx = sw * (float)ox;
            const int x0 = (int)fx, x1 = min(x0 + 1, p.W - 1);
            const float w1 = fx - (float)x0, w0 = 1.f - w1;
            const int c0 = col_ok ? (x0 - lo_x0) * LO_PIX + v * 16 : 0, c1 = col_ok ? (x1 - lo_x0) * LO_PIX + v * 16 : 0;
            unsigned char *dst = up + ux * PIX + v * 16;  // + uy * UP_PITCH
            int uy = 0;                                      // patch row to emit next (uniform over the workgroup)
            while (uy < UP_H && R0 + uy < 0) {               // rows above the image
                *reinterpret_cast<uint4 *>(dst + uy * UP_PITCH) = zero;
                ++uy;
            }
            // all 14 reads of the column pair first (56 registers, free in this phase): one LDS round trip for the walk instead of seven
            vec<T, 8> col0[LO_H], col1[LO_H];
#pragma unroll
            for (int r = 0; r < LO_H; ++r) col0[r] = lds_read8<T>(lo, r * LO_W * LO_PIX + c0), col1[r] = lds_read8<T>(lo, r * LO_W * LO_PIX + c1);
            float ta[8], tb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) ta[j] = w0 * (float)col0[0][j] + w1 * (float)col1[0][j];
#pragma unroll
            for (int r = 0; r < LO_H - 1; ++r) {  // low-resolution rows r, r + 1 (clamped duplicates past the image edge)
#pragma unroll
                for (int j = 0; j < 8; ++j) tb[j] = w0 * (float)col0[r + 1][j] + w1 * (float)col1[r + 1][j];
                while (uy < UP_H && R0 + uy < OH) {  // the output rows whose upper source row is r: at most three
                    const float fy = sh * (float)(R0 + uy);
                    const int y0 = (int)fy;
                    if (__builtin_amdgcn_readfirstlane(y0 - lo_y0) != r) break;
                    const float h1 = fy - (float)y0, h0 = 1.f - h1;
                    uint4 packed = zero;
                    if (col_ok) {
                        vec<T, 8> o;
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = (T)(h0 * ta[j] + h1 * tb[j]);
                        packed = *reinterpret_cast<const uint4 *>(&o);
                    }
                    *reinterpret_cast<uint4 *>(dst + uy * UP_PITCH) = packed;
                    ++uy;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) ta[j] = tb[j];
            }
            while (uy < UP_H) {  // rows below the image
                *reinterpret_cast<uint4 *>(dst + uy * UP_PITCH) = zero;
                ++uy;
            }
            for (int u = tid; u < 2 * 16 * UP_H; u += 256) {  // pass 2: unit = (halo column, vector, patch row)
                const int v2 = u & 15, ux2 = 16 + ((u >> 4) & 1), uy2 = u >> 5, ox2 = C0 + ux2, oy2 = R0 + uy2;
                uint4 packed = zero;
                if (ox2 < OW && oy2 >= 0 && oy2 < OH) {  // (ox2 >= 15 > 0)
                    const float fx2 = sw * (float)ox2, fy2 = sh * (float)oy2;
                    const int xa = (int)fx2, xb = min(xa + 1, p.W - 1), ya = (int)fy2;
                    const float q1 = fx2 - (float)xa, q0 = 1.f - q1, g1 = fy2 - (float)ya, g0 = 1.f - g1;
                    const int ra = (ya - lo_y0) * LO_W * LO_PIX, ca = (xa - lo_x0) * LO_PIX + v2 * 16, cb = (xb - lo_x0) * LO_PIX + v2 * 16;
                    const vec<T, 8> a00 = lds_read8<T>(lo, ra + ca), a01 = lds_read8<T>(lo, ra + cb);
                    const vec<T, 8> a10 = lds_read8<T>(lo, ra + LO_W * LO_PIX + ca), a11 = lds_read8<T>(lo, ra + LO_W * LO_PIX + cb);
                    vec<T, 8> o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (T)(g0 * (q0 * (float)a00[j] + q1 * (float)a01[j]) + g1 * (q0 * (float)a10[j] + q1 * (float)a11[j]));
                    packed = *reinterpret_cast<const uint4 *>(&o);
                }
                *reinterpret_cast<uint4 *>(up + uy2 * UP_PITCH + ux2 * PIX + v2 * 16) = packed;
            }
        }
        __syncthreads();
        // nobody reads `lo` any more: the next tile's patch streams in under phases (c), (d) and the epilogue
        if (!(HIVE_HEAD_ABLATE & 1) && tile + (int)gridDim.x < n_tiles) issue_lo(tile + gridDim.x);
        // (c) 9 taps x 2 k-steps x 4 pixel tiles of 32 (tile m = output rows 2m, 2m+1 of the 8 x 16 patch)
        f32x16 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[m][v] = 0.f;
        const int pix_base = (nn >> 4) * UP_PITCH + (nn & 15) * PIX + (32 * wave + 8 * hh) * 2;
#pragma unroll
        for (int t = 0; t < ((HIVE_HEAD_ABLATE & 4) ? 0 : 9); ++t) {
            const int tap_off = (t / 3) * UP_PITCH + (t % 3) * PIX;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const vec<T, 8> xf = lds_read8<T>(up, pix_base + tap_off + 2 * m * UP_PITCH + ks * 32);
                    acc[m] = hive_mfma::mfma32(wf[t][ks], xf, acc[m]);
                }
        }
        __syncthreads();  // every wave is done reading the patch: its space becomes the exchange buffer
        // (d) wave w owns pixel tile w: the other three waves hand it their channel-partial sums
        typedef hive_mfma::f32x4 f32x4;
        f32x4 *part = reinterpret_cast<f32x4 *>(up);
#pragma unroll
        for (int m = 0; m < ((HIVE_HEAD_ABLATE & 8) ? 0 : 4); ++m)
            if (wave != m) {
                const int slot = wave < m ? wave : wave - 1;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    part[((m * 3 + slot) * 4 + g) * 64 + lane] = f32x4{acc[m][4 * g], acc[m][4 * g + 1], acc[m][4 * g + 2], acc[m][4 * g + 3]};
            }
        __syncthreads();
        f32x16 mine = wave == 0 ? acc[0] : wave == 1 ? acc[1] : wave == 2 ? acc[2] : acc[3];
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // registers 4g .. 4g+3 = output channels 8g + 4hh .. + 3 of pixel nn
            f32x4 sum = f32x4{mine[4 * g], mine[4 * g + 1], mine[4 * g + 2], mine[4 * g + 3]};
#pragma unroll
            for (int slot = 0; slot < ((HIVE_HEAD_ABLATE & 8) ? 0 : 3); ++slot) sum += part[((wave * 3 + slot) * 4 + g) * 64 + lane];
            const f32x4 b3q = *reinterpret_cast<const f32x4 *>(tab + 8 * g + 4 * hh);
            const f32x4 w1q = *reinterpret_cast<const f32x4 *>(tab + COUT + 8 * g + 4 * hh);
#pragma unroll
            for (int e = 0; e < 4; ++e) s += fmaxf(sum[e] + b3q[e], 0.f) * w1q[e];
        }
        s += __shfl_xor(s, 32);
        const int oy = ty * TH + 2 * wave + (nn >> 4), ox = tx * TW + (nn & 15);
        if (hh == 0 && oy < OH && ox < OW) {
            float acc1 = s + p.b1;
            if (p.non_negative) acc1 = fmaxf(acc1, 0.f);
            float depth = acc1;
            if (p.invert) depth = 1.0f / fmaxf(p.scale * acc1 + p.shift, 1e-8f);
            const size_t o = ((size_t)n * OH + oy) * OW + ox;
            if (p.out_depth) p.out_depth[o] = depth;
            if (p.out_mm || p.out_m) {
                const uint16_t mm = (uint16_t)(int)fminf(fmaxf(depth * 1000.0f, 0.0f), 65535.0f);
                float m = p.depth_scale * (float)mm;
                if (m > p.max_depth) m = 0.0f;
                if (p.out_mm) p.out_mm[o] = mm;
                if (p.out_m) p.out_m[o] = m;
            }
        }
        __syncthreads();  // exchange buffer and low-resolution patch are free again
    }
}

template <typename T>
int launch_head(hive_ctx *ctx, const void *d_x, const float *d_b0, int N, int H, int W, const void *d_w3, const float *h_b3, const float *h_w1, float b1,
                int non_negative, int invert, float scale, float shift, float *d_depth, float depth_scale, float max_depth, uint16_t *d_out_mm, float *d_out_m) {
    HeadParams<T> p;
    p.x = (const T *)d_x;
    p.w3 = (const T *)d_w3;
    p.b0 = d_b0;
    for (int i = 0; i < COUT; ++i) {
        p.b3[i] = h_b3[i];
        p.w1[i] = h_w1[i];
    }
    p.b1 = b1;
    p.scale = scale;
    p.shift = shift;
    p.non_negative = non_negative;
    p.invert = invert;
    p.N = N;
    p.H = H;
    p.W = W;
    p.depth_scale = depth_scale;
    p.max_depth = max_depth;
    p.out_depth = d_depth;
    p.out_mm = d_out_mm;
    p.out_m = d_out_m;
    static bool attr_set[64] = {};
    const int lds = UP_BYTES + LO_BYTES + 2 * COUT * (int)sizeof(float);
    if (ctx->device >= 64 || !attr_set[ctx->device]) {
        HIVE_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)head_conv_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        if (ctx->device < 64) attr_set[ctx->device] = true;
    }
    const long long tiles = (long long)N * ((2 * H + TH - 1) / TH) * ((2 * W + TW - 1) / TW);
    const dim3 grid((unsigned)std::min<long long>(tiles, (long long)ctx->num_cus * 2));
    hipLaunchKernelGGL(head_conv_kernel<T>, grid, dim3(256), lds, ctx->stream, p);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

}  // namespace

extern "C" int hive_dpt_head_fused(hive_ctx *ctx, const void *d_x, const float *d_b0, int dtype, int N, int H, int W, int C_in, int C_mid,
                                   const void *d_w3, const float *h_b3, const float *h_w1, float b1, int non_negative, int invert,
                                   float scale, float shift, float *d_depth, float depth_scale, float max_depth,
                                   uint16_t *d_out_mm, float *d_out_m) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, d_x && d_w3 && h_b3 && h_w1, "dpt_head_fused: NULL argument");
    HIVE_REQUIRE(ctx, dtype == HIVE_BF16 || dtype == HIVE_F16, "dpt_head_fused: dtype must be HIVE_F16 or HIVE_BF16");
    HIVE_REQUIRE(ctx, C_in == CIN && C_mid == COUT, "dpt_head_fused: built for 128 -> 32 channels, got %d -> %d", C_in, C_mid);
    HIVE_REQUIRE(ctx, N > 0 && H > 1 && W > 1 && (long long)N * H * W < (1ll << 28), "dpt_head_fused: bad shape N=%d H=%d W=%d", N, H, W);
    HIVE_REQUIRE(ctx, d_depth || d_out_mm || d_out_m, "dpt_head_fused: no output requested");
    if (dtype == HIVE_BF16)
        return launch_head<__bf16>(ctx, d_x, d_b0, N, H, W, d_w3, h_b3, h_w1, b1, non_negative, invert, scale, shift, d_depth, depth_scale, max_depth, d_out_mm, d_out_m);
    return launch_head<_Float16>(ctx, d_x, d_b0, N, H, W, d_w3, h_b3, h_w1, b1, non_negative, invert, scale, shift, d_depth, depth_scale, max_depth, d_out_mm, d_out_m);
}
