// Frames that are not the network's size (BASELINE config 4: 1920 x 1080 -> 864 x 480): the two resampling steps of
// /root/reference/hive/dataset_adaptors.py `estimate_depth_dpt` on the device, each fused with the elementwise step next to it.
//
//  * :1376-1389  dpt.transforms.Resize(640, 480, keep_aspect_ratio, multiple of 32, "minimal", cv2.INTER_CUBIC) on `image / 255.0`, then
//                NormalizeImage(0.5, 0.5), PrepareForNet and the cast to the network's 16-bit type (:1407-1417):
//                `hive_dpt_resize_preprocess`, uint8 frames in -> channels-last network input out.  The output SIZE is the caller's
//                (hive_amd.dpt.transforms.Resize.get_size restates the rule); this file restates cv2.resize's INTER_CUBIC arithmetic:
//                source coordinate fx = (float)((dx + 0.5) * (1 / (dst / src)) - 0.5), sx = floor(fx), the four taps sx - 1 .. sx + 2 with
//                indices clamped to the image (no anti-aliasing when shrinking), weights from the a = -0.75 cubic in float32
//                (`interpolateCubic`), the horizontal pass first, then the vertical one, each sum left to right.
//                cv2 is not in this image and the reference holds no fixture of it: parity with cv2 itself is UNPINNED; the C
//                oracle's restatement (oracle/__init__.py `resize_bicubic_cv2`) is pinned against torch's bicubic (same kernel, same
//                clamping, an independent implementation) and this kernel against the oracle.
//  * :1421-1426  torch.nn.functional.interpolate(prediction, size = frame size, mode = "nearest") followed by the uint16-millimetre
//                hand-off (:1432-1433 -> hive/io.py:1032-1039): `hive_depth_resize_nearest`.  Source index = min((int)floorf(dst * scale),
//                src - 1) with scale = (float)src / dst (ATen's nearest_neighbor_compute_source_index).
#include "hive_internal.hpp"

namespace {

struct CubicTaps {
    int idx[4];
    float w[4];
};

// cv2 resize.cpp: the taps of destination index d along an axis of `src` samples resized to `dst`
__device__ __forceinline__ CubicTaps cubic_taps(int d, int src, double scale) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    const int s = (int)floorf(f);
    f -= (float)s;
    const float A = -0.75f;
    CubicTaps t;
    t.w[0] = ((A * (f + 1.0f) - 5.0f * A) * (f + 1.0f) + 8.0f * A) * (f + 1.0f) - 4.0f * A;
    t.w[1] = ((A + 2.0f) * f - (A + 3.0f)) * f * f + 1.0f;
    t.w[2] = ((A + 2.0f) * (1.0f - f) - (A + 3.0f)) * (1.0f - f) * (1.0f - f) + 1.0f;
    t.w[3] = 1.0f - t.w[0] - t.w[1] - t.w[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) t.idx[k] = min(max(s - 1 + k, 0), src - 1);
    return t;
}

// one thread = one output pixel (3 channels); rgb u8 [B][H][W][3] -> out T [B][oh][ow][3]
template <typename T>
__global__ __launch_bounds__(256) void resize_preprocess_kernel(const uint8_t *__restrict__ rgb, int B, int H, int W, int oh, int ow, double scale_y, double scale_x,
                                                                float mean, float std, T *__restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * oh * ow;
    if (i >= total) return;
    const int ox = (int)(i % ow), oy = (int)((i / ow) % oh), b = (int)(i / ((long long)ow * oh));
    const CubicTaps tx = cubic_taps(ox, W, scale_x), ty = cubic_taps(oy, H, scale_y);
    const uint8_t *img = rgb + (size_t)b * H * W * 3;
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
        const uint8_t *row = img + (size_t)ty.idx[ky] * W * 3;
        float h[3];
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
            const uint8_t *px = row + (size_t)tx.idx[kx] * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = (float)px[c] * tx.w[kx];
                h[c] = kx == 0 ? v : h[c] + v;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = h[c] * ty.w[ky];
            acc[c] = ky == 0 ? v : acc[c] + v;
        }
    }
    T *o = out + (size_t)i * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = (T)((acc[c] * (1.0f / 255.0f) - mean) / std);
}

// depth f32 [B][h][w] -> nearest-neighbour [B][H][W], with the uint16-mm hand-off of the head's tail (vit.hip head_tail_kernel's arithmetic)
__global__ __launch_bounds__(256) void resize_nearest_handoff_kernel(const float *__restrict__ depth, int B, int h, int w, int H, int W, float scale_y, float scale_x,
                                                                     float depth_scale, float max_depth, float *__restrict__ out_depth,
                                                                     uint16_t *__restrict__ out_mm, float *__restrict__ out_m) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * H * W;
    if (i >= total) return;
    const int x = (int)(i % W), y = (int)((i / W) % H), b = (int)(i / ((long long)W * H));
    const int sy = min((int)floorf((float)y * scale_y), h - 1), sx = min((int)floorf((float)x * scale_x), w - 1);
    const float d = depth[((size_t)b * h + sy) * w + sx];
    if (out_depth) out_depth[i] = d;
    if (out_mm || out_m) {
        const uint16_t mm = (uint16_t)(int)fminf(fmaxf(d * 1000.0f, 0.0f), 65535.0f);
        float m = depth_scale * (float)mm;
        if (m > max_depth) m = 0.0f;
        if (out_mm) out_mm[i] = mm;
        if (out_m) out_m[i] = m;
    }
}

}  // namespace

extern "C" {

int hive_dpt_resize_preprocess(hive_ctx *ctx, const uint8_t *d_rgb, int B, int H, int W, int out_h, int out_w, float mean, float std, int dtype, void *d_out) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, d_rgb && d_out && std != 0.f, "dpt_resize_preprocess: bad arguments");
    HIVE_REQUIRE(ctx, B > 0 && H > 0 && W > 0 && out_h > 0 && out_w > 0 && (long long)B * out_h * out_w < (1ll << 40) && H < (1 << 20) && W < (1 << 20),
                 "dpt_resize_preprocess: bad sizes %d x %d x %d -> %d x %d", B, H, W, out_h, out_w);
    // cv2: inv_scale = dst / src (double), scale = 1 / inv_scale
    const double scale_x = 1.0 / ((double)out_w / (double)W), scale_y = 1.0 / ((double)out_h / (double)H);
    const long long total = (long long)B * out_h * out_w;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == HIVE_BF16)
        hipLaunchKernelGGL(resize_preprocess_kernel<__bf16>, grid, dim3(256), 0, ctx->stream, d_rgb, B, H, W, out_h, out_w, scale_y, scale_x, mean, std, (__bf16 *)d_out);
    else if (dtype == HIVE_F16)
        hipLaunchKernelGGL(resize_preprocess_kernel<_Float16>, grid, dim3(256), 0, ctx->stream, d_rgb, B, H, W, out_h, out_w, scale_y, scale_x, mean, std, (_Float16 *)d_out);
    else if (dtype == HIVE_F32)
        hipLaunchKernelGGL(resize_preprocess_kernel<float>, grid, dim3(256), 0, ctx->stream, d_rgb, B, H, W, out_h, out_w, scale_y, scale_x, mean, std, (float *)d_out);
    else
        return hive_fail(ctx, HIVE_ERR_INVALID, "dpt_resize_preprocess: unknown dtype %d", dtype);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

int hive_depth_resize_nearest(hive_ctx *ctx, const float *d_depth, int B, int h, int w, int H, int W, float depth_scale, float max_depth, float *d_out_depth,
                              uint16_t *d_out_mm, float *d_out_m) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, d_depth && (d_out_depth || d_out_mm || d_out_m), "depth_resize_nearest: NULL argument");
    HIVE_REQUIRE(ctx, B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && (long long)B * H * W < (1ll << 40), "depth_resize_nearest: bad sizes %d x %d x %d -> %d x %d", B, h, w, H,
                 W);
    const float scale_y = (float)h / (float)H, scale_x = (float)w / (float)W;  // ATen compute_scales_value<float>: (float)src / dst
    const long long total = (long long)B * H * W;
    hipLaunchKernelGGL(resize_nearest_handoff_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, d_depth, B, h, w, H, W, scale_y, scale_x, depth_scale,
                       max_depth, d_out_depth, d_out_mm, d_out_m);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

}  // extern "C"
