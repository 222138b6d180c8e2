// View frustum, depth <-> point-cloud (un)projection, mask dilation, depth hand-off.  gfx950 only.
//
//   hive_view_frustum   -> fusion.get_view_frustum            (call site hive/fusion.py:59)
//   hive_unproject      -> point_cloud_from_depth / _from_rgbd (hive/geometric.py:107-152, image2world :183-206)
//   hive_project        -> world2image                         (hive/geometric.py:155-180)
//   hive_dilate_mask    -> dilate_mask                         (hive/image_processing.py:30-45)
//   hive_depth_quantize -> uint16-mm PNG round trip            (hive/dataset_adaptors.py:1432-1433, hive/io.py:1032-1039)
//
// The geometric functions are float64 in the reference, and so are these kernels; the stream
// compaction keeps np.nonzero's row-major (v,u) order through a block count -> scan -> write scheme
// (wave shuffles inside a block).
#include "hive_internal.hpp"

#include <algorithm>
#include <cmath>

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void max_depth_kernel(const float *__restrict__ depth, int n, unsigned *max_bits) {
    unsigned bits = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float d = depth[i];
        if (d > 0.f) bits = max(bits, __float_as_uint(d));
    }
    for (int off = 32; off > 0; off >>= 1) bits = max(bits, (unsigned)__shfl_xor((int)bits, off));
    if ((threadIdx.x & 63) == 0 && bits) atomicMax(max_bits, bits);
}

// one launch for a whole frame set: blockIdx.y = frame
__global__ __launch_bounds__(256) void max_depth_batch_kernel(const float *__restrict__ depth, int n_px, unsigned *max_bits) {
    const float *d = depth + (size_t)blockIdx.y * n_px;
    unsigned bits = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_px; i += gridDim.x * 256) {
        const float v = d[i];
        if (v > 0.f) bits = max(bits, __float_as_uint(v));
    }
    for (int off = 32; off > 0; off >>= 1) bits = max(bits, (unsigned)__shfl_xor((int)bits, off));
    if ((threadIdx.x & 63) == 0 && bits) atomicMax(max_bits + blockIdx.y, bits);
}

// ------------------------------------------------------------------------------------------------
constexpr int UP_VPT = 4;
constexpr int UP_TILE = 256 * UP_VPT;

__device__ __forceinline__ bool px_valid(const float *depth, const uint8_t *mask, int i) {
    return (!mask || mask[i]) && depth[i] > 0.0f;
}

__global__ __launch_bounds__(256) void unproject_count_kernel(const float *__restrict__ depth, const uint8_t *__restrict__ mask,
                                                              int n, unsigned *__restrict__ blk) {
    __shared__ unsigned lds[4];
    const int base = blockIdx.x * UP_TILE + threadIdx.x * UP_VPT;
    unsigned c = 0;
    for (int j = 0; j < UP_VPT; ++j)
        if (base + j < n && px_valid(depth, mask, base + j)) ++c;
    for (int off = 32; off > 0; off >>= 1) c += (unsigned)__shfl_xor((int)c, off);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

__global__ __launch_bounds__(1024) void scan1_kernel(unsigned *__restrict__ a, int nb, unsigned long long *total) {
    __shared__ unsigned long long part[1024];
    const int t = threadIdx.x;
    const int per = (nb + 1023) / 1024;
    const int lo = min(t * per, nb), hi = min(lo + per, nb);
    unsigned long long s = 0;
    for (int i = lo; i < hi; ++i) s += a[i];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        unsigned long long r = 0;
        for (int i = 0; i < 1024; ++i) {
            const unsigned long long v = part[i];
            part[i] = r;
            r += v;
        }
        *total = r;
    }
    __syncthreads();
    unsigned long long r = part[t];
    for (int i = lo; i < hi; ++i) {
        const unsigned v = a[i];
        a[i] = (unsigned)r;
        r += v;
    }
}

struct UnprojectParams {
    double Kinv[9], R[9], t[3];
    int H, W;
    long long capacity;
};

__global__ __launch_bounds__(256) void unproject_write_kernel(const float *__restrict__ depth, const uint8_t *__restrict__ mask,
                                                              const uint8_t *__restrict__ rgb, UnprojectParams p,
                                                              const unsigned *__restrict__ blk, double *__restrict__ out_xyz,
                                                              uint8_t *__restrict__ out_rgba) {
    __shared__ unsigned lds[4];
    const int n = p.H * p.W;
    const int base = blockIdx.x * UP_TILE + threadIdx.x * UP_VPT;
    bool ok[UP_VPT];
    unsigned c = 0;
#pragma unroll
    for (int j = 0; j < UP_VPT; ++j) {
        ok[j] = base + j < n && px_valid(depth, mask, base + j);
        c += ok[j];
    }
    // exclusive scan over the block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = c;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = (unsigned)__shfl_up((int)inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    unsigned before = 0;
    for (int w = 0; w < wave; ++w) before += lds[w];
    long long id = (long long)blk[blockIdx.x] + before + inc - c;
#pragma unroll
    for (int j = 0; j < UP_VPT; ++j) {
        if (!ok[j]) continue;
        const int i = base + j;
        if (id < p.capacity) {
            const double d = (double)depth[i];
            const double pu = (double)(i % p.W), pv = (double)(i / p.W);
            double cam[3];
            for (int r = 0; r < 3; ++r) cam[r] = d * (p.Kinv[3 * r + 0] * pu + p.Kinv[3 * r + 1] * pv + p.Kinv[3 * r + 2]) - p.t[r];
            for (int r = 0; r < 3; ++r) out_xyz[3 * id + r] = p.R[0 * 3 + r] * cam[0] + p.R[1 * 3 + r] * cam[1] + p.R[2 * 3 + r] * cam[2];
            if (rgb && out_rgba) {
                out_rgba[4 * id + 0] = rgb[3 * i + 0];
                out_rgba[4 * id + 1] = rgb[3 * i + 1];
                out_rgba[4 * id + 2] = rgb[3 * i + 2];
                out_rgba[4 * id + 3] = 255;
            }
        }
        ++id;
    }
}

struct ProjectParams {
    double K[9], R[9], t[3];
    double scale;
};

template <int RM>
__global__ __launch_bounds__(256) void project_kernel(const double *__restrict__ pts, long long n, ProjectParams p,
                                                      int32_t *__restrict__ uv_i, double *__restrict__ uv_f,
                                                      double *__restrict__ out_depth) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double X[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    double cam[3], c[3];
    for (int r = 0; r < 3; ++r) cam[r] = p.R[3 * r + 0] * X[0] + p.R[3 * r + 1] * X[1] + p.R[3 * r + 2] * X[2] + p.t[r];
    for (int r = 0; r < 3; ++r) c[r] = p.K[3 * r + 0] * cam[0] + p.K[3 * r + 1] * cam[1] + p.K[3 * r + 2] * cam[2];
    const double u = c[0] / c[2] / p.scale;
    const double v = c[1] / c[2] / p.scale;
    if (out_depth) out_depth[i] = c[2];
    if (uv_i) {
        uv_i[2 * i + 0] = (int32_t)(RM ? round(u) : rint(u));
        uv_i[2 * i + 1] = (int32_t)(RM ? round(v) : rint(v));
    }
    if (uv_f) {
        uv_f[2 * i + 0] = u;
        uv_f[2 * i + 1] = v;
    }
}

// world2image + the visibility reduction of HiveDataset.select_key_frames (hive/io.py:1161-1175): project,
// round (np.round), keep pixels inside [0, W) x [0, H), and reduce them to (min u, max u, min v, max v, count).
// out[0..4] must be initialised to {INT_MAX, INT_MIN, INT_MAX, INT_MIN, 0}.
__global__ __launch_bounds__(256) void project_bbox_kernel(const double *__restrict__ pts, long long n, ProjectParams p, int W, int H,
                                                           int *__restrict__ out) {
    int mn_u = 0x7fffffff, mx_u = (int)0x80000000, mn_v = 0x7fffffff, mx_v = (int)0x80000000, cnt = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double X[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
        double cam[3], c[3];
        for (int r = 0; r < 3; ++r) cam[r] = p.R[3 * r + 0] * X[0] + p.R[3 * r + 1] * X[1] + p.R[3 * r + 2] * X[2] + p.t[r];
        for (int r = 0; r < 3; ++r) c[r] = p.K[3 * r + 0] * cam[0] + p.K[3 * r + 1] * cam[1] + p.K[3 * r + 2] * cam[2];
        const int u = (int)rint(c[0] / c[2] / p.scale), v = (int)rint(c[1] / c[2] / p.scale);
        if (u >= 0 && u < W && v >= 0 && v < H) {
            mn_u = min(mn_u, u);
            mx_u = max(mx_u, u);
            mn_v = min(mn_v, v);
            mx_v = max(mx_v, v);
            ++cnt;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        mn_u = min(mn_u, __shfl_xor(mn_u, off));
        mx_u = max(mx_u, __shfl_xor(mx_u, off));
        mn_v = min(mn_v, __shfl_xor(mn_v, off));
        mx_v = max(mx_v, __shfl_xor(mx_v, off));
        cnt += __shfl_xor(cnt, off);
    }
    if ((threadIdx.x & 63) == 0 && cnt) {
        atomicMin(out + 0, mn_u);
        atomicMax(out + 1, mx_u);
        atomicMin(out + 2, mn_v);
        atomicMax(out + 3, mx_v);
        atomicAdd(out + 4, cnt);
    }
}

// image2world for an explicit list of pixel coordinates (hive/geometric.py:183-206)
__global__ __launch_bounds__(256) void image2world_kernel(const double *__restrict__ uv, const double *__restrict__ depth, long long n,
                                                          UnprojectParams p, double scale, double *__restrict__ out_xyz) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double d = depth[i];
    const double pu = uv[2 * i] * scale, pv = uv[2 * i + 1] * scale;
    double cam[3];
    for (int r = 0; r < 3; ++r) cam[r] = d * (p.Kinv[3 * r + 0] * pu + p.Kinv[3 * r + 1] * pv + p.Kinv[3 * r + 2]) - p.t[r];
    for (int r = 0; r < 3; ++r) out_xyz[3 * i + r] = p.R[0 * 3 + r] * cam[0] + p.R[1 * 3 + r] * cam[1] + p.R[2 * 3 + r] * cam[2];
}

// separable (2r+1) box max with out-of-image pixels ignored
__global__ __launch_bounds__(256) void dilate_rows_kernel(const uint8_t *__restrict__ in, int H, int W, int r, uint8_t *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const int v = i / W, u = i % W;
    uint8_t m = 0;
    for (int uu = max(0, u - r); uu <= min(W - 1, u + r); ++uu) m |= (in[v * W + uu] != 0);
    out[i] = m;
}

__global__ __launch_bounds__(256) void dilate_cols_kernel(const uint8_t *__restrict__ in, int H, int W, int r, uint8_t *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const int v = i / W, u = i % W;
    uint8_t m = 0;
    for (int vv = max(0, v - r); vv <= min(H - 1, v + r); ++vv) m |= in[vv * W + u];
    out[i] = m;
}

// frame-set versions of the two passes (blockIdx.y = frame; no bleeding across frame borders), and the masking itself:
// mode 0 (background, hive/fusion.py:118-121): depth = 0 where the dilated mask is set;
// mode 1 (foreground = the complement): depth = 0 where the (undilated) mask is clear or is another instance's.
__global__ __launch_bounds__(256) void dilate_rows_batch_kernel(const uint8_t *__restrict__ in, int H, int W, int r, int instance,
                                                                uint8_t *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const size_t base = (size_t)blockIdx.y * H * W;
    const int v = i / W, u = i % W;
    uint8_t m = 0;
    for (int uu = max(0, u - r); uu <= min(W - 1, u + r); ++uu) {
        const uint8_t s = in[base + (size_t)v * W + uu];
        m |= instance ? (s == instance) : (s != 0);
    }
    out[base + i] = m;
}

__global__ __launch_bounds__(256) void dilate_cols_apply_kernel(const uint8_t *__restrict__ rows, int H, int W, int r,
                                                                const float *__restrict__ depth, float *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const size_t base = (size_t)blockIdx.y * H * W;
    const int v = i / W, u = i % W;
    uint8_t m = 0;
    for (int vv = max(0, v - r); vv <= min(H - 1, v + r); ++vv) m |= rows[base + (size_t)vv * W + u];
    out[base + i] = m ? 0.0f : depth[base + i];
}

// cv2.dilate with an ARBITRARY structuring element (hive/options.py:245-268 lets the caller pass any `dilation_filter`), one iteration:
// out(v, u) = OR over the set taps (j, i) of in(v + j - kh / 2, u + i - kw / 2); taps that fall outside the image contribute nothing
// (cv2's default border for dilation).  blockIdx.y = frame.  `instance` applies to the FIRST iteration's input (instance ids -> set / clear).
struct StructuringElement {
    int kh, kw;
    uint8_t m[32 * 32];
};
__global__ __launch_bounds__(256) void dilate_se_kernel(const uint8_t *__restrict__ in, int H, int W, StructuringElement se, int instance, int first,
                                                        uint8_t *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const size_t base = (size_t)blockIdx.y * H * W;
    const int v = i / W, u = i % W;
    const int ay = se.kh / 2, ax = se.kw / 2;
    uint8_t m = 0;
    for (int j = 0; j < se.kh; ++j) {
        const int vv = v + j - ay;
        if (vv < 0 || vv >= H) continue;
        for (int k = 0; k < se.kw; ++k) {
            const int uu = u + k - ax;
            if (!se.m[j * se.kw + k] || uu < 0 || uu >= W) continue;
            const uint8_t s = in[base + (size_t)vv * W + uu];
            m |= (first && instance) ? (s == instance) : (s != 0);
        }
    }
    out[base + i] = m;
}

__global__ __launch_bounds__(256) void zero_under_mask_kernel(const uint8_t *__restrict__ mask, size_t n, const float *__restrict__ depth, float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = mask[i] ? 0.0f : depth[i];
}

__global__ __launch_bounds__(256) void keep_mask_kernel(const uint8_t *__restrict__ mask, size_t n, int instance,
                                                        const float *__restrict__ depth, float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint8_t s = mask[i];
    const bool keep = instance ? (s == instance) : (s != 0);
    out[i] = keep ? depth[i] : 0.0f;
}

__global__ __launch_bounds__(256) void depth_mm_to_m_kernel(const uint16_t *__restrict__ mm, size_t n, float depth_scale, float max_depth,
                                                            float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float m = depth_scale * (float)mm[i];
    if (m > max_depth) m = 0.0f;
    out[i] = m;
}

template <typename T>
__device__ __forceinline__ float load_depth(const void *p, int i);
template <>
__device__ __forceinline__ float load_depth<float>(const void *p, int i) {
    return ((const float *)p)[i];
}
template <>
__device__ __forceinline__ float load_depth<_Float16>(const void *p, int i) {
    return (float)((const _Float16 *)p)[i];
}
template <>
__device__ __forceinline__ float load_depth<unsigned short>(const void *p, int i) {  // bf16
    return __uint_as_float((unsigned)((const unsigned short *)p)[i] << 16);
}

template <typename T>
__global__ __launch_bounds__(256) void depth_quantize_kernel(const void *__restrict__ in, int n, float depth_scale, float max_depth,
                                                             const uint8_t *__restrict__ mask, uint16_t *__restrict__ out_mm,
                                                             float *__restrict__ out_m) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float mm_f = load_depth<T>(in, i) * 1000.0f;
    // astype(np.uint16) of a float: truncation towards zero (values are within [0, 65535])
    const uint16_t mm = (uint16_t)(int)fminf(fmaxf(mm_f, 0.0f), 65535.0f);
    float m = depth_scale * (float)mm;
    if (m > max_depth) m = 0.0f;
    if (mask && mask[i]) m = 0.0f;
    if (out_mm) out_mm[i] = mm;
    if (out_m) out_m[i] = m;
}

// ------------------------------------------------------------------------------------------------
static int to_device(hive_ctx *ctx, const void *src, size_t bytes, size_t offset, int mem, const void **out) {
    if (!src) {
        *out = nullptr;
        return HIVE_OK;
    }
    if (mem == HIVE_MEM_DEVICE) {
        *out = src;
        return HIVE_OK;
    }
    int rc = hive_upload(ctx, (char *)ctx->d_in + offset, src, bytes);
    *out = (char *)ctx->d_in + offset;
    return rc;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// fusion.get_view_frustum: apex + the four image corners at max(depth), camera -> world.  15 float64 values from one
// scalar: evaluated on the host, in the reference library's float64
static void frustum_corners(float max_depth, int H, int W, const float K[9], const double cam_pose[16], double out[15]) {
    const double md = (double)max_depth;
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const double col[5] = {0, 0, 0, (double)W, (double)W};
    const double row[5] = {0, 0, (double)H, 0, (double)H};
    const double dep[5] = {0, md, md, md, md};
    for (int c = 0; c < 5; ++c) {
        const double p[3] = {(col[c] - cx) * dep[c] / fx, (row[c] - cy) * dep[c] / fy, dep[c]};
        for (int r = 0; r < 3; ++r)
            out[r * 5 + c] = cam_pose[4 * r + 0] * p[0] + cam_pose[4 * r + 1] * p[1] + cam_pose[4 * r + 2] * p[2] + cam_pose[4 * r + 3];
    }
}

extern "C" {

int hive_view_frustum(hive_ctx *ctx, const float *depth, int H, int W, const float K[9], const double cam_pose[16], int mem,
                      double out[15]) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, depth && K && cam_pose && out, "view_frustum: NULL argument");
    HIVE_REQUIRE(ctx, H > 0 && W > 0, "view_frustum: bad image size %dx%d", H, W);
    const int n = H * W;
    const void *d_depth;
    int rc;
    if (mem == HIVE_MEM_HOST && (rc = hive_reserve_device(ctx, &ctx->d_in, &ctx->in_bytes, (size_t)n * 4))) return rc;
    if ((rc = to_device(ctx, depth, (size_t)n * 4, 0, mem, &d_depth))) return rc;
    unsigned *d_max = ctx->d_scalars + 16;
    HIVE_CHECK_HIP(ctx, hipMemsetAsync(d_max, 0, 4, ctx->stream));
    hipLaunchKernelGGL(max_depth_kernel, dim3(std::min((n + 255) / 256, 1024)), dim3(256), 0, ctx->stream, (const float *)d_depth, n,
                       d_max);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    unsigned bits = 0;
    HIVE_CHECK_HIP(ctx, hipMemcpyAsync(&bits, d_max, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIVE_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float max_depth;
    memcpy(&max_depth, &bits, 4);
    frustum_corners(max_depth, H, W, K, cam_pose, out);
    return HIVE_OK;
}

int hive_view_frustum_batch(hive_ctx *ctx, const float *depth, int n, int H, int W, const float K[9], const double *cam_poses, int mem,
                            double *out) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, depth && K && cam_poses && out, "view_frustum_batch: NULL argument");
    HIVE_REQUIRE(ctx, n > 0 && n <= 65535 && H > 0 && W > 0 && (long long)H * W < (1ll << 30), "view_frustum_batch: bad sizes n=%d %dx%d", n, H, W);
    const int n_px = H * W;
    const size_t bytes = (size_t)n * n_px * sizeof(float);
    const void *d_depth;
    int rc;
    if (mem == HIVE_MEM_HOST && (rc = hive_reserve_device(ctx, &ctx->d_in, &ctx->in_bytes, bytes))) return rc;
    if ((rc = to_device(ctx, depth, bytes, 0, mem, &d_depth))) return rc;
    // per-frame maxima in the generic scratch (n words), ONE launch and ONE read-back for the whole frame set
    if ((rc = hive_reserve_device(ctx, &ctx->d_scratch, &ctx->scratch_bytes, (size_t)n * sizeof(unsigned)))) return rc;
    unsigned *d_max = (unsigned *)ctx->d_scratch;
    HIVE_CHECK_HIP(ctx, hipMemsetAsync(d_max, 0, (size_t)n * sizeof(unsigned), ctx->stream));
    hipLaunchKernelGGL(max_depth_batch_kernel, dim3(std::min((n_px + 255) / 256, 64), n), dim3(256), 0, ctx->stream, (const float *)d_depth,
                       n_px, d_max);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    std::vector<unsigned> bits((size_t)n);
    HIVE_CHECK_HIP(ctx, hipMemcpyAsync(bits.data(), d_max, (size_t)n * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    HIVE_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int f = 0; f < n; ++f) {
        float max_depth;
        memcpy(&max_depth, &bits[f], 4);
        frustum_corners(max_depth, H, W, K, cam_poses + 16 * (size_t)f, out + 15 * (size_t)f);
    }
    return HIVE_OK;
}

int hive_depth_apply_mask(hive_ctx *ctx, const float *d_depth, const uint8_t *d_mask, int n, int H, int W, int iterations, int mode,
                          int instance_id, float *d_out) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, d_depth && d_mask && d_out, "depth_apply_mask: NULL argument");
    HIVE_REQUIRE(ctx, n > 0 && n <= 65535 && H > 0 && W > 0 && iterations >= 0, "depth_apply_mask: bad arguments n=%d %dx%d, %d iterations", n, H, W, iterations);
    HIVE_REQUIRE(ctx, mode == 0 || mode == 1, "depth_apply_mask: mode must be 0 (zero under the dilated mask) or 1 (keep the mask only)");
    HIVE_REQUIRE(ctx, instance_id >= 0 && instance_id <= 255, "depth_apply_mask: instance id %d", instance_id);
    const size_t n_px = (size_t)H * W, total = n_px * (size_t)n;
    if (mode == 1) {
        hipLaunchKernelGGL(keep_mask_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, d_mask, total, instance_id, d_depth, d_out);
    } else {
        int rc = hive_reserve_device(ctx, &ctx->d_scratch, &ctx->scratch_bytes, total);
        if (rc) return rc;
        uint8_t *rows = (uint8_t *)ctx->d_scratch;
        const dim3 grid((unsigned)((n_px + 255) / 256), n);
        hipLaunchKernelGGL(dilate_rows_batch_kernel, grid, dim3(256), 0, ctx->stream, d_mask, H, W, iterations, instance_id, rows);
        hipLaunchKernelGGL(dilate_cols_apply_kernel, grid, dim3(256), 0, ctx->stream, (const uint8_t *)rows, H, W, iterations, d_depth, d_out);
    }
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

// A structuring element that is a full rectangle of odd sides dilates, iterated n times, like ONE (n (kw - 1) + 1) x (n (kh - 1) + 1)
// box maximum (the separable fast path); anything else is iterated literally.
static bool se_is_full_odd_rect(const uint8_t *se, int kh, int kw) {
    if (kh % 2 == 0 || kw % 2 == 0) return false;
    for (int i = 0; i < kh * kw; ++i)
        if (!se[i]) return false;
    return true;
}
static int check_se(hive_ctx *ctx, const uint8_t *se, int kh, int kw, StructuringElement *out) {
    HIVE_REQUIRE(ctx, se && kh >= 1 && kw >= 1 && kh <= 32 && kw <= 32, "dilate: structuring element must be 1x1 .. 32x32, got %dx%d", kh, kw);
    bool any = false;
    out->kh = kh;
    out->kw = kw;
    memset(out->m, 0, sizeof(out->m));
    for (int i = 0; i < kh * kw; ++i) {
        out->m[i] = se[i] ? 1 : 0;
        any = any || se[i];
    }
    HIVE_REQUIRE(ctx, any, "dilate: the structuring element has no set element");
    return HIVE_OK;
}
// `iterations` literal passes of the structuring element over n frames: d_in (instance ids on the first pass) -> result in *d_result
// (one of the two scratch planes a, b; both n * H * W bytes)
static int dilate_se_iterate(hive_ctx *ctx, const uint8_t *d_in, int n, int H, int W, const StructuringElement &se, int iterations, int instance, uint8_t *a,
                             uint8_t *b, const uint8_t **d_result) {
    const dim3 grid((unsigned)(((size_t)H * W + 255) / 256), n);
    const uint8_t *src = d_in;
    uint8_t *dst = a;
    for (int it = 0; it < iterations; ++it) {
        hipLaunchKernelGGL(dilate_se_kernel, grid, dim3(256), 0, ctx->stream, src, H, W, se, instance, it == 0 ? 1 : 0, dst);
        src = dst;
        dst = dst == a ? b : a;
    }
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    *d_result = src;
    return HIVE_OK;
}

int hive_dilate_mask_se(hive_ctx *ctx, const uint8_t *mask, int H, int W, const uint8_t *se, int kh, int kw, int iterations, int mem, uint8_t *out) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, mask && out, "dilate_mask: NULL argument");
    HIVE_REQUIRE(ctx, H > 0 && W > 0 && iterations >= 0, "dilate_mask: bad arguments %dx%d, %d iterations", H, W, iterations);
    StructuringElement el;
    int rc = check_se(ctx, se, kh, kw, &el);
    if (rc) return rc;
    const size_t n = (size_t)H * W;
    const bool host = mem == HIVE_MEM_HOST;
    // planes: [0] the input (host calls), [1], [2] ping-pong
    if ((rc = hive_reserve_device(ctx, &ctx->d_in, &ctx->in_bytes, 3 * align256(n)))) return rc;
    uint8_t *base = (uint8_t *)ctx->d_in;
    const uint8_t *d_mask = mask;
    if (host) {
        const void *dm;
        if ((rc = to_device(ctx, mask, n, 0, mem, &dm))) return rc;
        d_mask = (const uint8_t *)dm;
    }
    uint8_t *a = base + align256(n), *b = base + 2 * align256(n);
    const uint8_t *res;
    if (se_is_full_odd_rect(se, kh, kw)) {
        const dim3 grid((unsigned)((n + 255) / 256));
        hipLaunchKernelGGL(dilate_rows_kernel, grid, dim3(256), 0, ctx->stream, d_mask, H, W, iterations * (kw / 2), a);
        hipLaunchKernelGGL(dilate_cols_kernel, grid, dim3(256), 0, ctx->stream, (const uint8_t *)a, H, W, iterations * (kh / 2), b);
        HIVE_CHECK_HIP(ctx, hipGetLastError());
        res = b;
    } else if (iterations == 0) {  // cv2.dilate(iterations=0) copies; astype(bool) on the way out
        hipLaunchKernelGGL(dilate_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_mask, H, W, 0, a);
        HIVE_CHECK_HIP(ctx, hipGetLastError());
        res = a;
    } else if ((rc = dilate_se_iterate(ctx, d_mask, 1, H, W, el, iterations, 0, a, b, &res))) {
        return rc;
    }
    HIVE_CHECK_HIP(ctx, hipMemcpyAsync(out, res, n, host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, ctx->stream));
    if (host) HIVE_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HIVE_OK;
}

int hive_depth_apply_mask_se(hive_ctx *ctx, const float *d_depth, const uint8_t *d_mask, int n, int H, int W, const uint8_t *se, int kh, int kw, int iterations,
                             int mode, int instance_id, float *d_out) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    StructuringElement el;
    int rc = check_se(ctx, se, kh, kw, &el);
    if (rc) return rc;
    if (mode == 1) return hive_depth_apply_mask(ctx, d_depth, d_mask, n, H, W, iterations, mode, instance_id, d_out);  // the foreground keeps the UNDILATED mask
    HIVE_REQUIRE(ctx, d_depth && d_mask && d_out, "depth_apply_mask: NULL argument");
    HIVE_REQUIRE(ctx, n > 0 && n <= 65535 && H > 0 && W > 0 && iterations >= 0, "depth_apply_mask: bad arguments n=%d %dx%d, %d iterations", n, H, W, iterations);
    HIVE_REQUIRE(ctx, mode == 0, "depth_apply_mask: mode must be 0 (zero under the dilated mask) or 1 (keep the mask only)");
    HIVE_REQUIRE(ctx, instance_id >= 0 && instance_id <= 255, "depth_apply_mask: instance id %d", instance_id);
    if (kh % 2 == 1 && kh == kw && se_is_full_odd_rect(se, kh, kw))  // square box: the separable two-launch path (radius iterations * (k / 2))
        return hive_depth_apply_mask(ctx, d_depth, d_mask, n, H, W, iterations * (kh / 2), mode, instance_id, d_out);
    const size_t total = (size_t)H * W * (size_t)n;
    if ((rc = hive_reserve_device(ctx, &ctx->d_scratch, &ctx->scratch_bytes, 2 * align256(total)))) return rc;
    uint8_t *a = (uint8_t *)ctx->d_scratch, *b = a + align256(total);
    const uint8_t *res;
    // (at least one literal pass, so that instance ids become set / clear; zero iterations = the undilated mask: a 1x1 pass)
    StructuringElement one;
    one.kh = one.kw = 1;
    memset(one.m, 0, sizeof(one.m));
    one.m[0] = 1;
    if ((rc = dilate_se_iterate(ctx, d_mask, n, H, W, iterations ? el : one, iterations ? iterations : 1, instance_id, a, b, &res))) return rc;
    hipLaunchKernelGGL(zero_under_mask_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, res, total, d_depth, d_out);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

int hive_depth_mm_to_m(hive_ctx *ctx, const uint16_t *d_mm, int64_t n, float depth_scale, float max_depth, float *d_out) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, d_mm && d_out && n > 0, "depth_mm_to_m: bad argument");
    hipLaunchKernelGGL(depth_mm_to_m_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_mm, (size_t)n, depth_scale, max_depth, d_out);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

int hive_unproject(hive_ctx *ctx, const float *depth, const uint8_t *mask, const uint8_t *rgb, int H, int W,
                   const double Kinv[9], const double R[9], const double t[3], int mem, double *out_xyz, uint8_t *out_rgba,
                   int64_t capacity, int64_t *n_out) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, depth && Kinv && R && t && n_out, "unproject: NULL argument");
    HIVE_REQUIRE(ctx, H > 0 && W > 0 && (long long)H * W < (1ll << 30), "unproject: bad image size %dx%d", H, W);
    HIVE_REQUIRE(ctx, capacity >= 0 && (capacity == 0 || out_xyz), "unproject: out_xyz is NULL");
    HIVE_REQUIRE(ctx, !out_rgba || rgb, "unproject: out_rgba given without rgb");
    const int n = H * W;
    const int nb = (n + UP_TILE - 1) / UP_TILE;
    int rc;
    const size_t off_mask = align256((size_t)n * 4), off_rgb = off_mask + align256((size_t)n);
    const size_t off_xyz = off_rgb + align256((size_t)n * 3), off_rgba = off_xyz + align256((size_t)capacity * 24);
    const size_t off_blk = off_rgba + align256((size_t)capacity * 4);
    if (mem == HIVE_MEM_HOST) {
        if ((rc = hive_reserve_device(ctx, &ctx->d_in, &ctx->in_bytes, off_blk + (size_t)nb * 4))) return rc;
    } else {
        if ((rc = hive_reserve_device(ctx, &ctx->d_scratch, &ctx->scratch_bytes, (size_t)nb * 4))) return rc;
    }
    const void *d_depth, *d_mask, *d_rgb;
    if ((rc = to_device(ctx, depth, (size_t)n * 4, 0, mem, &d_depth))) return rc;
    if ((rc = to_device(ctx, mask, (size_t)n, off_mask, mem, &d_mask))) return rc;
    if ((rc = to_device(ctx, out_rgba ? rgb : nullptr, (size_t)n * 3, off_rgb, mem, &d_rgb))) return rc;
    unsigned *blk = mem == HIVE_MEM_HOST ? (unsigned *)((char *)ctx->d_in + off_blk) : (unsigned *)ctx->d_scratch;
    double *d_xyz = mem == HIVE_MEM_HOST ? (double *)((char *)ctx->d_in + off_xyz) : out_xyz;
    uint8_t *d_rgba = !out_rgba ? nullptr : (mem == HIVE_MEM_HOST ? (uint8_t *)ctx->d_in + off_rgba : out_rgba);
    unsigned long long *d_total = (unsigned long long *)(ctx->d_scalars + 20);
    hipLaunchKernelGGL(unproject_count_kernel, dim3(nb), dim3(256), 0, ctx->stream, (const float *)d_depth, (const uint8_t *)d_mask, n, blk);
    hipLaunchKernelGGL(scan1_kernel, dim3(1), dim3(1024), 0, ctx->stream, blk, nb, d_total);
    UnprojectParams p;
    memcpy(p.Kinv, Kinv, sizeof(p.Kinv));
    memcpy(p.R, R, sizeof(p.R));
    memcpy(p.t, t, sizeof(p.t));
    p.H = H;
    p.W = W;
    p.capacity = capacity;
    hipLaunchKernelGGL(unproject_write_kernel, dim3(nb), dim3(256), 0, ctx->stream, (const float *)d_depth, (const uint8_t *)d_mask,
                       (const uint8_t *)d_rgb, p, blk, d_xyz, d_rgba);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    unsigned long long total = 0;
    HIVE_CHECK_HIP(ctx, hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIVE_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = (int64_t)total;
    if (mem == HIVE_MEM_HOST) {
        const size_t m = (size_t)std::min<long long>((long long)total, capacity);
        if (m) HIVE_CHECK_HIP(ctx, hipMemcpy(out_xyz, d_xyz, m * 24, hipMemcpyDeviceToHost));
        if (m && out_rgba) HIVE_CHECK_HIP(ctx, hipMemcpy(out_rgba, d_rgba, m * 4, hipMemcpyDeviceToHost));
    }
    if ((long long)total > capacity && capacity > 0)
        return hive_fail(ctx, HIVE_ERR_INVALID, "unproject: %llu points do not fit capacity %lld", total, (long long)capacity);
    return HIVE_OK;
}

int hive_image2world(hive_ctx *ctx, const double *uv, const double *depth, int64_t n, const double Kinv[9], const double R[9],
                     const double t[3], double scale_factor, int mem, double *out_xyz) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, Kinv && R && t, "image2world: NULL argument");
    HIVE_REQUIRE(ctx, n >= 0 && (n == 0 || (uv && depth && out_xyz)), "image2world: bad arguments");
    if (n == 0) return HIVE_OK;
    int rc;
    const size_t off_depth = align256((size_t)n * 16), off_out = off_depth + align256((size_t)n * 8);
    const void *d_uv, *d_depth;
    double *d_out = out_xyz;
    if (mem == HIVE_MEM_HOST) {
        if ((rc = hive_reserve_device(ctx, &ctx->d_in, &ctx->in_bytes, off_out + (size_t)n * 24))) return rc;
        d_out = (double *)((char *)ctx->d_in + off_out);
    }
    if ((rc = to_device(ctx, uv, (size_t)n * 16, 0, mem, &d_uv))) return rc;
    if ((rc = to_device(ctx, depth, (size_t)n * 8, off_depth, mem, &d_depth))) return rc;
    UnprojectParams p;
    memcpy(p.Kinv, Kinv, sizeof(p.Kinv));
    memcpy(p.R, R, sizeof(p.R));
    memcpy(p.t, t, sizeof(p.t));
    p.H = p.W = 0;
    p.capacity = n;
    hipLaunchKernelGGL(image2world_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const double *)d_uv,
                       (const double *)d_depth, (long long)n, p, scale_factor, d_out);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    if (mem == HIVE_MEM_HOST) {
        HIVE_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        HIVE_CHECK_HIP(ctx, hipMemcpy(out_xyz, d_out, (size_t)n * 24, hipMemcpyDeviceToHost));
    }
    return HIVE_OK;
}

int hive_project(hive_ctx *ctx, const double *points, int64_t n, const double K[9], const double R[9], const double t[3],
                 double scale_factor, int mem, int32_t *out_uv_i32, double *out_uv_f64, double *out_depth) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, K && R && t, "project: NULL argument");
    HIVE_REQUIRE(ctx, n >= 0 && (n == 0 || points), "project: bad points");
    HIVE_REQUIRE(ctx, (out_uv_i32 != nullptr) != (out_uv_f64 != nullptr), "project: exactly one of out_uv_i32 / out_uv_f64 must be given");
    if (n == 0) return HIVE_OK;
    int rc;
    const size_t off_uv = align256((size_t)n * 24), off_depth = off_uv + align256((size_t)n * 16);
    const void *d_pts;
    int32_t *d_uvi = out_uv_i32;
    double *d_uvf = out_uv_f64, *d_depth = out_depth;
    if (mem == HIVE_MEM_HOST) {
        if ((rc = hive_reserve_device(ctx, &ctx->d_in, &ctx->in_bytes, off_depth + (size_t)n * 8))) return rc;
        if (out_uv_i32) d_uvi = (int32_t *)((char *)ctx->d_in + off_uv);
        if (out_uv_f64) d_uvf = (double *)((char *)ctx->d_in + off_uv);
        if (out_depth) d_depth = (double *)((char *)ctx->d_in + off_depth);
    }
    if ((rc = to_device(ctx, points, (size_t)n * 24, 0, mem, &d_pts))) return rc;
    ProjectParams p;
    memcpy(p.K, K, sizeof(p.K));
    memcpy(p.R, R, sizeof(p.R));
    memcpy(p.t, t, sizeof(p.t));
    p.scale = scale_factor;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (ctx->round_mode)
        hipLaunchKernelGGL(project_kernel<1>, grid, dim3(256), 0, ctx->stream, (const double *)d_pts, (long long)n, p, d_uvi, d_uvf, d_depth);
    else
        hipLaunchKernelGGL(project_kernel<0>, grid, dim3(256), 0, ctx->stream, (const double *)d_pts, (long long)n, p, d_uvi, d_uvf, d_depth);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    if (mem == HIVE_MEM_HOST) {
        HIVE_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (out_uv_i32) HIVE_CHECK_HIP(ctx, hipMemcpy(out_uv_i32, d_uvi, (size_t)n * 8, hipMemcpyDeviceToHost));
        if (out_uv_f64) HIVE_CHECK_HIP(ctx, hipMemcpy(out_uv_f64, d_uvf, (size_t)n * 16, hipMemcpyDeviceToHost));
        if (out_depth) HIVE_CHECK_HIP(ctx, hipMemcpy(out_depth, d_depth, (size_t)n * 8, hipMemcpyDeviceToHost));
    }
    return HIVE_OK;
}

int hive_project_bbox(hive_ctx *ctx, const double *points, int64_t n, const double K[9], const double R[9], const double t[3],
                      int W, int H, int mem, int32_t out[5]) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, K && R && t && out && W > 0 && H > 0, "project_bbox: bad arguments");
    HIVE_REQUIRE(ctx, n >= 0 && (n == 0 || points), "project_bbox: bad points");
    int32_t init[5] = {0x7fffffff, (int32_t)0x80000000, 0x7fffffff, (int32_t)0x80000000, 0};
    memcpy(out, init, sizeof(init));
    if (n == 0) return HIVE_OK;
    int rc;
    const void *d_pts;
    if (mem == HIVE_MEM_HOST && (rc = hive_reserve_device(ctx, &ctx->d_in, &ctx->in_bytes, (size_t)n * 24))) return rc;
    if ((rc = to_device(ctx, points, (size_t)n * 24, 0, mem, &d_pts))) return rc;
    int *d_out = (int *)(ctx->d_scalars + 24);
    HIVE_CHECK_HIP(ctx, hipMemcpyAsync(d_out, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    ProjectParams p;
    memcpy(p.K, K, sizeof(p.K));
    memcpy(p.R, R, sizeof(p.R));
    memcpy(p.t, t, sizeof(p.t));
    p.scale = 1.0;
    const dim3 grid((unsigned)std::min<long long>((n + 255) / 256, (long long)ctx->num_cus * 4));
    hipLaunchKernelGGL(project_bbox_kernel, grid, dim3(256), 0, ctx->stream, (const double *)d_pts, (long long)n, p, W, H, d_out);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    HIVE_CHECK_HIP(ctx, hipMemcpyAsync(out, d_out, sizeof(init), hipMemcpyDeviceToHost, ctx->stream));
    HIVE_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HIVE_OK;
}

int hive_dilate_mask(hive_ctx *ctx, const uint8_t *mask, int H, int W, int iterations, int mem, uint8_t *out) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, mask && out, "dilate_mask: NULL argument");
    HIVE_REQUIRE(ctx, H > 0 && W > 0 && iterations >= 0, "dilate_mask: bad arguments %dx%d, %d iterations", H, W, iterations);
    const size_t n = (size_t)H * W;
    int rc;
    const size_t off_tmp = align256(n), off_out = 2 * align256(n);
    const void *d_mask;
    uint8_t *d_tmp, *d_out;
    if (mem == HIVE_MEM_HOST) {
        if ((rc = hive_reserve_device(ctx, &ctx->d_in, &ctx->in_bytes, 3 * align256(n)))) return rc;
        if ((rc = to_device(ctx, mask, n, 0, mem, &d_mask))) return rc;
        d_tmp = (uint8_t *)ctx->d_in + off_tmp;
        d_out = (uint8_t *)ctx->d_in + off_out;
    } else {
        if ((rc = hive_reserve_device(ctx, &ctx->d_scratch, &ctx->scratch_bytes, n))) return rc;
        d_mask = mask;
        d_tmp = (uint8_t *)ctx->d_scratch;
        d_out = out;
    }
    const dim3 grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(dilate_rows_kernel, grid, dim3(256), 0, ctx->stream, (const uint8_t *)d_mask, H, W, iterations, d_tmp);
    hipLaunchKernelGGL(dilate_cols_kernel, grid, dim3(256), 0, ctx->stream, (const uint8_t *)d_tmp, H, W, iterations, d_out);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    if (mem == HIVE_MEM_HOST) {
        HIVE_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        HIVE_CHECK_HIP(ctx, hipMemcpy(out, d_out, n, hipMemcpyDeviceToHost));
    }
    return HIVE_OK;
}

int hive_depth_quantize(hive_ctx *ctx, const void *d_depth, int dtype, int H, int W, float depth_scale, float max_depth,
                        const uint8_t *d_mask, uint16_t *d_out_mm, float *d_out_m) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, d_depth && (d_out_mm || d_out_m), "depth_quantize: NULL argument");
    HIVE_REQUIRE(ctx, H > 0 && W > 0, "depth_quantize: bad image size");
    const int n = H * W;
    const dim3 grid((unsigned)((n + 255) / 256));
    switch (dtype) {
        case HIVE_F32:
            hipLaunchKernelGGL(depth_quantize_kernel<float>, grid, dim3(256), 0, ctx->stream, d_depth, n, depth_scale, max_depth, d_mask, d_out_mm, d_out_m);
            break;
        case HIVE_F16:
            hipLaunchKernelGGL(depth_quantize_kernel<_Float16>, grid, dim3(256), 0, ctx->stream, d_depth, n, depth_scale, max_depth, d_mask, d_out_mm, d_out_m);
            break;
        case HIVE_BF16:
            hipLaunchKernelGGL(depth_quantize_kernel<unsigned short>, grid, dim3(256), 0, ctx->stream, d_depth, n, depth_scale, max_depth, d_mask, d_out_mm, d_out_m);
            break;
        default:
            return hive_fail(ctx, HIVE_ERR_INVALID, "depth_quantize: unknown dtype %d", dtype);
    }
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

}  // extern "C"
