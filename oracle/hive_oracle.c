/*
 * hive_oracle.c -- CPU restatement (plain C) of HIVE's depth->TSDF hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under hive_amd/ may import, link or call this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY STATUS
 *   - oracle_unproject / oracle_project restate hive/geometric.py (present in the reference)
 *     and are PINNED by golden vectors generated from the real module
 *     (tests/golden/make_golden.py -> tests/golden/geometric_*.npz).
 *   - oracle_tsdf_integrate / oracle_view_frustum / oracle_marching_cubes restate
 *     third_party/tsdf_fusion_python (AnthonyDickson/tsdf-fusion-python, a fork of
 *     andyzeng/tsdf-fusion-python; submodule EMPTY in /root/reference, commit pin lost,
 *     runtime pins numba==0.55.0, pycuda==2021.1, scikit-image==0.19.1 in requirements.txt:13-16).
 *     Their published algorithm is restated from the call sites hive/fusion.py:59,104,124,127;
 *     the reference holds no test, fixture or golden vector for them:  **parity unpinned**.
 *     Known-answer tests (analytic plane / sphere) stand in for golden vectors.
 *
 * Arithmetic contract for integrate (the spec of record):
 *   single precision, the operation order of the reference library's `integrate` CUDA kernel,
 *   no fused multiply-add (build with -ffp-contract=off), IEEE correctly rounded + - * / sqrt.
 *   Rounding of pixel coordinates and colours is selectable: 0 = half-to-even (np.round, the
 *   reference library's CPU path), 1 = half-away-from-zero (roundf, its CUDA kernel).
 *   Frustum test is the CPU path's (cam_z > 0).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/hive_mc_tables.h"

static inline float round_mode_f(float x, int mode) { return mode ? roundf(x) : rintf(x); }

/* threads of the integrate loops (the only parallel region; every other function is serial): 0 = the OpenMP default */
#ifdef _OPENMP
#include <omp.h>
int oracle_set_threads(int n) {
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
}
#else
int oracle_set_threads(int n) { (void)n; return 1; }
#endif

/* ------------------------------------------------------------------------------------ */
/* fusion.TSDFVolume.__init__ (call site hive/fusion.py:104): vol_dim = ceil((max-min)/voxel) */
void oracle_tsdf_dims(const double vol_bnds[6], double voxel_size, int64_t vol_dim[3]) {
    for (int a = 0; a < 3; ++a)
        vol_dim[a] = (int64_t)ceil((vol_bnds[2 * a + 1] - vol_bnds[2 * a]) / voxel_size);
}

/* fusion.TSDFVolume.integrate (call site hive/fusion.py:124).
 * tsdf/weight/color: float32 [X][Y][Z]; color_im u8 [H][W][3] RGB; depth f32 [H][W];
 * K f32 3x3; cam_pose f64 4x4 camera-to-world.  Returns the number of voxels written. */
uint64_t oracle_tsdf_integrate(float *tsdf, float *weight, float *color, const int64_t vol_dim[3],
                               const float origin[3], float voxel_size, float trunc_margin,
                               const uint8_t *color_im, const float *depth_im, int H, int W,
                               const float K[9], const double cam_pose[16], float obs_weight,
                               int round_mode) {
    float P[16];
    for (int i = 0; i < 16; ++i) P[i] = (float)cam_pose[i];
    const float fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const int64_t X = vol_dim[0], Y = vol_dim[1], Z = vol_dim[2];
    uint64_t n_upd = 0;
    /* voxels are independent: the x planes run on the host's cores (OMP_NUM_THREADS / oracle_set_threads); same bits as one thread */
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : n_upd)
    for (int64_t x = 0; x < X; ++x)
        for (int64_t y = 0; y < Y; ++y)
            for (int64_t z = 0; z < Z; ++z) {
                const int64_t idx = (x * Y + y) * Z + z;
                /* voxel grid -> world */
                const float pt_x = origin[0] + (float)x * voxel_size;
                const float pt_y = origin[1] + (float)y * voxel_size;
                const float pt_z = origin[2] + (float)z * voxel_size;
                /* world -> camera: R^T (p - t) */
                const float tx = pt_x - P[3], ty = pt_y - P[7], tz = pt_z - P[11];
                const float cam_x = P[0] * tx + P[4] * ty + P[8] * tz;
                const float cam_y = P[1] * tx + P[5] * ty + P[9] * tz;
                const float cam_z = P[2] * tx + P[6] * ty + P[10] * tz;
                if (!(cam_z > 0.0f)) continue;
                /* camera -> pixel */
                const float px = round_mode_f(fx * (cam_x / cam_z) + cx, round_mode);
                const float py = round_mode_f(fy * (cam_y / cam_z) + cy, round_mode);
                if (!(px >= 0.0f && px < (float)W && py >= 0.0f && py < (float)H)) continue;
                const int pix = (int)py * W + (int)px;
                const float depth_value = depth_im[pix];
                if (depth_value == 0.0f) continue;
                const float depth_diff = depth_value - cam_z;
                if (depth_diff < -trunc_margin) continue;
                const float dist = fminf(1.0f, depth_diff / trunc_margin);
                const float w_old = weight[idx];
                const float w_new = w_old + obs_weight;
                weight[idx] = w_new;
                tsdf[idx] = (tsdf[idx] * w_old + obs_weight * dist) / w_new;
                /* colour: running average per channel, rounded and clamped every frame */
                const float old_color = color[idx];
                const float old_b = floorf(old_color / 65536.0f);
                const float old_g = floorf((old_color - old_b * 65536.0f) / 256.0f);
                const float old_r = old_color - old_b * 65536.0f - old_g * 256.0f;
                const float new_r = (float)color_im[3 * pix + 0];
                const float new_g = (float)color_im[3 * pix + 1];
                const float new_b = (float)color_im[3 * pix + 2];
                const float b = fminf(round_mode_f((old_b * w_old + obs_weight * new_b) / w_new, round_mode), 255.0f);
                const float g = fminf(round_mode_f((old_g * w_old + obs_weight * new_g) / w_new, round_mode), 255.0f);
                const float r = fminf(round_mode_f((old_r * w_old + obs_weight * new_r) / w_new, round_mode), 255.0f);
                color[idx] = b * 65536.0f + g * 256.0f + r;
                ++n_upd;
            }
    return n_upd;
}

/* Frame-sharded accumulation (new design): accum planes [5][N] =
 * num = sum(w*dist), w = sum(w), r,g,b = sum(w*c).  Same inclusion tests as integrate. */
uint64_t oracle_tsdf_accum_integrate(float *accum, const int64_t vol_dim[3], const float origin[3],
                                     float voxel_size, float trunc_margin, const uint8_t *color_im,
                                     const float *depth_im, int H, int W, const float K[9],
                                     const double cam_pose[16], float obs_weight, int round_mode) {
    float P[16];
    for (int i = 0; i < 16; ++i) P[i] = (float)cam_pose[i];
    const float fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const int64_t X = vol_dim[0], Y = vol_dim[1], Z = vol_dim[2], N = X * Y * Z;
    uint64_t n_upd = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : n_upd)
    for (int64_t x = 0; x < X; ++x)
        for (int64_t y = 0; y < Y; ++y)
            for (int64_t z = 0; z < Z; ++z) {
                const int64_t idx = (x * Y + y) * Z + z;
                const float pt_x = origin[0] + (float)x * voxel_size;
                const float pt_y = origin[1] + (float)y * voxel_size;
                const float pt_z = origin[2] + (float)z * voxel_size;
                const float tx = pt_x - P[3], ty = pt_y - P[7], tz = pt_z - P[11];
                const float cam_x = P[0] * tx + P[4] * ty + P[8] * tz;
                const float cam_y = P[1] * tx + P[5] * ty + P[9] * tz;
                const float cam_z = P[2] * tx + P[6] * ty + P[10] * tz;
                if (!(cam_z > 0.0f)) continue;
                const float px = round_mode_f(fx * (cam_x / cam_z) + cx, round_mode);
                const float py = round_mode_f(fy * (cam_y / cam_z) + cy, round_mode);
                if (!(px >= 0.0f && px < (float)W && py >= 0.0f && py < (float)H)) continue;
                const int pix = (int)py * W + (int)px;
                const float depth_value = depth_im[pix];
                if (depth_value == 0.0f) continue;
                const float depth_diff = depth_value - cam_z;
                if (depth_diff < -trunc_margin) continue;
                const float dist = fminf(1.0f, depth_diff / trunc_margin);
                accum[0 * N + idx] = accum[0 * N + idx] + obs_weight * dist;
                accum[1 * N + idx] = accum[1 * N + idx] + obs_weight;
                accum[2 * N + idx] = accum[2 * N + idx] + obs_weight * (float)color_im[3 * pix + 0];
                accum[3 * N + idx] = accum[3 * N + idx] + obs_weight * (float)color_im[3 * pix + 1];
                accum[4 * N + idx] = accum[4 * N + idx] + obs_weight * (float)color_im[3 * pix + 2];
                ++n_upd;
            }
    return n_upd;
}

/* fold accumulators into the (tsdf, weight, colour) volumes: tsdf = num/w (1 where w == 0) */
void oracle_tsdf_accum_finalize(const float *accum, int64_t N, float *tsdf, float *weight, float *color,
                                int round_mode) {
    for (int64_t i = 0; i < N; ++i) {
        const float w = accum[1 * N + i];
        if (w > 0.0f) {
            tsdf[i] = accum[0 * N + i] / w;
            const float r = fminf(round_mode_f(accum[2 * N + i] / w, round_mode), 255.0f);
            const float g = fminf(round_mode_f(accum[3 * N + i] / w, round_mode), 255.0f);
            const float b = fminf(round_mode_f(accum[4 * N + i] / w, round_mode), 255.0f);
            color[i] = b * 65536.0f + g * 256.0f + r;
        } else {
            tsdf[i] = 1.0f;
            color[i] = 0.0f;
        }
        weight[i] = w;
    }
}

/* fusion.get_view_frustum (call site hive/fusion.py:59): apex + the four image corners pushed to
 * max(depth), camera -> world.  The reference library evaluates this in float64 on float32 inputs.
 * out row-major [3][5]. */
void oracle_view_frustum(const float *depth_im, int H, int W, const float K[9], const double cam_pose[16],
                         double out[15]) {
    float max_depth = depth_im[0];
    for (int64_t i = 1; i < (int64_t)H * W; ++i)
        if (depth_im[i] > max_depth) max_depth = depth_im[i];
    const double md = (double)max_depth;
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const double col[5] = {0, 0, 0, (double)W, (double)W};
    const double row[5] = {0, 0, (double)H, 0, (double)H};
    const double dep[5] = {0, md, md, md, md};
    for (int c = 0; c < 5; ++c) {
        const double p[3] = {(col[c] - cx) * dep[c] / fx, (row[c] - cy) * dep[c] / fy, dep[c]};
        for (int r = 0; r < 3; ++r)
            out[r * 5 + c] = cam_pose[4 * r + 0] * p[0] + cam_pose[4 * r + 1] * p[1] +
                             cam_pose[4 * r + 2] * p[2] + cam_pose[4 * r + 3];
    }
}

/* hive/geometric.py:107-126 point_cloud_from_depth + :183-206 image2world (+ :129-152 rgbd).
 * valid = mask & (depth > 0) in row-major (v,u) order (np.nonzero order);
 * X = R^T (d * Kinv [u,v,1]^T - t), float64.  Kinv is np.linalg.inv(K) *in K's dtype*
 * (geometric.py:203), computed by the caller and widened to float64.
 * mask may be NULL (all true); rgb / out_rgba may be NULL.  Returns the number of points. */
int64_t oracle_unproject(const float *depth, const uint8_t *mask, const uint8_t *rgb, int H, int W,
                         const double Kinv[9], const double R[9], const double t[3], double *out_xyz,
                         uint8_t *out_rgba) {
    int64_t n = 0;
    for (int v = 0; v < H; ++v)
        for (int u = 0; u < W; ++u) {
            const int64_t i = (int64_t)v * W + u;
            if ((mask && !mask[i]) || !(depth[i] > 0.0f)) continue;
            const double d = (double)depth[i];
            const double pu = (double)u, pv = (double)v;
            double c[3];
            for (int r = 0; r < 3; ++r)
                c[r] = d * (Kinv[3 * r + 0] * pu + Kinv[3 * r + 1] * pv + Kinv[3 * r + 2]) - t[r];
            for (int r = 0; r < 3; ++r) /* R^T */
                out_xyz[3 * n + r] = R[0 * 3 + r] * c[0] + R[1 * 3 + r] * c[1] + R[2 * 3 + r] * c[2];
            if (rgb && out_rgba) {
                out_rgba[4 * n + 0] = rgb[3 * i + 0];
                out_rgba[4 * n + 1] = rgb[3 * i + 1];
                out_rgba[4 * n + 2] = rgb[3 * i + 2];
                out_rgba[4 * n + 3] = 255;
            }
            ++n;
        }
    return n;
}

/* hive/geometric.py:155-180 world2image: c = K (R X + t); depth = c_z; uv = c_xy / depth / scale;
 * integer dtype: np.round (half-even) then cast (:175-178).  All float64. */
void oracle_project(const double *points, int64_t n, const double K[9], const double R[9], const double t[3],
                    double scale_factor, int round_mode, int32_t *out_uv_i32, double *out_uv_f64,
                    double *out_depth) {
    for (int64_t i = 0; i < n; ++i) {
        const double *X = points + 3 * i;
        double cam[3], c[3];
        for (int r = 0; r < 3; ++r) cam[r] = R[3 * r + 0] * X[0] + R[3 * r + 1] * X[1] + R[3 * r + 2] * X[2] + t[r];
        for (int r = 0; r < 3; ++r) c[r] = K[3 * r + 0] * cam[0] + K[3 * r + 1] * cam[1] + K[3 * r + 2] * cam[2];
        const double u = c[0] / c[2] / scale_factor;
        const double v = c[1] / c[2] / scale_factor;
        if (out_depth) out_depth[i] = c[2];
        if (out_uv_i32) {
            out_uv_i32[2 * i + 0] = (int32_t)(round_mode ? round(u) : rint(u));
            out_uv_i32[2 * i + 1] = (int32_t)(round_mode ? round(v) : rint(v));
        }
        if (out_uv_f64) {
            out_uv_f64[2 * i + 0] = u;
            out_uv_f64[2 * i + 1] = v;
        }
    }
}

/* hive/image_processing.py:30-45 dilate_mask with ANY structuring element (hive/options.py:245-268 `dilation_filter`), as cv2.dilate
 * defines it: dst(v, u) = max over the set taps (j, i) of src(v + j - kh / 2, u + i - kw / 2), anchor at the element's centre
 * (integer division), taps outside the image ignored, the pass repeated `iterations` times. */
void oracle_dilate_mask_se(const uint8_t *mask, int H, int W, const uint8_t *se, int kh, int kw, int iterations, uint8_t *out) {
    uint8_t *a = (uint8_t *)malloc((size_t)H * W), *b = (uint8_t *)malloc((size_t)H * W);
    for (int64_t i = 0; i < (int64_t)H * W; ++i) a[i] = mask[i] ? 1 : 0;
    for (int it = 0; it < iterations; ++it) {
        for (int v = 0; v < H; ++v)
            for (int u = 0; u < W; ++u) {
                uint8_t m = 0;
                for (int j = 0; j < kh; ++j)
                    for (int i = 0; i < kw; ++i) {
                        const int vv = v + j - kh / 2, uu = u + i - kw / 2;
                        if (se[j * kw + i] && vv >= 0 && vv < H && uu >= 0 && uu < W && a[(int64_t)vv * W + uu]) m = 1;
                    }
                b[(int64_t)v * W + u] = m;
            }
        uint8_t *s = a;
        a = b;
        b = s;
    }
    memcpy(out, a, (size_t)H * W);
    free(a);
    free(b);
}

/* hive/image_processing.py:30-45 dilate_mask with the default 3x3 rectangle (hive/options.py:248),
 * applied `iterations` times, literally (one 3x3 max per iteration, outside pixels ignored). */
void oracle_dilate_mask(const uint8_t *mask, int H, int W, int iterations, uint8_t *out) {
    uint8_t *a = (uint8_t *)malloc((size_t)H * W), *b = (uint8_t *)malloc((size_t)H * W);
    for (int64_t i = 0; i < (int64_t)H * W; ++i) a[i] = mask[i] ? 1 : 0;
    for (int it = 0; it < iterations; ++it) {
        for (int v = 0; v < H; ++v)
            for (int u = 0; u < W; ++u) {
                uint8_t m = 0;
                for (int dv = -1; dv <= 1; ++dv)
                    for (int du = -1; du <= 1; ++du) {
                        const int vv = v + dv, uu = u + du;
                        if (vv >= 0 && vv < H && uu >= 0 && uu < W && a[(int64_t)vv * W + uu]) m = 1;
                    }
                b[(int64_t)v * W + u] = m;
            }
        uint8_t *s = a;
        a = b;
        b = s;
    }
    memcpy(out, a, (size_t)H * W);
    free(a);
    free(b);
}

/* hive/dataset_adaptors.py:1432-1433 (x1000, astype(uint16) truncation) followed by the loader's
 * hive/io.py:1032-1039 (x depth_scale as float32, > max_depth -> 0) and fusion.py:121 (mask -> 0). */
void oracle_depth_quantize(const float *depth_m, int64_t n, float depth_scale, float max_depth,
                           const uint8_t *mask, uint16_t *out_mm, float *out_m) {
    for (int64_t i = 0; i < n; ++i) {
        const float mm_f = depth_m[i] * 1000.0f;
        const uint16_t mm = (uint16_t)(int32_t)mm_f; /* DPT depth <= 7.257 m: no overflow */
        float m = depth_scale * (float)mm;
        if (m > max_depth) m = 0.0f;
        if (mask && mask[i]) m = 0.0f;
        if (out_mm) out_mm[i] = mm;
        if (out_m) out_m[i] = m;
    }
}

/* ------------------------------------------------------------------------------------ */
/* fusion.TSDFVolume.get_mesh (call site hive/fusion.py:127): marching cubes at level 0, vertices
 * to world coordinates, colours looked up at round(vertex) -- see tools/gen_mc_tables.py for the
 * table conventions.  One vertex per sign-changing grid edge, ordered by (voxel linear index, axis);
 * faces ordered by cell linear index then table order.
 * Two-call protocol: call with outputs NULL to get the counts, then with buffers. */
static inline float grad_axis(const float *v, int64_t x, int64_t y, int64_t z, const int64_t d[3], int axis) {
    int64_t p[3] = {x, y, z}, lo[3] = {x, y, z}, hi[3] = {x, y, z};
    if (p[axis] > 0) lo[axis] -= 1;
    if (p[axis] < d[axis] - 1) hi[axis] += 1;
    const float a = v[(hi[0] * d[1] + hi[1]) * d[2] + hi[2]];
    const float b = v[(lo[0] * d[1] + lo[1]) * d[2] + lo[2]];
    const float span = (float)(hi[axis] - lo[axis]);
    return span > 0.0f ? (a - b) / span : 0.0f;
}

int oracle_marching_cubes(const float *tsdf, const float *color, const int64_t vol_dim[3], const float origin[3],
                          float voxel_size, int64_t *n_verts, int64_t *n_faces, float *verts, int32_t *faces,
                          float *norms, uint8_t *colors, float *verts_voxel) {
    const int64_t X = vol_dim[0], Y = vol_dim[1], Z = vol_dim[2], N = X * Y * Z;
    const int64_t stride[3] = {Y * Z, Z, 1};
    int32_t *vid = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)N);
    if (!vid) return -3;
    int64_t nv = 0;
    for (int64_t x = 0; x < X; ++x)
        for (int64_t y = 0; y < Y; ++y)
            for (int64_t z = 0; z < Z; ++z) {
                const int64_t idx = (x * Y + y) * Z + z;
                const int64_t p[3] = {x, y, z};
                const float v0 = tsdf[idx];
                for (int a = 0; a < 3; ++a) {
                    vid[3 * idx + a] = -1;
                    if (p[a] + 1 >= vol_dim[a]) continue;
                    const float v1 = tsdf[idx + stride[a]];
                    if ((v0 < 0.0f) == (v1 < 0.0f)) continue;
                    vid[3 * idx + a] = (int32_t)nv;
                    if (verts) {
                        const float t = v0 / (v0 - v1);
                        float pos[3] = {(float)x, (float)y, (float)z};
                        pos[a] = pos[a] + t;
                        if (verts_voxel)
                            for (int r = 0; r < 3; ++r) verts_voxel[3 * nv + r] = pos[r];
                        for (int r = 0; r < 3; ++r) verts[3 * nv + r] = pos[r] * voxel_size + origin[r];
                        int64_t q[3] = {x, y, z};
                        q[a] += 1;
                        float g[3];
                        for (int r = 0; r < 3; ++r) {
                            const float g0 = grad_axis(tsdf, x, y, z, vol_dim, r);
                            const float g1 = grad_axis(tsdf, q[0], q[1], q[2], vol_dim, r);
                            g[r] = g0 + t * (g1 - g0);
                        }
                        const float len = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
                        for (int r = 0; r < 3; ++r) norms[3 * nv + r] = len > 0.0f ? g[r] / len : 0.0f;
                        /* colour at round-half-even(vertex) (np.round in the reference library) */
                        int64_t ci[3];
                        for (int r = 0; r < 3; ++r) {
                            ci[r] = (int64_t)rintf(pos[r]);
                            if (ci[r] > vol_dim[r] - 1) ci[r] = vol_dim[r] - 1;
                        }
                        const float c = color[(ci[0] * Y + ci[1]) * Z + ci[2]];
                        const float cb = floorf(c / 65536.0f);
                        const float cg = floorf((c - cb * 65536.0f) / 256.0f);
                        const float cr = c - cb * 65536.0f - cg * 256.0f;
                        colors[3 * nv + 0] = (uint8_t)cr;
                        colors[3 * nv + 1] = (uint8_t)cg;
                        colors[3 * nv + 2] = (uint8_t)cb;
                    }
                    ++nv;
                }
            }
    int64_t nf = 0;
    for (int64_t x = 0; x + 1 < X; ++x)
        for (int64_t y = 0; y + 1 < Y; ++y)
            for (int64_t z = 0; z + 1 < Z; ++z) {
                int cs = 0;
                for (int c = 0; c < 8; ++c) {
                    const int64_t i = ((x + HIVE_MC_CORNER_OFFSET[c][0]) * Y + (y + HIVE_MC_CORNER_OFFSET[c][1])) * Z +
                                      (z + HIVE_MC_CORNER_OFFSET[c][2]);
                    if (tsdf[i] < 0.0f) cs |= 1 << c;
                }
                const int nt = HIVE_MC_NUM_TRIS[cs];
                for (int k = 0; k < nt; ++k) {
                    if (faces)
                        for (int j = 0; j < 3; ++j) {
                            const int e = HIVE_MC_TRI_TABLE[cs][3 * k + j];
                            const int64_t oi = ((x + HIVE_MC_EDGE_OWNER[e][0]) * Y + (y + HIVE_MC_EDGE_OWNER[e][1])) * Z +
                                               (z + HIVE_MC_EDGE_OWNER[e][2]);
                            faces[3 * nf + j] = vid[3 * oi + HIVE_MC_EDGE_OWNER[e][3]];
                        }
                    ++nf;
                }
            }
    free(vid);
    *n_verts = nv;
    *n_faces = nf;
    return 0;
}
