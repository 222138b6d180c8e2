// Foreground per-frame meshing: the inner loops of Pipeline._create_scene (/root/reference/hive/pipeline.py:340-483) that sit
// behind point_cloud_from_depth (csrc/geometry.hip), for gfx950.
//
//   hive_grid_mesh     _triangulate_faces (:651-667) + _filter_faces (:670-694).  The reference runs a Delaunay triangulation
//                      (Qhull) over the valid pixels' integer (u, v) coordinates and then drops every face that has an edge
//                      longer than max_pixel_distance pixels or spanning more than max_depth_distance metres.  The points
//                      ARE a pixel lattice, so the triangulation is implicit: every 2 x 2 pixel block whose corners are valid
//                      gives the two halves of its unit square (fixed diagonal top-right / bottom-left), a block with exactly
//                      three valid corners gives their triangle; no point set, no Qhull.  Faces index the rows of
//                      point_cloud_from_depth's output (valid pixels in row-major order) and are wound like the reference's
//                      reversed simplices (negative cross product in (u, v)).  Delaunay of a lattice also BRIDGES an invalid pixel whose
//                      4-neighbours are valid: the triangle through three of them has the hole as its circumcentre (radius 1, no valid
//                      point inside) and sides (sqrt 2, sqrt 2, 2), which pass the default 2-pixel limit; with all four neighbours valid
//                      the diamond is split along one of its diagonals (they are co-circular: Qhull's arbitrary choice, here west - east).
//                      These are the only triangles of a lattice Delaunay whose sides are all <= 2: with them the face SET after the
//                      filter equals the reference's up to the diagonal inside each co-circular unit square / diamond
//                      (tests/test_fgmesh_gpu.py against scipy's Delaunay).
//   hive_texture_window _get_mesh_texture_and_uv (:782-808): project the vertices (world2image, its default int32 pixels), their
//                      bounding box (the crop of the frame that becomes the texture), uv relative to the box's corner.
//
// All integer work (vertex ids, face lists, the crop window) is exact; the filter compares in the reference's types
// (float32 depth differences, float64 pixel distances).
#include "hive_internal.hpp"

#include <algorithm>

namespace {

constexpr int TILE = 1024;  // pixels (and 2 x 2 blocks, by their top-left pixel) per workgroup

__device__ __forceinline__ bool px_valid(const float *depth, const uint8_t *mask, int i) { return (!mask || mask[i]) && depth[i] > 0.0f; }

struct GridParams {
    int H, W;
    double max_px;    // max_pixel_distance, compared in float64 (points2d is an integer array: np.linalg.norm gives float64)
    float max_depth;  // max_depth_distance, compared in float32 (depth is a float32 array)
};

__device__ __forceinline__ bool edge_ok(const GridParams &p, const float *depth, int a, int b, int du, int dv) {
    const double dist = sqrt((double)(du * du + dv * dv));
    return dist <= p.max_px && fabsf(depth[a] - depth[b]) <= p.max_depth;
}

// triangles that belong to pixel i = (v, u), in order: those of the 2 x 2 block whose top-left pixel it is (corners a b / c d), then, where
// pixel i itself is invalid, the triangles that bridge it (4-neighbours n / w e / s).  Returns the count and the corners (pixel indices).
constexpr int MAX_PIXEL_FACES = 4;
__device__ __forceinline__ int pixel_faces(const GridParams &p, const float *depth, const uint8_t *mask, int i, int (&tri)[MAX_PIXEL_FACES][3]) {
    const int v = i / p.W, u = i - v * p.W;
    int n = 0;
    // every triangle is listed clockwise-on-screen-with-v-down reversed, i.e. with a negative (u, v) cross product
    auto push = [&](int p0, int p1, int p2, bool ok) {
        if (ok) {
            tri[n][0] = p0;
            tri[n][1] = p1;
            tri[n][2] = p2;
            ++n;
        }
    };
    if (v + 1 < p.H && u + 1 < p.W) {
        const int a = i, b = i + 1, c = i + p.W, d = i + p.W + 1;
        const bool va = px_valid(depth, mask, a), vb = px_valid(depth, mask, b), vc = px_valid(depth, mask, c), vd = px_valid(depth, mask, d);
        const int nv = va + vb + vc + vd;
        if (nv >= 3) {
            const bool e_ab = va && vb && edge_ok(p, depth, a, b, 1, 0), e_ac = va && vc && edge_ok(p, depth, a, c, 0, 1);
            const bool e_bd = vb && vd && edge_ok(p, depth, b, d, 0, 1), e_cd = vc && vd && edge_ok(p, depth, c, d, 1, 0);
            const bool e_bc = vb && vc && edge_ok(p, depth, b, c, 1, 1), e_ad = va && vd && edge_ok(p, depth, a, d, 1, 1);
            if (nv == 4) {            // diagonal b - c
                push(a, c, b, e_ab && e_ac && e_bc);
                push(b, c, d, e_bd && e_cd && e_bc);
            } else if (!vd) {
                push(a, c, b, e_ab && e_ac && e_bc);
            } else if (!va) {
                push(b, c, d, e_bd && e_cd && e_bc);
            } else if (!vb) {
                push(a, c, d, e_ac && e_cd && e_ad);
            } else {  // !vc
                push(a, d, b, e_ab && e_bd && e_ad);
            }
        }
    }
    if (!px_valid(depth, mask, i)) {  // a hole: bridged where at least three of its 4-neighbours are valid -- a neighbour outside the image counts as invalid, so a
        // one-pixel hole ON the border with its three in-image neighbours valid gets its (sqrt 2, sqrt 2, 2) triangle too, as Delaunay gives it
        const int nn = i - p.W, ss = i + p.W, ww = i - 1, ee = i + 1;
        const bool vn = v > 0 && px_valid(depth, mask, nn), vs = v + 1 < p.H && px_valid(depth, mask, ss);
        const bool vw = u > 0 && px_valid(depth, mask, ww), ve = u + 1 < p.W && px_valid(depth, mask, ee);
        if (vn + vs + vw + ve >= 3) {
            const bool e_we = vw && ve && edge_ok(p, depth, ww, ee, 2, 0), e_ns = vn && vs && edge_ok(p, depth, nn, ss, 0, 2);
            const bool e_wn = vw && vn && edge_ok(p, depth, ww, nn, 1, 1), e_ne = vn && ve && edge_ok(p, depth, nn, ee, 1, 1);
            const bool e_ws = vw && vs && edge_ok(p, depth, ww, ss, 1, 1), e_se = vs && ve && edge_ok(p, depth, ss, ee, 1, 1);
            if (vw && ve) {           // (all four valid: the west - east diagonal)
                push(ww, ee, nn, vn && e_we && e_wn && e_ne);
                push(ww, ss, ee, vs && e_we && e_ws && e_se);
            } else if (!vw) {
                push(nn, ss, ee, e_ns && e_ne && e_se);
            } else {  // !ve
                push(nn, ww, ss, e_ns && e_wn && e_ws);
            }
        }
    }
    return n;
}

__global__ __launch_bounds__(256) void grid_count_kernel(const float *__restrict__ depth, const uint8_t *__restrict__ mask, GridParams p,
                                                         unsigned *__restrict__ blk_valid, unsigned *__restrict__ blk_faces) {
    __shared__ unsigned lds[8];
    const int n = p.H * p.W;
    unsigned cv = 0, cf = 0;
    for (int j = 0; j < TILE / 256; ++j) {
        const int i = blockIdx.x * TILE + threadIdx.x * (TILE / 256) + j;
        if (i < n) {
            cv += px_valid(depth, mask, i);
            int tri[MAX_PIXEL_FACES][3];
            cf += (unsigned)pixel_faces(p, depth, mask, i, tri);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        cv += (unsigned)__shfl_xor((int)cv, off);
        cf += (unsigned)__shfl_xor((int)cf, off);
    }
    if ((threadIdx.x & 63) == 0) {
        lds[threadIdx.x >> 6] = cv;
        lds[4 + (threadIdx.x >> 6)] = cf;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        blk_valid[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
        blk_faces[blockIdx.x] = lds[4] + lds[5] + lds[6] + lds[7];
    }
}

// exclusive scan of up to a few thousand block counts by one workgroup; totals[which] = sum
__global__ __launch_bounds__(1024) void scan_blocks_kernel(unsigned *__restrict__ a, int nb, unsigned *total) {
    __shared__ unsigned part[1024];
    const int t = threadIdx.x;
    const int per = (nb + 1023) / 1024;
    const int lo = min(t * per, nb), hi = min(lo + per, nb);
    unsigned s = 0;
    for (int i = lo; i < hi; ++i) s += a[i];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        unsigned r = 0;
        for (int i = 0; i < 1024; ++i) {
            const unsigned v = part[i];
            part[i] = r;
            r += v;
        }
        *total = r;
    }
    __syncthreads();
    unsigned r = part[t];
    for (int i = lo; i < hi; ++i) {
        const unsigned v = a[i];
        a[i] = r;
        r += v;
    }
}


// hive_fg_frame_mesh: both block-count arrays scanned by ONE workgroup (threads 0..511: valid pixels, 512..1023: faces), totals[0..1] = sums, and the texture
// window's bounding box initialised for the atomics of window_project_kernel (no host-to-device copy in the call)
__global__ __launch_bounds__(1024) void scan_blocks2_kernel(unsigned *__restrict__ a, unsigned *__restrict__ b, int nb, unsigned *totals, int *bbox) {
    __shared__ unsigned part[1024];
    const int half = threadIdx.x >> 9, t = threadIdx.x & 511;
    unsigned *arr = half ? b : a;
    const int per = (nb + 511) / 512;
    const int lo = min(t * per, nb), hi = min(lo + per, nb);
    unsigned s = 0;
    for (int i = lo; i < hi; ++i) s += arr[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (t == 0) {
        unsigned r = 0;
        for (int i = 0; i < 512; ++i) {
            const unsigned v = part[half * 512 + i];
            part[half * 512 + i] = r;
            r += v;
        }
        totals[half] = r;
    }
    if (threadIdx.x == 1) {
        bbox[0] = bbox[1] = 0x7fffffff;
        bbox[2] = bbox[3] = (int)0x80000000;
    }
    __syncthreads();
    unsigned r = part[threadIdx.x];
    for (int i = lo; i < hi; ++i) {
        const unsigned v = arr[i];
        arr[i] = r;
        r += v;
    }
}

// point_cloud_from_depth's rows (csrc/geometry.hip unproject_write_kernel's arithmetic, verbatim): vertex vid[i] of valid pixel i
struct FrameMeshCam {
    double Kinv[9], R[9], t[3];
};
__global__ __launch_bounds__(256) void grid_vertices_kernel(const float *__restrict__ depth, const int *__restrict__ vid, int n, int W, FrameMeshCam p,
                                                            double *__restrict__ out_xyz, long long capacity) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long id = vid[i];
    if (id < 0 || id >= capacity) return;
    const double d = (double)depth[i];
    const double pu = (double)(i % W), pv = (double)(i / W);
    double cam[3];
    for (int r = 0; r < 3; ++r) cam[r] = d * (p.Kinv[3 * r + 0] * pu + p.Kinv[3 * r + 1] * pv + p.Kinv[3 * r + 2]) - p.t[r];
    for (int r = 0; r < 3; ++r) out_xyz[3 * id + r] = p.R[0 * 3 + r] * cam[0] + p.R[1 * 3 + r] * cam[1] + p.R[2 * 3 + r] * cam[2];
}

// block-wide exclusive offset of this thread's count `c` (4 consecutive items per thread keep row-major order)
__device__ __forceinline__ unsigned block_exclusive(unsigned c, unsigned *lds) {
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
    __syncthreads();
    return before + inc - c;
}

// vid[i] = row of pixel i in point_cloud_from_depth's output, or -1
__global__ __launch_bounds__(256) void grid_vid_kernel(const float *__restrict__ depth, const uint8_t *__restrict__ mask, int n,
                                                       const unsigned *__restrict__ blk_valid, int *__restrict__ vid) {
    __shared__ unsigned lds[4];
    const int base = blockIdx.x * TILE + threadIdx.x * (TILE / 256);
    bool ok[TILE / 256];
    unsigned c = 0;
#pragma unroll
    for (int j = 0; j < TILE / 256; ++j) {
        ok[j] = base + j < n && px_valid(depth, mask, base + j);
        c += ok[j];
    }
    unsigned id = blk_valid[blockIdx.x] + block_exclusive(c, lds);
#pragma unroll
    for (int j = 0; j < TILE / 256; ++j)
        if (base + j < n) vid[base + j] = ok[j] ? (int)id++ : -1;
}

__global__ __launch_bounds__(256) void grid_faces_kernel(const float *__restrict__ depth, const uint8_t *__restrict__ mask, GridParams p,
                                                         const unsigned *__restrict__ blk_faces, const int *__restrict__ vid,
                                                         int32_t *__restrict__ faces, long long capacity) {
    __shared__ unsigned lds[4];
    const int n = p.H * p.W;
    const int base = blockIdx.x * TILE + threadIdx.x * (TILE / 256);
    int tri[TILE / 256][MAX_PIXEL_FACES][3];
    int cnt[TILE / 256];
    unsigned c = 0;
#pragma unroll
    for (int j = 0; j < TILE / 256; ++j) {
        cnt[j] = base + j < n ? pixel_faces(p, depth, mask, base + j, tri[j]) : 0;
        c += (unsigned)cnt[j];
    }
    long long f = (long long)blk_faces[blockIdx.x] + block_exclusive(c, lds);
#pragma unroll
    for (int j = 0; j < TILE / 256; ++j)
        for (int k = 0; k < cnt[j]; ++k, ++f)
            if (f < capacity) {
                faces[3 * f + 0] = vid[tri[j][k][0]];
                faces[3 * f + 1] = vid[tri[j][k][1]];
                faces[3 * f + 2] = vid[tri[j][k][2]];
            }
}

// _filter_faces on an explicit face list (any triangulation): keep face f iff all three edges pass both limits; order kept.
struct FilterParams {
    const int32_t *points2d;  // [n][2] (u, v)
    const float *depth;       // [n]
    const int32_t *faces;     // [F][3]
    long long F;
    double max_px;
    float max_depth;
};

__device__ __forceinline__ bool face_ok(const FilterParams &p, long long f) {
    const int i0 = p.faces[3 * f], i1 = p.faces[3 * f + 1], i2 = p.faces[3 * f + 2];
    const int e[3][2] = {{i0, i1}, {i2, i1}, {i0, i2}};  // faces[:, [0, 2, 0]] against faces[:, [1, 1, 2]]
    for (int k = 0; k < 3; ++k) {
        const int a = e[k][0], b = e[k][1];
        const long long du = (long long)p.points2d[2 * a] - p.points2d[2 * b], dv = (long long)p.points2d[2 * a + 1] - p.points2d[2 * b + 1];
        if (!(sqrt((double)(du * du + dv * dv)) <= p.max_px)) return false;
        if (!(fabsf(p.depth[a] - p.depth[b]) <= p.max_depth)) return false;
    }
    return true;
}

__global__ __launch_bounds__(256) void filter_count_kernel(FilterParams p, unsigned *__restrict__ blk) {
    __shared__ unsigned lds[4];
    unsigned c = 0;
    for (int j = 0; j < TILE / 256; ++j) {
        const long long f = (long long)blockIdx.x * TILE + threadIdx.x * (TILE / 256) + j;
        if (f < p.F) c += face_ok(p, f);
    }
    for (int off = 32; off > 0; off >>= 1) c += (unsigned)__shfl_xor((int)c, off);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

__global__ __launch_bounds__(256) void filter_write_kernel(FilterParams p, const unsigned *__restrict__ blk, int32_t *__restrict__ out) {
    __shared__ unsigned lds[4];
    const long long base = (long long)blockIdx.x * TILE + threadIdx.x * (TILE / 256);
    bool ok[TILE / 256];
    unsigned c = 0;
#pragma unroll
    for (int j = 0; j < TILE / 256; ++j) {
        ok[j] = base + j < p.F && face_ok(p, base + j);
        c += ok[j];
    }
    long long o = (long long)blk[blockIdx.x] + block_exclusive(c, lds);
#pragma unroll
    for (int j = 0; j < TILE / 256; ++j)
        if (ok[j]) {
            out[3 * o + 0] = p.faces[3 * (base + j) + 0];
            out[3 * o + 1] = p.faces[3 * (base + j) + 1];
            out[3 * o + 2] = p.faces[3 * (base + j) + 2];
            ++o;
        }
}

struct WindowParams {
    double K[9], R[9], t[3];
    double scale;
};

// uv = world2image(points) in its default int32 form (np.round, half to even -- geometric.py:175-178); bbox = {min, max} per axis
// (int atomics); out[0..3] = INT_MAX, INT_MAX, INT_MIN, INT_MIN
__global__ __launch_bounds__(256) void window_project_kernel(const double *__restrict__ pts, long long n, WindowParams p, int32_t *__restrict__ uv,
                                                             int *__restrict__ out) {
    int mn_u = 0x7fffffff, mn_v = 0x7fffffff, mx_u = (int)0x80000000, mx_v = (int)0x80000000;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double X[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
        double cam[3], c[3];
        for (int r = 0; r < 3; ++r) cam[r] = p.R[3 * r + 0] * X[0] + p.R[3 * r + 1] * X[1] + p.R[3 * r + 2] * X[2] + p.t[r];
        for (int r = 0; r < 3; ++r) c[r] = p.K[3 * r + 0] * cam[0] + p.K[3 * r + 1] * cam[1] + p.K[3 * r + 2] * cam[2];
        const double u = c[0] / c[2] / p.scale, v = c[1] / c[2] / p.scale;
        const int ru = (int)rint(u), rv = (int)rint(v);  // np.round: half to even
        uv[2 * i + 0] = ru;
        uv[2 * i + 1] = rv;
        mn_u = min(mn_u, ru);
        mx_u = max(mx_u, ru);
        mn_v = min(mn_v, rv);
        mx_v = max(mx_v, rv);
    }
    for (int off = 32; off > 0; off >>= 1) {
        mn_u = min(mn_u, __shfl_xor(mn_u, off));
        mx_u = max(mx_u, __shfl_xor(mx_u, off));
        mn_v = min(mn_v, __shfl_xor(mn_v, off));
        mx_v = max(mx_v, __shfl_xor(mx_v, off));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(out + 0, mn_u);
        atomicMin(out + 1, mn_v);
        atomicMax(out + 2, mx_u);
        atomicMax(out + 3, mx_v);
    }
}

__global__ __launch_bounds__(256) void window_shift_kernel(int32_t *__restrict__ uv, long long n, const int *__restrict__ bbox) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uv[2 * i + 0] -= bbox[0];
    uv[2 * i + 1] -= bbox[1];
}

// the same two steps with the point count in device memory (hive_fg_frame_mesh: the vertex count is known to the device only)
__global__ __launch_bounds__(256) void window_project_dev_kernel(const double *__restrict__ pts, const unsigned *__restrict__ n_ptr, long long capacity, WindowParams p,
                                                                 int32_t *__restrict__ uv, int *__restrict__ out) {
    const long long n = min((long long)*n_ptr, capacity);
    int mn_u = 0x7fffffff, mn_v = 0x7fffffff, mx_u = (int)0x80000000, mx_v = (int)0x80000000;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double X[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
        double cam[3], c[3];
        for (int r = 0; r < 3; ++r) cam[r] = p.R[3 * r + 0] * X[0] + p.R[3 * r + 1] * X[1] + p.R[3 * r + 2] * X[2] + p.t[r];
        for (int r = 0; r < 3; ++r) c[r] = p.K[3 * r + 0] * cam[0] + p.K[3 * r + 1] * cam[1] + p.K[3 * r + 2] * cam[2];
        const double u = c[0] / c[2] / p.scale, v = c[1] / c[2] / p.scale;
        const int ru = (int)rint(u), rv = (int)rint(v);  // np.round: half to even
        uv[2 * i + 0] = ru;
        uv[2 * i + 1] = rv;
        mn_u = min(mn_u, ru);
        mx_u = max(mx_u, ru);
        mn_v = min(mn_v, rv);
        mx_v = max(mx_v, rv);
    }
    for (int off = 32; off > 0; off >>= 1) {
        mn_u = min(mn_u, __shfl_xor(mn_u, off));
        mx_u = max(mx_u, __shfl_xor(mx_u, off));
        mn_v = min(mn_v, __shfl_xor(mn_v, off));
        mx_v = max(mx_v, __shfl_xor(mx_v, off));
    }
    if ((threadIdx.x & 63) == 0 && mn_u != 0x7fffffff) {
        atomicMin(out + 0, mn_u);
        atomicMin(out + 1, mn_v);
        atomicMax(out + 2, mx_u);
        atomicMax(out + 3, mx_v);
    }
}

__global__ __launch_bounds__(256) void window_shift_dev_kernel(int32_t *__restrict__ uv, const unsigned *__restrict__ n_ptr, long long capacity, const int *__restrict__ bbox) {
    const long long n = min((long long)*n_ptr, capacity);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        uv[2 * i + 0] -= bbox[0];
        uv[2 * i + 1] -= bbox[1];
    }
}

}  // namespace

extern "C" {

int hive_grid_mesh(hive_ctx *ctx, const float *depth, const uint8_t *mask, int H, int W, double max_pixel_distance,
                   double max_depth_distance, int mem, int32_t *out_faces, int64_t capacity, int64_t *n_faces, int64_t *n_vertices) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, depth && n_faces, "grid_mesh: NULL argument");
    HIVE_REQUIRE(ctx, H > 0 && W > 0 && (long long)H * W < (1ll << 30), "grid_mesh: bad image size %dx%d", H, W);
    HIVE_REQUIRE(ctx, capacity >= 0 && (capacity == 0 || out_faces), "grid_mesh: bad output buffer");
    HIVE_REQUIRE(ctx, mem == HIVE_MEM_HOST || mem == HIVE_MEM_DEVICE, "grid_mesh: bad mem kind %d", mem);
    const int n = H * W, nb = (n + TILE - 1) / TILE;
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    // device scratch: [depth | mask] (host inputs) | block counts x 2 | vid | faces (host outputs)
    const size_t off_mask = align((size_t)n * 4), off_bv = off_mask + align((size_t)n), off_bf = off_bv + align((size_t)nb * 4);
    const size_t off_vid = off_bf + align((size_t)nb * 4), off_faces = off_vid + align((size_t)n * 4);
    const size_t total = off_faces + (mem == HIVE_MEM_HOST ? align((size_t)capacity * 12) : 0);
    int rc = hive_reserve_device(ctx, &ctx->d_scratch, &ctx->scratch_bytes, total);
    if (rc) return rc;
    char *base = (char *)ctx->d_scratch;
    const float *d_depth = depth;
    const uint8_t *d_mask = mask;
    int32_t *d_faces = out_faces;
    if (mem == HIVE_MEM_HOST) {
        if ((rc = hive_upload(ctx, base, depth, (size_t)n * 4))) return rc;
        d_depth = (const float *)base;
        if (mask) {
            if ((rc = hive_upload(ctx, base + off_mask, mask, (size_t)n))) return rc;
            d_mask = (const uint8_t *)(base + off_mask);
        }
        d_faces = (int32_t *)(base + off_faces);
    }
    unsigned *bv = (unsigned *)(base + off_bv), *bf = (unsigned *)(base + off_bf);
    int *vid = (int *)(base + off_vid);
    unsigned *d_tot = ctx->d_scalars + 32;  // [32] = vertices, [33] = faces
    GridParams p{H, W, max_pixel_distance, (float)max_depth_distance};
    hipLaunchKernelGGL(grid_count_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_depth, d_mask, p, bv, bf);
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, ctx->stream, bv, nb, d_tot);
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, ctx->stream, bf, nb, d_tot + 1);
    hipLaunchKernelGGL(grid_vid_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_depth, d_mask, n, (const unsigned *)bv, vid);
    if (capacity > 0)
        hipLaunchKernelGGL(grid_faces_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_depth, d_mask, p, (const unsigned *)bf, (const int *)vid, d_faces,
                           (long long)capacity);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    unsigned tot[2] = {0, 0};
    HIVE_CHECK_HIP(ctx, hipMemcpyAsync(tot, d_tot, sizeof(tot), hipMemcpyDeviceToHost, ctx->stream));
    HIVE_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_faces = tot[1];
    if (n_vertices) *n_vertices = tot[0];
    if (mem == HIVE_MEM_HOST && capacity > 0) {
        const size_t nf = (size_t)std::min<int64_t>(capacity, (int64_t)tot[1]);
        if (nf) HIVE_CHECK_HIP(ctx, hipMemcpy(out_faces, d_faces, nf * 12, hipMemcpyDeviceToHost));
    }
    return HIVE_OK;
}

int hive_fg_frame_mesh(hive_ctx *ctx, const float *d_depth, const uint8_t *d_mask, int H, int W, const double Kinv[9], const double K[9], const double R[9],
                       const double t[3], double max_pixel_distance, double max_depth_distance, double *d_vertices, int64_t vertex_capacity, int32_t *d_faces,
                       int64_t face_capacity, int32_t *d_uv, int64_t *n_vertices, int64_t *n_faces, int32_t bbox[4]) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, d_depth && Kinv && K && R && t && d_vertices && d_faces && d_uv && n_vertices && n_faces && bbox, "fg_frame_mesh: NULL argument");
    HIVE_REQUIRE(ctx, H > 0 && W > 0 && (long long)H * W < (1ll << 30), "fg_frame_mesh: bad image size %dx%d", H, W);
    HIVE_REQUIRE(ctx, vertex_capacity > 0 && face_capacity > 0, "fg_frame_mesh: empty output buffers");
    const int n = H * W, nb = (n + TILE - 1) / TILE;
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t off_bf = align((size_t)nb * 4), off_vid = off_bf + align((size_t)nb * 4);
    int rc = hive_reserve_device(ctx, &ctx->d_scratch, &ctx->scratch_bytes, off_vid + align((size_t)n * 4));
    if (rc) return rc;
    char *base = (char *)ctx->d_scratch;
    unsigned *bv = (unsigned *)base, *bf = (unsigned *)(base + off_bf);
    int *vid = (int *)(base + off_vid);
    unsigned *d_tot = ctx->d_scalars + 32;  // [32] = vertices, [33] = faces, [34..37] = the texture window's box
    int *d_box = (int *)(ctx->d_scalars + 34);
    GridParams p{H, W, max_pixel_distance, (float)max_depth_distance};
    FrameMeshCam cam;
    memcpy(cam.Kinv, Kinv, sizeof(cam.Kinv));
    memcpy(cam.R, R, sizeof(cam.R));
    memcpy(cam.t, t, sizeof(cam.t));
    WindowParams wp;
    memcpy(wp.K, K, sizeof(wp.K));
    memcpy(wp.R, R, sizeof(wp.R));
    memcpy(wp.t, t, sizeof(wp.t));
    wp.scale = 1.0;
    hipLaunchKernelGGL(grid_count_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_depth, d_mask, p, bv, bf);
    hipLaunchKernelGGL(scan_blocks2_kernel, dim3(1), dim3(1024), 0, ctx->stream, bv, bf, nb, d_tot, d_box);
    hipLaunchKernelGGL(grid_vid_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_depth, d_mask, n, (const unsigned *)bv, vid);
    hipLaunchKernelGGL(grid_vertices_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_depth, (const int *)vid, n, W, cam, d_vertices, (long long)vertex_capacity);
    hipLaunchKernelGGL(grid_faces_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_depth, d_mask, p, (const unsigned *)bf, (const int *)vid, d_faces,
                       (long long)face_capacity);
    const dim3 wgrid((unsigned)std::min<long long>((std::min<long long>(n, vertex_capacity) + 255) / 256, (long long)ctx->num_cus * 4));
    hipLaunchKernelGGL(window_project_dev_kernel, wgrid, dim3(256), 0, ctx->stream, (const double *)d_vertices, (const unsigned *)d_tot, (long long)vertex_capacity, wp, d_uv,
                       d_box);
    hipLaunchKernelGGL(window_shift_dev_kernel, wgrid, dim3(256), 0, ctx->stream, d_uv, (const unsigned *)d_tot, (long long)vertex_capacity, (const int *)d_box);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    // ONE read-back for the whole frame mesh: {vertices, faces, box[4]} through pinned memory
    if (!ctx->h_pinned_small) HIVE_CHECK_HIP(ctx, hipHostMalloc(&ctx->h_pinned_small, 256, hipHostMallocDefault));
    HIVE_CHECK_HIP(ctx, hipMemcpyAsync(ctx->h_pinned_small, d_tot, 6 * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    HIVE_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const unsigned *tot = (const unsigned *)ctx->h_pinned_small;
    const int *box = (const int *)(tot + 2);
    *n_vertices = tot[0];
    *n_faces = tot[1];
    bbox[0] = box[0];
    bbox[1] = box[1];
    bbox[2] = tot[0] ? box[2] + 1 : box[2];
    bbox[3] = tot[0] ? box[3] + 1 : box[3];
    HIVE_REQUIRE(ctx, (int64_t)tot[0] <= vertex_capacity && (int64_t)tot[1] <= face_capacity, "fg_frame_mesh: %u vertices / %u faces do not fit the buffers (%lld / %lld)",
                 tot[0], tot[1], (long long)vertex_capacity, (long long)face_capacity);
    return HIVE_OK;
}

int hive_filter_faces(hive_ctx *ctx, const int32_t *points2d, const float *depth, int64_t n_points, const int32_t *faces, int64_t n_faces_in,
                      double max_pixel_distance, double max_depth_distance, int mem, int32_t *out_faces, int64_t *n_faces_out) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, n_faces_out && n_points >= 0 && n_faces_in >= 0, "filter_faces: bad arguments");
    *n_faces_out = 0;
    if (n_faces_in == 0) return HIVE_OK;
    HIVE_REQUIRE(ctx, points2d && depth && faces && out_faces && n_points > 0, "filter_faces: NULL argument");
    HIVE_REQUIRE(ctx, mem == HIVE_MEM_HOST || mem == HIVE_MEM_DEVICE, "filter_faces: bad mem kind %d", mem);
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const int nb = (int)((n_faces_in + TILE - 1) / TILE);
    const size_t off_depth = align((size_t)n_points * 8), off_faces = off_depth + align((size_t)n_points * 4);
    const size_t off_out = off_faces + align((size_t)n_faces_in * 12), off_blk = off_out + align((size_t)n_faces_in * 12);
    int rc = hive_reserve_device(ctx, &ctx->d_scratch, &ctx->scratch_bytes, off_blk + (size_t)nb * 4);
    if (rc) return rc;
    char *base = (char *)ctx->d_scratch;
    FilterParams p{points2d, depth, faces, (long long)n_faces_in, max_pixel_distance, (float)max_depth_distance};
    int32_t *d_out = out_faces;
    if (mem == HIVE_MEM_HOST) {
        if ((rc = hive_upload(ctx, base, points2d, (size_t)n_points * 8))) return rc;
        if ((rc = hive_upload(ctx, base + off_depth, depth, (size_t)n_points * 4))) return rc;
        if ((rc = hive_upload(ctx, base + off_faces, faces, (size_t)n_faces_in * 12))) return rc;
        p.points2d = (const int32_t *)base;
        p.depth = (const float *)(base + off_depth);
        p.faces = (const int32_t *)(base + off_faces);
        d_out = (int32_t *)(base + off_out);
    }
    unsigned *blk = (unsigned *)(base + off_blk), *d_tot = ctx->d_scalars + 34;
    hipLaunchKernelGGL(filter_count_kernel, dim3(nb), dim3(256), 0, ctx->stream, p, blk);
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, ctx->stream, blk, nb, d_tot);
    hipLaunchKernelGGL(filter_write_kernel, dim3(nb), dim3(256), 0, ctx->stream, p, (const unsigned *)blk, d_out);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    unsigned tot = 0;
    HIVE_CHECK_HIP(ctx, hipMemcpyAsync(&tot, d_tot, sizeof(tot), hipMemcpyDeviceToHost, ctx->stream));
    HIVE_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_faces_out = tot;
    if (mem == HIVE_MEM_HOST && tot) HIVE_CHECK_HIP(ctx, hipMemcpy(out_faces, d_out, (size_t)tot * 12, hipMemcpyDeviceToHost));
    return HIVE_OK;
}

int hive_texture_window(hive_ctx *ctx, const double *points, int64_t n, const double K[9], const double R[9], const double t[3],
                        double scale_factor, int mem, int32_t *out_uv, int32_t bbox[4]) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, points && K && R && t && out_uv && bbox && n > 0, "texture_window: bad arguments");
    HIVE_REQUIRE(ctx, scale_factor != 0.0, "texture_window: scale_factor must not be 0");
    int rc;
    const double *d_pts = points;
    int32_t *d_uv = out_uv;
    if (mem == HIVE_MEM_HOST) {
        const size_t off_uv = ((size_t)n * 24 + 255) & ~(size_t)255;
        if ((rc = hive_reserve_device(ctx, &ctx->d_in, &ctx->in_bytes, off_uv + (size_t)n * 8))) return rc;
        if ((rc = hive_upload(ctx, ctx->d_in, points, (size_t)n * 24))) return rc;
        d_pts = (const double *)ctx->d_in;
        d_uv = (int32_t *)((char *)ctx->d_in + off_uv);
    }
    int *d_box = (int *)(ctx->d_scalars + 40);
    const int32_t init[4] = {0x7fffffff, 0x7fffffff, (int32_t)0x80000000, (int32_t)0x80000000};
    HIVE_CHECK_HIP(ctx, hipMemcpyAsync(d_box, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    WindowParams p;
    memcpy(p.K, K, sizeof(p.K));
    memcpy(p.R, R, sizeof(p.R));
    memcpy(p.t, t, sizeof(p.t));
    p.scale = scale_factor;
    const dim3 grid((unsigned)std::min<long long>((n + 255) / 256, (long long)ctx->num_cus * 4));
    hipLaunchKernelGGL(window_project_kernel, grid, dim3(256), 0, ctx->stream, d_pts, (long long)n, p, d_uv, d_box);
    hipLaunchKernelGGL(window_shift_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_uv, (long long)n, (const int *)d_box);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    int32_t box[4];
    HIVE_CHECK_HIP(ctx, hipMemcpyAsync(box, d_box, sizeof(box), hipMemcpyDeviceToHost, ctx->stream));
    HIVE_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // min_u, min_v, max_u + 1, max_v + 1: `texture = image[min_v:max_v, min_u:max_u]` (pipeline.py:802-805)
    bbox[0] = box[0];
    bbox[1] = box[1];
    bbox[2] = box[2] + 1;
    bbox[3] = box[3] + 1;
    if (mem == HIVE_MEM_HOST) HIVE_CHECK_HIP(ctx, hipMemcpy(out_uv, d_uv, (size_t)n * 8, hipMemcpyDeviceToHost));
    return HIVE_OK;
}

}  // extern "C"
