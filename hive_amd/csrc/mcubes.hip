// Marching cubes on the TSDF volume (level 0): wave-scan stream compaction, one vertex per
// sign-changing grid edge.  gfx950 only.
//
// Replaces `fusion.TSDFVolume.get_mesh()` of the reference's (absent) third_party/tsdf_fusion_python
// (call site /root/reference/hive/fusion.py:127), which wraps scikit-image's Lewiner marching cubes.
// Table conventions and ordering: tools/gen_mc_tables.py and include/hive_mi355x.h.
//
// Passes:
//   signs   ONE coalesced sweep over the float32 tsdf volume -> 1 bit per voxel (tsdf < 0), 32 consecutive z per word
//   count   per word (lane): the sign words of the 4 rows (x, y), (x+1, y), (x, y+1), (x+1, y+1) and their successors give
//           every owned sign-changing edge and every non-uniform cell of 32 voxels with a few XOR / popcount
//           instructions; the triangle count loops over the (rare) active cells only
//   scan    exclusive scan of the per-block counts (one workgroup)
//   verts   per-block scan -> vertex id; writes position / normal / colour and, per voxel, vbase = first vertex id |
//           owned-axis mask << 29 (only where the mask is non-zero)
//   faces   per-block scan -> triangle id; vertex ids looked up through vbase
// Only `signs` reads the whole volume (4 N bytes); the other passes read the N / 8-byte bitmap and touch tsdf / colour
// for surface voxels only.  (The first version classified every voxel from 8 scalar float loads in each of the three
// passes: 0.8-1.0 ms per pass at 512^3, bound by the CU's vector-memory instruction rate, not by bytes.)
#include "hive_internal.hpp"
#include "../../include/hive_mc_tables.h"

#include <algorithm>

__constant__ unsigned char c_num_tris[256];
__constant__ unsigned char c_tri_table[256][3 * HIVE_MC_MAX_TRIS];
__constant__ unsigned char c_edge_owner[12][4];

struct McParams {
    const float *tsdf;
    const float *color;
    const unsigned *sign;  // 1 bit per voxel (tsdf < 0), rows padded to WZ words
    int WZ;
    int X, Y, Z;
    long long n;
    float ox, oy, oz, vs;
};

constexpr int MC_BLOCK = 256;

// ---- sign bitmap ---------------------------------------------------------------------------------
// fast path (Z % 32 == 0): a lane loads 4 consecutive voxels (16 B, coalesced), 8 lanes assemble one word
__global__ __launch_bounds__(256) void mc_signs_vec_kernel(const float *__restrict__ tsdf, long long n, unsigned *__restrict__ sign) {
    const long long i4 = (long long)blockIdx.x * 256 + threadIdx.x;  // group of 4 voxels
    unsigned v = 0;
    if (i4 * 4 < n) {
        const float4 t = *reinterpret_cast<const float4 *>(tsdf + i4 * 4);
        v = (unsigned)(t.x < 0.f) | ((unsigned)(t.y < 0.f) << 1) | ((unsigned)(t.z < 0.f) << 2) | ((unsigned)(t.w < 0.f) << 3);
    }
    v <<= 4 * (threadIdx.x & 7);
    v |= (unsigned)__shfl_xor((int)v, 1);
    v |= (unsigned)__shfl_xor((int)v, 2);
    v |= (unsigned)__shfl_xor((int)v, 4);
    if ((threadIdx.x & 7) == 0 && i4 * 4 < n) sign[i4 >> 3] = v;
}

// general path: one lane per word, rows padded to whole words (bits past Z are 0)
__global__ __launch_bounds__(256) void mc_signs_row_kernel(const float *__restrict__ tsdf, int Z, int WZ, long long n_words,
                                                           unsigned *__restrict__ sign) {
    const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
    if (w >= n_words) return;
    const long long row = w / WZ;
    const int z0 = (int)(w % WZ) * 32;
    const float *src = tsdf + row * Z + z0;
    unsigned v = 0;
    for (int b = 0; b < 32 && z0 + b < Z; ++b) v |= (unsigned)(src[b] < 0.f) << b;
    sign[w] = v;
}

// ---- per-word classification ----------------------------------------------------------------------
struct WordClass {
    unsigned xe, ye, ze;  // bit b set: voxel z0 + b owns a sign-changing edge towards +x / +y / +z
    unsigned act;         // bit b set: the cell whose corner 0 is voxel z0 + b is cut by the surface
    unsigned c[8];        // sign words of the 8 cell corners (marching-cubes corner order), aligned to corner 0
    int x, y, z0;
    long long row;        // x * Y + y
};

__device__ __forceinline__ WordClass classify_word(const McParams &p, long long w, bool need_cells) {
    WordClass k;
    k.row = w / p.WZ;
    const int wz = (int)(w % p.WZ);
    k.z0 = wz * 32;
    k.y = (int)(k.row % p.Y);
    k.x = (int)(k.row / p.Y);
    const bool hx = k.x + 1 < p.X, hy = k.y + 1 < p.Y, hw = wz + 1 < p.WZ;
    const long long sx = (long long)p.Y * p.WZ, sy = p.WZ;
    const int left = p.Z - k.z0;  // voxels of the row from z0 on (>= 1)
    const unsigned in_mask = left >= 32 ? 0xffffffffu : (1u << left) - 1u;                    // z < Z
    const unsigned vz_mask = left - 1 >= 32 ? 0xffffffffu : (1u << max(left - 1, 0)) - 1u;    // z + 1 < Z
    const unsigned w00 = p.sign[w];
    const unsigned w10 = hx ? p.sign[w + sx] : w00;
    const unsigned w01 = hy ? p.sign[w + sy] : w00;
    const unsigned n00 = hw ? p.sign[w + 1] : 0u;
    const unsigned s00 = (w00 >> 1) | (n00 << 31);
    k.xe = (w00 ^ w10) & in_mask;
    k.ye = (w00 ^ w01) & in_mask;
    k.ze = (w00 ^ s00) & vz_mask;
    k.act = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) k.c[i] = 0;
    if (need_cells && hx && hy) {
        const unsigned w11 = p.sign[w + sx + sy];
        const unsigned n10 = hw ? p.sign[w + sx + 1] : 0u, n01 = hw ? p.sign[w + sy + 1] : 0u, n11 = hw ? p.sign[w + sx + sy + 1] : 0u;
        k.c[0] = w00;
        k.c[1] = w10;
        k.c[2] = w11;
        k.c[3] = w01;
        k.c[4] = s00;
        k.c[5] = (w10 >> 1) | (n10 << 31);
        k.c[6] = (w11 >> 1) | (n11 << 31);
        k.c[7] = (w01 >> 1) | (n01 << 31);
        unsigned any = 0u, all = 0xffffffffu;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            any |= k.c[i];
            all &= k.c[i];
        }
        k.act = any & ~all & vz_mask;
    }
    return k;
}

__device__ __forceinline__ int cell_case(const WordClass &k, int b) {
    int cs = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) cs |= (int)((k.c[i] >> b) & 1u) << i;
    return cs;
}

// exclusive scan of one unsigned per thread over the 256-thread block; returns the block total in `total`
__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned *lds4, unsigned &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = (unsigned)__shfl_up((int)inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) lds4[wave] = inc;
    __syncthreads();
    unsigned before = 0;
    total = 0;
    for (int w = 0; w < MC_BLOCK / 64; ++w) {
        if (w < wave) before += lds4[w];
        total += lds4[w];
    }
    __syncthreads();
    return before + inc - v;
}

__global__ __launch_bounds__(MC_BLOCK) void mc_count_kernel(McParams p, long long n_words, unsigned *__restrict__ blk_v,
                                                            unsigned *__restrict__ blk_t) {
    __shared__ unsigned lds[8];
    const long long w = (long long)blockIdx.x * MC_BLOCK + threadIdx.x;
    unsigned nv = 0, nt = 0;
    if (w < n_words) {
        const WordClass k = classify_word(p, w, true);
        nv = __popc(k.xe) + __popc(k.ye) + __popc(k.ze);
        for (unsigned m = k.act; m; m &= m - 1u) nt += c_num_tris[cell_case(k, __ffs((int)m) - 1)];
    }
    for (int off = 32; off > 0; off >>= 1) {
        nv += (unsigned)__shfl_xor((int)nv, off);
        nt += (unsigned)__shfl_xor((int)nt, off);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        lds[wave] = nv;
        lds[4 + wave] = nt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        blk_v[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
        blk_t[blockIdx.x] = lds[4] + lds[5] + lds[6] + lds[7];
    }
}

// exclusive scan of two arrays of nb counts in place; totals (u64) to totals[0], totals[1]
__global__ __launch_bounds__(1024) void mc_scan_kernel(unsigned *__restrict__ a, unsigned *__restrict__ b, int nb,
                                                       unsigned long long *totals) {
    __shared__ unsigned long long part[2][1024];
    const int t = threadIdx.x;
    const int per = (nb + 1023) / 1024;
    const int lo = min(t * per, nb), hi = min(lo + per, nb);
    unsigned long long sa = 0, sb = 0;
    for (int i = lo; i < hi; ++i) {
        sa += a[i];
        sb += b[i];
    }
    part[0][t] = sa;
    part[1][t] = sb;
    __syncthreads();
    if (t == 0) {
        unsigned long long ra = 0, rb = 0;
        for (int i = 0; i < 1024; ++i) {
            const unsigned long long ta = part[0][i], tb = part[1][i];
            part[0][i] = ra;
            part[1][i] = rb;
            ra += ta;
            rb += tb;
        }
        totals[0] = ra;
        totals[1] = rb;
    }
    __syncthreads();
    unsigned long long ra = part[0][t], rb = part[1][t];
    for (int i = lo; i < hi; ++i) {
        const unsigned va = a[i], vb = b[i];
        a[i] = (unsigned)ra;
        b[i] = (unsigned)rb;
        ra += va;
        rb += vb;
    }
}

__device__ __forceinline__ float grad_axis(const McParams &p, int x, int y, int z, int axis) {
    int lo[3] = {x, y, z}, hi[3] = {x, y, z};
    const int d[3] = {p.X, p.Y, p.Z};
    const int c = axis == 0 ? x : (axis == 1 ? y : z);
    if (c > 0) lo[axis] -= 1;
    if (c < d[axis] - 1) hi[axis] += 1;
    const float a = p.tsdf[((long long)hi[0] * p.Y + hi[1]) * p.Z + hi[2]];
    const float b = p.tsdf[((long long)lo[0] * p.Y + lo[1]) * p.Z + lo[2]];
    const float span = (float)(hi[axis] - lo[axis]);
    return span > 0.0f ? (a - b) / span : 0.0f;
}

__global__ __launch_bounds__(MC_BLOCK) void mc_verts_kernel(McParams p, long long n_words, const unsigned *__restrict__ blk_v,
                                                            unsigned *__restrict__ vbase, float *__restrict__ verts,
                                                            float *__restrict__ verts_vox, float *__restrict__ norms,
                                                            uint8_t *__restrict__ colors) {
    __shared__ unsigned lds[4];
    const long long w = (long long)blockIdx.x * MC_BLOCK + threadIdx.x;
    WordClass k;
    k.xe = k.ye = k.ze = 0;
    if (w < n_words) k = classify_word(p, w, false);
    const unsigned cnt = __popc(k.xe) + __popc(k.ye) + __popc(k.ze);
    unsigned total;
    unsigned id = blk_v[blockIdx.x] + block_exclusive_scan(cnt, lds, total);
    for (unsigned m = k.xe | k.ye | k.ze; m; m &= m - 1u) {  // voxels in ascending z, axes in x, y, z order
        const int b = __ffs((int)m) - 1;
        const unsigned mask = ((k.xe >> b) & 1u) | (((k.ye >> b) & 1u) << 1) | (((k.ze >> b) & 1u) << 2);
        const int x = k.x, y = k.y, z = k.z0 + b;
        const long long idx = k.row * p.Z + z;
        vbase[idx] = id | (mask << 29);
        const float v0 = p.tsdf[idx];
        const long long stride[3] = {(long long)p.Y * p.Z, p.Z, 1};
        for (int a = 0; a < 3; ++a) {
            if (!((mask >> a) & 1u)) continue;
            const float v1 = p.tsdf[idx + stride[a]];
            const float t = v0 / (v0 - v1);
            float pos[3] = {(float)x, (float)y, (float)z};
            pos[a] = pos[a] + t;
            int q[3] = {x, y, z};
            q[a] += 1;
            float g[3];
            for (int r = 0; r < 3; ++r) {
                const float g0 = grad_axis(p, x, y, z, r);
                const float g1 = grad_axis(p, q[0], q[1], q[2], r);
                g[r] = g0 + t * (g1 - g0);
            }
            const float len = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
            const float org[3] = {p.ox, p.oy, p.oz};
            const int dim[3] = {p.X, p.Y, p.Z};
            int ci[3];
            for (int r = 0; r < 3; ++r) {
                verts_vox[3ll * id + r] = pos[r];
                verts[3ll * id + r] = pos[r] * p.vs + org[r];
                norms[3ll * id + r] = len > 0.0f ? g[r] / len : 0.0f;
                ci[r] = min((int)rintf(pos[r]), dim[r] - 1);
            }
            const unsigned c = (unsigned)p.color[((long long)ci[0] * p.Y + ci[1]) * p.Z + ci[2]];
            colors[3ll * id + 0] = (uint8_t)(c & 255u);
            colors[3ll * id + 1] = (uint8_t)((c >> 8) & 255u);
            colors[3ll * id + 2] = (uint8_t)(c >> 16);
            ++id;
        }
    }
}

__global__ __launch_bounds__(MC_BLOCK) void mc_faces_kernel(McParams p, long long n_words, const unsigned *__restrict__ blk_t,
                                                            const unsigned *__restrict__ vbase, int32_t *__restrict__ faces) {
    __shared__ unsigned lds[4];
    const long long w = (long long)blockIdx.x * MC_BLOCK + threadIdx.x;
    WordClass k;
    k.act = 0;
    if (w < n_words) k = classify_word(p, w, true);
    unsigned cnt = 0;
    for (unsigned m = k.act; m; m &= m - 1u) cnt += c_num_tris[cell_case(k, __ffs((int)m) - 1)];
    unsigned total;
    unsigned long long fid = blk_t[blockIdx.x] + block_exclusive_scan(cnt, lds, total);
    for (unsigned m = k.act; m; m &= m - 1u) {  // cells in ascending z
        const int b = __ffs((int)m) - 1;
        const int cs = cell_case(k, b);
        const int nt = c_num_tris[cs];
        const long long idx = k.row * p.Z + k.z0 + b;
        for (int t = 0; t < nt; ++t) {
            for (int v = 0; v < 3; ++v) {
                const int e = c_tri_table[cs][3 * t + v];
                const long long oi = idx + ((long long)c_edge_owner[e][0] * p.Y + c_edge_owner[e][1]) * p.Z + c_edge_owner[e][2];
                const unsigned vb = vbase[oi];
                const unsigned am = vb >> 29;
                const int a = c_edge_owner[e][3];
                faces[3 * fid + v] = (int32_t)((vb & 0x1fffffffu) + __popc(am & ((1u << a) - 1u)));
            }
            ++fid;
        }
    }
}

void hive_tsdf_free_mesh(hive_tsdf *v) {
    if (v->d_verts) (void)hipFree(v->d_verts);
    if (v->d_norms) (void)hipFree(v->d_norms);
    if (v->d_verts_vox) (void)hipFree(v->d_verts_vox);
    if (v->d_faces) (void)hipFree(v->d_faces);
    if (v->d_vcolors) (void)hipFree(v->d_vcolors);
    v->d_verts = v->d_norms = v->d_verts_vox = nullptr;
    v->d_faces = nullptr;
    v->d_vcolors = nullptr;
    v->cap_verts = v->cap_faces = 0;
    v->n_verts = v->n_faces = -1;
}

// Result arrays of a volume are kept from one extraction to the next and only grow (with a quarter of head room): five hipMalloc /
// hipFree pairs per call cost more than the kernels at 512^3 (round 3: 2.0 ms end to end for 0.6 ms of kernels).
static int reserve_mesh(hive_tsdf *v, size_t nv, size_t nf) {
    hive_ctx *ctx = v->ctx;
    if ((size_t)v->cap_verts >= nv && (size_t)v->cap_faces >= nf && v->d_verts) return HIVE_OK;
    HIVE_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    hive_tsdf_free_mesh(v);
    const size_t cv = nv + nv / 4 + 1024, cf = nf + nf / 4 + 1024;
    HIVE_CHECK_HIP(ctx, hipMalloc((void **)&v->d_verts, cv * 3 * sizeof(float)));
    HIVE_CHECK_HIP(ctx, hipMalloc((void **)&v->d_verts_vox, cv * 3 * sizeof(float)));
    HIVE_CHECK_HIP(ctx, hipMalloc((void **)&v->d_norms, cv * 3 * sizeof(float)));
    HIVE_CHECK_HIP(ctx, hipMalloc((void **)&v->d_vcolors, cv * 3));
    HIVE_CHECK_HIP(ctx, hipMalloc((void **)&v->d_faces, cf * 3 * sizeof(int32_t)));
    v->cap_verts = (int64_t)cv;
    v->cap_faces = (int64_t)cf;
    return HIVE_OK;
}

static bool g_tables_uploaded[64] = {false};

static int upload_tables(hive_ctx *ctx) {
    if (ctx->device < 64 && g_tables_uploaded[ctx->device]) return HIVE_OK;
    HIVE_CHECK_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_num_tris), HIVE_MC_NUM_TRIS, sizeof(HIVE_MC_NUM_TRIS)));
    HIVE_CHECK_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_tri_table), HIVE_MC_TRI_TABLE, sizeof(HIVE_MC_TRI_TABLE)));
    HIVE_CHECK_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_edge_owner), HIVE_MC_EDGE_OWNER, sizeof(HIVE_MC_EDGE_OWNER)));
    if (ctx->device < 64) g_tables_uploaded[ctx->device] = true;
    return HIVE_OK;
}

extern "C" {

int hive_tsdf_extract_mesh(hive_tsdf *v, int64_t *n_verts, int64_t *n_faces) {
    HIVE_ENTER(v ? v->ctx : nullptr);
    if (v && (v->x_off != 0 || v->dim[0] != v->grid_dim0))
        return hive_fail(v->ctx, HIVE_ERR_STATE, "extract_mesh: this volume is an x-slab [%lld, %lld) of a %lld-wide grid; gather the slabs into a "
                                                 "whole volume first", (long long)v->x_off, (long long)(v->x_off + v->dim[0]), (long long)v->grid_dim0);
    if (!v) return hive_fail(nullptr, HIVE_ERR_INVALID, "vol is NULL");
    hive_ctx *ctx = v->ctx;
    HIVE_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    int rc = upload_tables(ctx);
    if (rc) return rc;
    v->n_verts = v->n_faces = -1;
    McParams p;
    p.tsdf = v->d_tsdf;
    p.color = v->d_color;
    p.X = (int)v->dim[0];
    p.Y = (int)v->dim[1];
    p.Z = (int)v->dim[2];
    p.n = v->n;
    p.ox = v->origin[0];
    p.oy = v->origin[1];
    p.oz = v->origin[2];
    p.vs = v->voxel_size;
    p.WZ = (p.Z + 31) / 32;
    const long long n_words = (long long)p.X * p.Y * p.WZ;
    const long long nb = (n_words + MC_BLOCK - 1) / MC_BLOCK;
    HIVE_REQUIRE(ctx, nb < (1ll << 31), "volume too large for mesh extraction");
    if ((rc = hive_reserve_device(ctx, &ctx->d_scratch, &ctx->scratch_bytes, (size_t)n_words * sizeof(unsigned)))) return rc;
    unsigned *d_sign = (unsigned *)ctx->d_scratch;
    p.sign = d_sign;
    if (p.Z % 32 == 0 && ((uintptr_t)v->d_tsdf % 16) == 0)
        hipLaunchKernelGGL(mc_signs_vec_kernel, dim3((unsigned)((v->n / 4 + 255) / 256)), dim3(256), 0, ctx->stream, v->d_tsdf, (long long)v->n, d_sign);
    else
        hipLaunchKernelGGL(mc_signs_row_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, ctx->stream, v->d_tsdf, p.Z, p.WZ, n_words, d_sign);
    if ((rc = hive_reserve_device(ctx, (void **)&v->d_blk, &v->blk_bytes, 2 * (size_t)nb * sizeof(unsigned)))) return rc;
    unsigned *blk_v = v->d_blk, *blk_t = v->d_blk + nb;
    unsigned long long *d_tot = (unsigned long long *)(ctx->d_scalars + 8);
    hipLaunchKernelGGL(mc_count_kernel, dim3((unsigned)nb), dim3(MC_BLOCK), 0, ctx->stream, p, n_words, blk_v, blk_t);
    hipLaunchKernelGGL(mc_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, blk_v, blk_t, (int)nb, d_tot);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    // the ONE read-back of an extraction: vertex and face totals, through pinned memory (a copy into pageable memory is staged by the runtime)
    if (!ctx->h_pinned_small) HIVE_CHECK_HIP(ctx, hipHostMalloc(&ctx->h_pinned_small, 256, hipHostMallocDefault));
    volatile unsigned long long *tot = (volatile unsigned long long *)ctx->h_pinned_small;
    HIVE_CHECK_HIP(ctx, hipMemcpyAsync(ctx->h_pinned_small, d_tot, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIVE_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (tot[0] == 0) {
        v->n_verts = v->n_faces = -1;
        return hive_fail(ctx, HIVE_ERR_EMPTY, "Surface level must be within volume data range.");
    }
    HIVE_REQUIRE(ctx, tot[0] < (1ull << 29) && 3 * tot[1] < (1ull << 31), "mesh too large: %llu vertices, %llu faces", (unsigned long long)tot[0],
                 (unsigned long long)tot[1]);
    if ((rc = hive_reserve_device(ctx, (void **)&v->d_vbase, &v->vbase_bytes, (size_t)v->n * sizeof(unsigned)))) return rc;
    const size_t nv = tot[0], nf = tot[1];
    if ((rc = reserve_mesh(v, nv, std::max<size_t>(nf, 1)))) return rc;
    hipLaunchKernelGGL(mc_verts_kernel, dim3((unsigned)nb), dim3(MC_BLOCK), 0, ctx->stream, p, n_words, blk_v, v->d_vbase, v->d_verts,
                       v->d_verts_vox, v->d_norms, v->d_vcolors);
    hipLaunchKernelGGL(mc_faces_kernel, dim3((unsigned)nb), dim3(MC_BLOCK), 0, ctx->stream, p, n_words, blk_t, v->d_vbase, v->d_faces);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    v->n_verts = (int64_t)nv;
    v->n_faces = (int64_t)nf;
    if (n_verts) *n_verts = v->n_verts;
    if (n_faces) *n_faces = v->n_faces;
    return HIVE_OK;
}

int hive_tsdf_copy_mesh(hive_tsdf *v, float *verts, int32_t *faces, float *norms, uint8_t *colors) {
    HIVE_ENTER(v ? v->ctx : nullptr);
    if (!v) return hive_fail(nullptr, HIVE_ERR_INVALID, "vol is NULL");
    hive_ctx *ctx = v->ctx;
    if (v->n_verts < 0) return hive_fail(ctx, HIVE_ERR_STATE, "copy_mesh: call hive_tsdf_extract_mesh first");
    const size_t nv = (size_t)v->n_verts, nf = (size_t)v->n_faces;
    if (verts) HIVE_CHECK_HIP(ctx, hipMemcpyAsync(verts, v->d_verts, nv * 12, hipMemcpyDeviceToHost, ctx->stream));
    if (norms) HIVE_CHECK_HIP(ctx, hipMemcpyAsync(norms, v->d_norms, nv * 12, hipMemcpyDeviceToHost, ctx->stream));
    if (colors) HIVE_CHECK_HIP(ctx, hipMemcpyAsync(colors, v->d_vcolors, nv * 3, hipMemcpyDeviceToHost, ctx->stream));
    if (faces && nf) HIVE_CHECK_HIP(ctx, hipMemcpyAsync(faces, v->d_faces, nf * 12, hipMemcpyDeviceToHost, ctx->stream));
    HIVE_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HIVE_OK;
}

int hive_tsdf_copy_mesh_voxel_coords(hive_tsdf *v, float *verts_vox) {
    HIVE_ENTER(v ? v->ctx : nullptr);
    if (!v) return hive_fail(nullptr, HIVE_ERR_INVALID, "vol is NULL");
    hive_ctx *ctx = v->ctx;
    if (v->n_verts < 0) return hive_fail(ctx, HIVE_ERR_STATE, "copy_mesh_voxel_coords: call hive_tsdf_extract_mesh first");
    HIVE_REQUIRE(ctx, verts_vox, "NULL argument");
    HIVE_CHECK_HIP(ctx, hipMemcpyAsync(verts_vox, v->d_verts_vox, (size_t)v->n_verts * 12, hipMemcpyDeviceToHost, ctx->stream));
    HIVE_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HIVE_OK;
}

}  // extern "C"
