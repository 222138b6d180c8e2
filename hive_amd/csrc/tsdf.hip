// TSDF volume: create / integrate / accumulate / finalize.  gfx950 only.
//
// Replaces `fusion.TSDFVolume.{__init__,integrate,get_volume}` of the reference's (absent)
// third_party/tsdf_fusion_python, call sites /root/reference/hive/fusion.py:104,124.
//
// Arithmetic contract (bit-exact against oracle/hive_oracle.c, build with -ffp-contract=off):
//   pt = origin + idx*voxel ; t = pt - T ; cam = (R0*tx + R1*ty) + R2*tz ; cam_z > 0 ;
//   pix = round(f*(cam/cam_z) + c) ; 0 <= pix < size ; depth != 0 ; depth - cam_z >= -trunc ;
//   dist = min(1, diff/trunc) ; w' = w + ow ; tsdf' = (tsdf*w + ow*dist)/w' ;
//   c' = min(255, round((c*w + ow*c_new)/w')) per channel.
//
// Kernel shape: voxel sweep over the part of the volume a frame can touch.  The volume is [X][Y][Z] with
// z fastest.  prep_frame fuses depth + colour into 8-byte texels and reduces max(depth) per 32 x 32-pixel tile; build_worklist clips
// every (x,y) row analytically against the view frustum and the depth tiles its image segment crosses (the camera-space position is
// affine in z; one LANE per row) and emits one item per 64-voxel segment of the surviving z interval; the fused sweep's list is then
// sorted by image band (its eighths go to the eight XCDs: texel gathers stay in one L2); integrate is a grid-stride
// sweep over that list: a wave takes 4 segments per trip (16 lanes x 4 consecutive z voxels = 16 bytes per
// lane and volume, for rows of ANY length: the accesses need dword alignment only), packed-f32 arithmetic.  The clip is
// padded and every voxel of a segment still runs the exact tests above, so results do not depend on it.
#include "hive_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

struct FrameParams {
    float R[9];  // cam_pose[r][c] for r,c < 3, row-major (camera-to-world rotation)
    float T[3];  // cam_pose[r][3]
    float fx, fy, cx, cy;
    float ox, oy, oz, vs, trunc, obs_w;
    int X, Y, Z, H, W;
    int x_off;                      // x-slab volumes: voxel (x, y, z) of this volume is voxel (x + x_off, y, z) of the scene grid
    const uint2 *frame;             // {depth bits, r | g<<8 | b<<16}
    const unsigned *tile_max;       // float bits of max(depth) per (32 << tile_shift)^2-pixel tile, [tiles_y][tiles_x] (prep_frame_kernel)
    int tile_shift, tiles_x, tiles_y;
    int row_far;                    // a row's far cut comes from the tiles its image segment crosses: 0 never (the frame's max depth), 1 where the tile table says it pays, 2 always
    int fast_colour;                // 1: obs_w == 1, roundf contract, every weight of the volume an integer < 65534: update_voxels FASTC
    unsigned long long *n_updated;
};

// ---------------------------------------------------------------------------------------------
// pre-pass, one workgroup per (32 << tile_shift)^2-pixel tile of one frame (blockIdx.y = frame of a batch: frames are n pixels apart in
// every buffer): depth f32 + colour u8x3 fused into one 8-byte texel, and max(depth) of the tile (the work list's per-row far cut,
// below).  VEC: 4 pixels per lane (one 16-byte depth load, three 4-byte colour loads, two 16-byte stores -- needs W % 4 == 0, depth
// 16-byte and colour 4-byte aligned); VEC = false is the scalar form for ragged / unaligned images.
template <bool VEC>
__global__ __launch_bounds__(256) void prep_frame_kernel(const float *__restrict__ depth, const uint8_t *__restrict__ color, int H, int W, int tile_shift,
                                                         int tiles_x, uint2 *__restrict__ out, unsigned *__restrict__ tile_max, int tile_stride,
                                                         unsigned *zero_next, int zero_words) {
    __shared__ unsigned wave_max[4];
    const int n = H * W;
    depth += (size_t)blockIdx.y * n;
    color += (size_t)blockIdx.y * n * 3;
    out += (size_t)blockIdx.y * n;
    tile_max += (size_t)blockIdx.y * tile_stride;
    // the NEXT sweep's scalar block (work-list length, ...) is cleared here instead of by a memset launch per sweep: the blocks
    // alternate, and this kernel runs after every kernel of the sweep that used that block last (stream order)
    if (blockIdx.x == 0 && blockIdx.y == 0 && (int)threadIdx.x < zero_words) zero_next[threadIdx.x] = 0u;
    const int TS = 32 << tile_shift;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
    // max over non-negative floats == max over their bit patterns; NaN / negatives are ignored (treated as 0)
    unsigned bits = 0;
    const int per_row = VEC ? TS / 4 : TS;
    for (int g = threadIdx.x; g < per_row * TS; g += 256) {
        const int r = ty * TS + g / per_row, c = tx * TS + (g % per_row) * (VEC ? 4 : 1);
        if (r >= H || c >= W) continue;
        const int i = r * W + c;
        if (VEC) {
            const float4 d = *reinterpret_cast<const float4 *>(depth + i);
            const uint3 cc = *reinterpret_cast<const uint3 *>(color + 3 * (size_t)i);  // 12 bytes = 4 RGB pixels
            const unsigned p0 = cc.x & 0xffffffu;
            const unsigned p1 = (cc.x >> 24) | ((cc.y & 0xffffu) << 8);
            const unsigned p2 = (cc.y >> 16) | ((cc.z & 0xffu) << 16);
            const unsigned p3 = cc.z >> 8;
            uint4 *o = reinterpret_cast<uint4 *>(out + i);
            o[0] = make_uint4(__float_as_uint(d.x), p0, __float_as_uint(d.y), p1);
            o[1] = make_uint4(__float_as_uint(d.z), p2, __float_as_uint(d.w), p3);
            const float m = fmaxf(fmaxf(d.x, d.y), fmaxf(d.z, d.w));
            bits = max(bits, (m > 0.f) ? __float_as_uint(m) : 0u);
        } else {
            const float d = depth[i];
            const unsigned rr = color[3 * (size_t)i + 0], gg = color[3 * (size_t)i + 1], bb = color[3 * (size_t)i + 2];
            out[i] = make_uint2(__float_as_uint(d), rr | (gg << 8) | (bb << 16));
            bits = max(bits, (d > 0.f) ? __float_as_uint(d) : 0u);
        }
    }
    for (int off = 32; off > 0; off >>= 1) bits = max(bits, (unsigned)__shfl_xor((int)bits, off));
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = bits;
    __syncthreads();
    if (threadIdx.x == 0) tile_max[blockIdx.x] = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
}

// ---------------------------------------------------------------------------------------------
// Work list.  One lane per (x,y) row clips the row analytically against the padded view frustum (the
// camera-space position is affine in z) and emits one item per SEG_LANES*VPT-voxel segment of the
// surviving z interval.  Only segments on the list are ever touched by the integrate kernel; every voxel
// of a segment still runs the exact inclusion tests, so the (conservative) clip cannot change a result.
// A wave sweeps 64 / SEG_LANES segments at once (one per 16-lane group): the clipped intervals are short
// (mean 120-190 voxels at 512^3), so with one 256-voxel item per wave 59 % of the lanes were padding --
// 16-lane segments (64 voxels, 256 contiguous bytes per volume) cut the wave count by 40 %.
#ifndef HIVE_SEG_LANES
#define HIVE_SEG_LANES 16
#endif
#ifndef HIVE_GRID_MULT
#define HIVE_GRID_MULT 16
#endif
constexpr int SEG_LANES = HIVE_SEG_LANES;
constexpr int VPT = 4;  // consecutive z voxels per lane: one 16-byte access per lane and volume plane
constexpr int COUNT_SLOTS = 64, COUNT_STRIDE = 16;  // update counters of the COUNT kernels: 64 x u64, 128 bytes apart (hive_ctx::d_scalars + 128)
struct WorkItem {
    unsigned xy;  // x | y << 16
    unsigned zz;  // segment start z | (voxels of the segment inside the row's interval - 1) << 16 | frames whose own interval meets the segment << 22 | image band << 26
};
constexpr int ITEM_NLIVE_SHIFT = 16, ITEM_MASK_SHIFT = 22, ITEM_BIN_SHIFT = 26;
constexpr int NBINS = 64;  // image bands (of the sweep's first frame) the fused sweep's work list is sorted by
static_assert(HIVE_SEG_LANES * 4 <= 64, "a segment's live-voxel count - 1 takes 6 bits of WorkItem::zz");
__device__ __forceinline__ int item_nlive(unsigned zz) { return ((zz >> ITEM_MASK_SHIFT) & 15u) ? (int)((zz >> ITEM_NLIVE_SHIFT) & 63u) + 1 : 0; }  // 0: a dead (all-zero) item

struct RowClip {
    int z0, z1;  // half-open voxel range that may pass the inclusion tests
};

// Per-tile depth maxima of a frame (prep_frame_kernel) -> LDS table of their 3 x 3-dilated values, and the frame's maximum.  A row of
// voxels projects to a straight image segment; every pixel a voxel of the row can round to lies within one tile (in each axis) of a
// sample taken every <= tile size along that segment, so the maximum of the DILATED table over the samples bounds the depth the row
// can meet: voxels with cam_z > that + trunc fail `depth - cam_z >= -trunc` at every pixel they can reach.
constexpr int MAX_TILES = 2048;
// (every thread takes (tile, neighbour) pairs: one independent load each, folded into the table with LDS atomics -- a thread per tile doing its
// nine loads in turn, frame after frame, kept the whole workgroup waiting for 36 round trips before the first row was clipped)
__device__ __forceinline__ void load_dilated_tiles(const unsigned *__restrict__ raw, int tiles_x, int tiles_y, float *dil, unsigned *gmax, float *gsum, int tid, int nthreads) {
    const int tiles = tiles_x * tiles_y;
    unsigned *bits = reinterpret_cast<unsigned *>(dil);  // maxima of non-negative floats == maxima of their bit patterns
    for (int t = tid; t < tiles; t += nthreads) bits[t] = 0u;
    __syncthreads();
    for (int i = tid; i < tiles * 9; i += nthreads) {
        const int t = i / 9, k = i - t * 9;
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int yy = min(max(ty + k / 3 - 1, 0), tiles_y - 1), xx = min(max(tx + k % 3 - 1, 0), tiles_x - 1);
        const unsigned v = raw[yy * tiles_x + xx];
        if (v) atomicMax(&bits[t], v);
    }
    __syncthreads();
    unsigned m_all = 0;
    float sum = 0.f;
    const int reach = max(1, min(tiles_x, tiles_y) / 5);  // a row's image segment crosses many tiles: what it can gain is what a WIDE neighbourhood's maximum leaves
    for (int t = tid; t < tiles; t += nthreads) {
        m_all = max(m_all, bits[t]);
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        unsigned w = 0u;
        for (int yy = max(ty - reach, 0); yy <= min(ty + reach, tiles_y - 1); ++yy)
            for (int xx = max(tx - reach, 0); xx <= min(tx + reach, tiles_x - 1); ++xx) w = max(w, bits[yy * tiles_x + xx]);
        sum += __uint_as_float(w);
    }
    for (int off = 32; off > 0; off >>= 1) {
        m_all = max(m_all, (unsigned)__shfl_xor((int)m_all, off));
        sum += __shfl_xor(sum, off);
    }
    if ((tid & 63) == 0 && m_all) {
        atomicMax(gmax, m_all);
        atomicAdd(gsum, sum);
    }
}
// The per-row far cut pays where the depth map has large regions much nearer than its deepest pixel (masked foreground / background depth, a near
// object filling part of the view); walking a row's image segment over the tile table costs ~9 us per four-frame sweep at 512^3 (a third of the
// work-list kernel).  Its best case removes about (1 - mean(w) / max) of the rows' depth range, w = the table's maximum over a neighbourhood a fifth of
// the image wide (a row's segment crosses that many tiles), so a frame with mean(w) above 70 % of its maximum keeps the frame's bound for all rows:
// on the two bench scenes (probe_sweep_ab.py) the cut removed 0.7 % (room) and 0.01 % (DPT depth of the seeded weights: noise-like, every region holds a
// pixel near the maximum) of the work list and cost 2-4 us per frame.  HIVE_TSDF_ROW_FAR=0 / 2 force it off / on.
__device__ __forceinline__ bool row_far_pays(float gsum, unsigned gmax_bits, int tiles) { return gsum < 0.70f * (float)tiles * __uint_as_float(gmax_bits); }

// Clip the row cam(tz) = a + b*tz against the padded frustum.  Conservative: pixel bounds widened by
// half a pixel, the result by 2 voxels on each side.  far_frame: max(depth) of the frame; dil (or null): the dilated tile table.
__device__ __forceinline__ RowClip clip_row(const FrameParams &p, float ax, float ay, float az, float far_frame, const float *dil) {
    const float bx = p.R[6], by = p.R[7], bz = p.R[8];
    // tz range of the row: tz(z) = (oz + z*vs) - T2
    float lo = (p.oz - p.T[2]) - p.vs;
    float hi = (p.oz + (float)p.Z * p.vs - p.T[2]) + p.vs;
    bool empty = false;
    auto cut = [&](float alpha, float beta) {  // keep alpha + beta*tz >= 0
        if (beta > 0.f) {
            lo = fmaxf(lo, -alpha / beta);
        } else if (beta < 0.f) {
            hi = fminf(hi, -alpha / beta);
        } else if (alpha < 0.f) {
            empty = true;
        }
    };
    // The two depth cuts carry a slack of ~1e-5 of the distances involved: the kernel evaluates cam_z = az + bz tz and depth - cam_z in
    // float32, where a row (nearly) PARALLEL to the image plane (|bz| ~ 1e-17 at a yaw of 90 degrees) absorbs bz tz entirely, while the
    // cut position -alpha / beta amplifies any rounding of alpha by 1 / |bz|: with the far wall exactly at max depth, alpha = 0 cut
    // such a row at tz <= 0 although every voxel of it sits ON the boundary and updates (found by the frame masks of round 4: the
    // union of a sweep's clips used to hide it).  The slack moves a cut by less than 1e-3 voxel wherever |bz| is not tiny.
    const float slack = 1.0e-5f * (1.0f + fabsf(az) + far_frame);
    cut(az + slack, bz);                                                               // cam_z >= 0
    cut((far_frame + p.trunc - az) + slack, -bz);                                      // cam_z <= max depth of the frame + trunc
    cut(p.fx * ax + (p.cx + 1.0f) * az, p.fx * bx + (p.cx + 1.0f) * bz);               // px >= -1
    cut(((float)p.W - p.cx) * az - p.fx * ax, ((float)p.W - p.cx) * bz - p.fx * bx);   // px <= W
    cut(p.fy * ay + (p.cy + 1.0f) * az, p.fy * by + (p.cy + 1.0f) * bz);               // py >= -1
    cut(((float)p.H - p.cy) * az - p.fy * ay, ((float)p.H - p.cy) * bz - p.fy * by);   // py <= H
    RowClip r;
    r.z0 = r.z1 = 0;
    if (empty || !(lo <= hi)) return r;
    if (dil) {
        // the row's own far cut: the largest depth in the tiles its image segment crosses.  The segment's end points are the
        // projections of the clipped interval's ends; an end closer than 5 cm to the camera plane projects with too little
        // precision for this (such rows keep the frame's bound -- a handful of rows next to the camera).
        const float z0 = az + bz * lo, z1 = az + bz * hi;
        if (fminf(z0, z1) >= 0.05f) {
            const float wmax = (float)(p.W - 1), hmax = (float)(p.H - 1);
            const float u0 = fminf(fmaxf(p.fx * ((ax + bx * lo) / z0) + p.cx, 0.f), wmax), v0 = fminf(fmaxf(p.fy * ((ay + by * lo) / z0) + p.cy, 0.f), hmax);
            const float u1 = fminf(fmaxf(p.fx * ((ax + bx * hi) / z1) + p.cx, 0.f), wmax), v1 = fminf(fmaxf(p.fy * ((ay + by * hi) / z1) + p.cy, 0.f), hmax);
            const float ts = (float)(32 << p.tile_shift);
            const int n = (int)(fmaxf(fabsf(u1 - u0), fabsf(v1 - v0)) / ts) + 1;  // samples 0 .. n, at most a tile apart in each axis
            const float du = (u1 - u0) / (float)n, dv = (v1 - v0) / (float)n;
            float m = 0.f;
            for (int i = 0; i <= n; ++i) {
                const int tx = min((int)(u0 + du * (float)i) >> (5 + p.tile_shift), p.tiles_x - 1);
                const int ty = min((int)(v0 + dv * (float)i) >> (5 + p.tile_shift), p.tiles_y - 1);
                m = fmaxf(m, dil[ty * p.tiles_x + tx]);
            }
            cut((m + p.trunc - az) + slack, -bz);  // cam_z <= row bound + trunc
            if (empty || !(lo <= hi)) return r;
        }
    }
    const float inv = 1.0f / p.vs;
    float zf0 = floorf((lo + p.T[2] - p.oz) * inv) - 2.0f;
    float zf1 = ceilf((hi + p.T[2] - p.oz) * inv) + 3.0f;
    zf0 = fminf(fmaxf(zf0, 0.f), (float)p.Z);
    zf1 = fminf(fmaxf(zf1, 0.f), (float)p.Z);
    r.z0 = (int)zf0;
    r.z1 = (int)zf1;
    return r;
}

// block-wide exclusive scan of the rows' segment counts (1024 threads), one atomic per workgroup; returns the lane's first slot
__device__ __forceinline__ unsigned worklist_slots(unsigned n_chunks, unsigned *n_items, unsigned *wave_sum, unsigned *block_base) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = n_chunks;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = (unsigned)__shfl_up((int)inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wave_sum[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned total = 0;
        for (int w = 0; w < 16; ++w) {
            const unsigned s = wave_sum[w];
            wave_sum[w] = total;
            total += s;
        }
        *block_base = total ? atomicAdd(n_items, total) : 0u;
    }
    __syncthreads();
    return *block_base + wave_sum[wave] + inc - n_chunks;
}

__global__ __launch_bounds__(1024) void build_worklist_kernel(FrameParams p, WorkItem *__restrict__ items, unsigned *n_items) {
    __shared__ unsigned wave_sum[16];
    __shared__ unsigned block_base;
    __shared__ float dil[MAX_TILES];
    __shared__ unsigned gmax;
    __shared__ float gsum;
    constexpr int CHUNK = SEG_LANES * VPT;
    if (threadIdx.x == 0) gmax = 0u, gsum = 0.f;
    __syncthreads();
    load_dilated_tiles(p.tile_max, p.tiles_x, p.tiles_y, dil, &gmax, &gsum, threadIdx.x, 1024);
    __syncthreads();
    const bool row_far = p.row_far == 2 || (p.row_far == 1 && row_far_pays(gsum, gmax, p.tiles_x * p.tiles_y));
    const long long row = (long long)blockIdx.x * 1024 + threadIdx.x;
    unsigned n_chunks = 0;
    int zstart = 0, z1 = 0, x = 0, y = 0;
    if (row < (long long)p.X * p.Y) {
        x = (int)(row / p.Y);
        y = (int)(row % p.Y);
        const float tx = (p.ox + (float)(x + p.x_off) * p.vs) - p.T[0];
        const float ty = (p.oy + (float)y * p.vs) - p.T[1];
        const float ax = p.R[0] * tx + p.R[3] * ty;
        const float ay = p.R[1] * tx + p.R[4] * ty;
        const float az = p.R[2] * tx + p.R[5] * ty;
        const RowClip clip = clip_row(p, ax, ay, az, __uint_as_float(gmax), row_far ? dil : nullptr);
        if (clip.z1 > clip.z0) {
            zstart = (clip.z0 / VPT) * VPT;
            z1 = clip.z1;
            n_chunks = (unsigned)((z1 - zstart + CHUNK - 1) / CHUNK);
        }
    }
    const unsigned slot = worklist_slots(n_chunks, n_items, wave_sum, &block_base);
    for (unsigned c = 0; c < n_chunks; ++c) {
        const int zs = zstart + (int)c * CHUNK;
        WorkItem it;
        it.xy = (unsigned)x | ((unsigned)y << 16);
        it.zz = (unsigned)zs | ((unsigned)(min(z1 - zs, CHUNK) - 1) << ITEM_NLIVE_SHIFT) | (1u << ITEM_MASK_SHIFT);
        items[slot + c] = it;
    }
}

// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Per-voxel arithmetic, written once for V = float and V = f2 (two consecutive z voxels in one 64-bit
// register pair).  On gfx950 a wave64 f32 VALU instruction occupies its SIMD for 4 cycles and the packed
// forms (v_pk_mul/add/fma_f32) produce two results in the same 4 -- the kernel was VALU-issue bound (PMC:
// 4.08 SIMD cycles per VALU instruction, 53 % of the kernel), so every mul/add/fma below is 2-wide.
// Element-wise, the operation order is exactly the contract's: packed instructions are IEEE per element.
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float v_splat(float x, float) { return x; }
__device__ __forceinline__ f2 v_splat(float x, f2) { return f2{x, x}; }
__device__ __forceinline__ float v_rcp(float d) { return __builtin_amdgcn_rcpf(d); }
__device__ __forceinline__ f2 v_rcp(f2 d) { return f2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)}; }
__device__ __forceinline__ float v_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ f2 v_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float v_min(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ f2 v_min(f2 a, f2 b) { return f2{fminf(a.x, b.x), fminf(a.y, b.y)}; }
template <int RM>
__device__ __forceinline__ float v_round(float x) { return hive_round<RM>(x); }
template <int RM>
__device__ __forceinline__ f2 v_round(f2 x) { return f2{hive_round<RM>(x.x), hive_round<RM>(x.y)}; }
// roundf of a NON-NEGATIVE value (the colour quotients): trunc(x + pred(0.5)) == roundf(x) for every float x >= 0 (exhaustively checked,
// DESIGN 5.1) -- a packed add and a v_trunc instead of roundf's six instructions; RM = 0 is one v_rndne either way
constexpr float HALF_PRED = 0.49999997f;  // the float below 0.5
template <int RM>
__device__ __forceinline__ float v_round_nonneg(float x) { return RM == 1 ? truncf(x + HALF_PRED) : rintf(x); }
template <int RM>
__device__ __forceinline__ f2 v_round_nonneg(f2 x) {
    if (RM != 1) return f2{rintf(x.x), rintf(x.y)};
    const f2 y = x + f2{HALF_PRED, HALF_PRED};
    return f2{truncf(y.x), truncf(y.y)};
}
__device__ __forceinline__ float v_get(float v, int) { return v; }
__device__ __forceinline__ float v_get(f2 v, int i) { return i ? v.y : v.x; }
__device__ __forceinline__ void v_set(float &v, int, float x) { v = x; }
__device__ __forceinline__ void v_set(f2 &v, int i, float x) {
    if (i) v.y = x; else v.x = x;
}

// IEEE-exact float division with a shared, refined reciprocal of the denominator.  This is the
// sequence hipcc itself emits for a correctly rounded a/d (rcp, one Newton step on the reciprocal,
// two on the quotient) minus the range scaling / special-case fix-up, which cannot trigger for the
// operands used here (|d| and |a/d| within 2^+-60); sharing y between quotients of one denominator
// removes a third of the instructions of the update.  Checked bit-for-bit against the CPU oracle.
template <typename V>
__device__ __forceinline__ V refined_rcp(V d) {
    const V r0 = v_rcp(d);
    const V e = v_fma(-d, r0, v_splat(1.0f, d));
    return v_fma(e, r0, r0);
}

template <typename V>
__device__ __forceinline__ V div_exact(V a, V d, V y) {
    const V q0 = a * y;
    const V e0 = v_fma(-d, q0, a);
    const V q1 = v_fma(e0, y, q0);
    const V e1 = v_fma(-d, q1, a);
    return v_fma(e1, y, q1);
}

// Geometry half of the per-voxel work (no memory access) for NV consecutive z voxels starting at z:
// camera depth and the pixel each voxel centre rounds to, or -1 (behind the camera / outside the image).
// (row constants per ELEMENT: the elements of a packed pair may belong to different rows -- the fused sweep's gather role)
template <int RM, typename V, int NV>
__device__ __forceinline__ void voxel_pixels_rows(const FrameParams &p, V ax, V ay, V az, V zf, V &cam_z, int (&pix)[NV]) {
    const V pt_z = v_splat(p.oz, zf) + zf * v_splat(p.vs, zf);
    const V tz = pt_z - v_splat(p.T[2], zf);
    const V cam_x = ax + v_splat(p.R[6], zf) * tz;
    const V cam_y = ay + v_splat(p.R[7], zf) * tz;
    cam_z = az + v_splat(p.R[8], zf) * tz;
    bool tiny = false;
#pragma unroll
    for (int i = 0; i < NV; ++i) tiny = tiny || fabsf(v_get(cam_z, i)) < 1.0e-18f;  // (one comparison; zero / negative depths take the full divisions too: same results)
    V qx, qy;
    if (__builtin_expect(__any(tiny), 0)) {  // outside div_exact's domain (never in practice): full divisions, wave-uniform branch
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v_set(qx, i, v_get(cam_x, i) / v_get(cam_z, i));
            v_set(qy, i, v_get(cam_y, i) / v_get(cam_z, i));
        }
    } else {
        const V zr = refined_rcp(cam_z);
        qx = div_exact(cam_x, cam_z, zr);
        qy = div_exact(cam_y, cam_z, zr);
    }
    const V ux = v_splat(p.fx, zf) * qx + v_splat(p.cx, zf), uy = v_splat(p.fy, zf) * qy + v_splat(p.cy, zf);
    if (RM == 1) {
        // roundf(u) and the two bounds tests of each coordinate in 3.5 instructions instead of 9 (roundf alone expands to six):
        // f = floor(u + pred(0.5)) equals roundf(u) for every u >= 0, is +0 for -0.5 < u < 0 (roundf: -0, which the contract's `>= 0` accepts
        // as pixel 0) and negative for u <= -0.5 -- checked over every float in [-4, 2^25) -- so "0 <= roundf(u) < W" is
        // one unsigned comparison of f's bit pattern with (float)W's: negative f have the sign bit set, non-negative floats order like
        // their bits, and f is never -0.
        const V fx_ = ux + v_splat(HALF_PRED, zf), fy_ = uy + v_splat(HALF_PRED, zf);
        const unsigned wbits = __float_as_uint((float)p.W), hbits = __float_as_uint((float)p.H);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float rx = floorf(v_get(fx_, i)), ry = floorf(v_get(fy_, i));
            const bool ok = (v_get(cam_z, i) > 0.0f) & (__float_as_uint(rx) < wbits) & (__float_as_uint(ry) < hbits);
            pix[i] = ok ? (__mul24((int)ry, p.W) + (int)rx) : -1;  // H, W < 2^23 (checked on the host)
        }
    } else {
        const V px = v_round<RM>(ux), py = v_round<RM>(uy);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float fx_ = v_get(px, i), fy_ = v_get(py, i);
            const bool ok = v_get(cam_z, i) > 0.0f && (fx_ >= 0.0f) && (fx_ < (float)p.W) && (fy_ >= 0.0f) && (fy_ < (float)p.H);
            pix[i] = ok ? (__mul24((int)fy_, p.W) + (int)fx_) : -1;
        }
    }
}

template <int RM, typename V, int NV, int ZSTEP = 1>
__device__ __forceinline__ void voxel_pixels(const FrameParams &p, float ax, float ay, float az, int z, V &cam_z, int (&pix)[NV]) {
    V zf;
#pragma unroll
    for (int i = 0; i < NV; ++i) v_set(zf, i, (float)(z + i * ZSTEP));
    voxel_pixels_rows<RM, V, NV>(p, v_splat(ax, zf), v_splat(ay, zf), v_splat(az, zf), zf, cam_z, pix);
}

// Camera depth of NV consecutive z voxels: the cam_z expressions of voxel_pixels, verbatim (bit-identical results).
template <typename V, int NV>
__device__ __forceinline__ V voxel_cam_z(const FrameParams &p, float az, int z) {
    V zf;
#pragma unroll
    for (int i = 0; i < NV; ++i) v_set(zf, i, (float)(z + i));
    const V pt_z = v_splat(p.oz, zf) + zf * v_splat(p.vs, zf);
    const V tz = pt_z - v_splat(p.T[2], zf);
    return v_splat(az, zf) + v_splat(p.R[8], zf) * tz;
}

// Running-average update of NV voxels; lanes / elements with ok == false keep their values.
// UNIT: obs_weight == 1 (the reference's only call): ow * x == x exactly, the four multiplications are left out.
// The colour plane holds b 65536 + g 256 + r as a float (exact: an integer below 2^24).  The kernels keep the three channels as
// floats while a voxel is in registers (the fused sweep: across all of its frames) and pack once before the store.
template <typename V, int NV>
__device__ __forceinline__ void unpack_colour(V c, V &r, V &g, V &b) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const unsigned oc = (unsigned)v_get(c, i);
        v_set(r, i, (float)(oc & 255u));
        v_set(g, i, (float)((oc >> 8) & 255u));
        v_set(b, i, (float)(oc >> 16));
    }
}
template <typename V>
__device__ __forceinline__ V pack_colour(V r, V g, V b) {  // (the oracle's own expression; two packed fma)
    return v_fma(b, v_splat(65536.0f, r), v_fma(g, v_splat(256.0f, r), r));
}
__device__ __forceinline__ float v_floor(float x) { return floorf(x); }
__device__ __forceinline__ f2 v_floor(f2 x) { return f2{floorf(x.x), floorf(x.y)}; }

// Running-average update of NV voxels; lanes / elements with ok == false keep their values.
// UNIT: obs_weight == 1 (the reference's only call): ow * x == x exactly, the four multiplications are left out.
// FASTC (needs RM == 1, UNIT, and every weight of the volume an integer below 65534 -- the host tracks that: hive_tsdf::unit_weights):
// the contract's colour update c' = min(255, roundf((c w + c_new) / (w + 1))) without its three divisions.  With n = c w + c_new and
// d = w + 1 integers (both exact in float32 below 2^24), n / d = c + delta / d, delta = c_new - c in [-255, 255].  The correctly
// rounded quotient fl(n / d) lies on the same side of every k + 1/2 as n / d itself -- a quotient that is not exactly k + 1/2 is at
// least 1 / (2 d) > 2^-17 away from it, more than half an ulp of any float below 256 -- so roundf(fl(n / d)) = floor(n / d + 1/2)
// = c + floor(delta / d + 1/2), and that is <= 255 by itself.  floor(delta / d + 1/2) is evaluated as floor(fma(delta, y, 1/2 + y / 4))
// with y = the refined reciprocal of d: the bias y / 4 lifts exact ties (delta / d + 1/2 an integer) clear of the rounding errors
// (<= 4.6e-5 / d + 7.5e-8, against 0.25 / d) and leaves every other value (at least 1 / (2 d) from an integer) on its side.
// Exhaustively checked against the contract's expression on the CPU (tests/test_oracle_cpu.py::test_fast_colour_update_identity).
template <int RM, typename V, int NV, bool UNIT, bool FASTC>
__device__ __forceinline__ void update_voxels(V &t, V &w, V &r, V &g, V &b, V dist, const unsigned (&rgb)[NV], const bool (&ok)[NV], float ow) {
    static_assert(!FASTC || (RM == 1 && UNIT), "the division-free colour update is the roundf / unit-weight contract's");
    const V w_old = w;
    V nr, ng, nb;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v_set(nr, i, (float)(rgb[i] & 255u));
        v_set(ng, i, (float)((rgb[i] >> 8) & 255u));
        v_set(nb, i, (float)((rgb[i] >> 16) & 255u));
    }
    if (FASTC) {
        const V w_new = w_old + v_splat(1.0f, w);
        const V wr = refined_rcp(w_new);
        const V t_new = div_exact(t * w_old + dist, w_new, wr);
        V y, inc;  // elements that do not update: y = 0 -> floor(0 + 1/2) = 0 leaves the channels alone; weight + 0
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v_set(y, i, ok[i] ? v_get(wr, i) : 0.0f);
            v_set(inc, i, ok[i] ? 1.0f : 0.0f);
        }
        const V bias = v_fma(v_splat(0.25f, w), y, v_splat(0.5f, w));
        r = r + v_floor(v_fma(nr - r, y, bias));
        g = g + v_floor(v_fma(ng - g, y, bias));
        b = b + v_floor(v_fma(nb - b, y, bias));
        w = w_old + inc;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (ok[i]) v_set(t, i, v_get(t_new, i));
        return;
    }
    const V vow = v_splat(ow, w);
    const V w_new = w_old + vow;
    const V wr = refined_rcp(w_new);
    const V t_new = div_exact(t * w_old + (UNIT ? dist : vow * dist), w_new, wr);
    const V lim = v_splat(255.0f, w);
    const V r_new = v_min(v_round_nonneg<RM>(div_exact(r * w_old + (UNIT ? nr : vow * nr), w_new, wr)), lim);
    const V g_new = v_min(v_round_nonneg<RM>(div_exact(g * w_old + (UNIT ? ng : vow * ng), w_new, wr)), lim);
    const V b_new = v_min(v_round_nonneg<RM>(div_exact(b * w_old + (UNIT ? nb : vow * nb), w_new, wr)), lim);
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (ok[i]) {
            v_set(t, i, v_get(t_new, i));
            v_set(w, i, v_get(w_new, i));
            v_set(r, i, v_get(r_new, i));
            v_set(g, i, v_get(g_new, i));
            v_set(b, i, v_get(b_new, i));
        }
}

// The depth tests of the volume role for a lane's 4 voxels (2 packed pairs) and the truncated distance.
// (Round 3, measured and taken out: skipping the distance division where `diff >= trunc` holds for every voxel of the WAVE -- min(1, .) = 1
// exactly -- and the tsdf division where tsdf and dist are both 1.  A wave spans four 64-voxel segments of different rows; 13 % of the
// waves lie entirely outside the 5-voxel truncation band, so the tests cost more instructions than the skips save: SQ_INSTS_VALU +4 %.)
template <typename V>
__device__ __forceinline__ bool depth_tests(const FrameParams &p, float trunc_rcp, const float (&depth_v)[4], const V (&cam_z)[2], bool live, int nrow,
                                            V (&dist)[2], bool (&ok)[4]) {
    bool any = false;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        V depth;
#pragma unroll
        for (int i = 0; i < 2; ++i) v_set(depth, i, depth_v[2 * g + i]);
        const V diff = depth - cam_z[g];
        dist[g] = v_min(v_splat(1.0f, diff), div_exact(diff, v_splat(p.trunc, diff), v_splat(trunc_rcp, diff)));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = 2 * g + i;
            ok[j] = live && j < nrow && (v_get(depth, i) != 0.0f) && !(v_get(diff, i) < -p.trunc);
            any = any || ok[j];
        }
    }
    return any;
}

// Volume accesses: one 16-byte access per lane and volume.  Default cache policy -- the non-temporal forms
// measured 6 % faster on a frame that updates every voxel and 3 % slower on the room scene (A/B in one
// process group, tools/ab_integrate.py), so they are not used.
// A row starts at ((x Y) + y) Z floats: 16-byte aligned only when Z % 4 == 0.  global_load / global_store_dwordx4 need dword alignment
// only (gfx950 runs in unaligned-access mode), so the volume side of the access is typed with 4-byte alignment and rows of ANY length take
// the vector path; `n` < 4 is the last quad of a row with Z % 4 != 0: its tail lies in the NEXT row and is neither read nor written.
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void vol_load4(float *dst, const float *src, int n) {
    if (n >= 4) {
        *reinterpret_cast<f4 *>(dst) = *reinterpret_cast<const f4u *>(src);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j] = j < n ? src[j] : 0.0f;
    }
}
__device__ __forceinline__ void vol_store4(float *dst, const float *src, int n) {
    if (n >= 4) {
        *reinterpret_cast<f4u *>(dst) = *reinterpret_cast<const f4 *>(src);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < n) dst[j] = src[j];
    }
}

// Shape of one lane's share of a work item: VPT = 4 consecutive z voxels of one segment, as NG = 2 packed pairs.
struct ItemShape {
    static constexpr int NV = 2;
    static constexpr int NG = VPT / NV;
    typedef f2 V;
};

// Grid-stride sweep over the work list: a wave takes 64 / SEG_LANES consecutive segments per trip (16 lanes
// x 4 voxels each), 16-byte accesses per lane and volume.  Lanes past the end of the list or of their row's
// clipped interval are dead (no volume access); the voxels of a row's last quad that lie past the row (Z % 4 != 0)
// are excluded from the tests and from the access.
//
// What bounds it (512^3, room scene; experiments in DESIGN.md section 5): the CU's vector-memory pipeline, per
// INSTRUCTION.  A trip issues 4 texel gathers (8 B / lane) + 3 volume loads + 3 volume stores (16 B / lane).  A
// 16-byte streaming instruction costs the pipeline ~120 cycles at the HBM rate; a gather costs ~2.3 cycles per
// DISTINCT 64-byte line its 64 lanes touch (tools/ubench/gather.hip: 150 cycles for 64 lines, 31 for 12, the same for
// 4- and 8-byte texels), so what a gather instruction costs is decided by which voxels share it.
//   * volume role: lane j of a segment owns voxels 4j .. 4j+3 (one 16-byte access per volume).  Gathering in this
//     role puts voxels 4 apart (~12 pixels, 96 bytes) on neighbouring lanes: 64 lines per instruction.
//   * gather role: for the texel fetch the lanes of a segment take CONSECUTIVE voxels -- gather k covers
//     voxels 16k .. 16k+15 of each of the wave's 4 segments -- so neighbouring lanes hit neighbouring pixels (a z run
//     projects to a pixel run; ~3 lanes per line when it runs along the image rows).  The fetched {depth, rgb} pairs
//     go through a 2 KB per-wave LDS exchange (ds_write_b64 / 2 x ds_read_b128, no workgroup barrier: LDS operations
//     of one wave complete in order) into the volume role, which recomputes its own camera depths (two packed
//     instructions) and runs the update.  Pixels outside the image / behind the camera are sent as depth 0, which the
//     depth test rejects -- the same outcome as the contract's separate test.
// ACCUM = false: running-average update of (tsdf, weight, colour) -- the reference semantics.
// ACCUM = true : add into the 5 accumulator planes [num, w, r, g, b] (frame-sharded fusion).
template <int RM, bool COUNT, bool ACCUM>
__global__ __launch_bounds__(256) void integrate_kernel(FrameParams p, const WorkItem *__restrict__ items,
                                                        const unsigned *__restrict__ n_items_ptr, float *__restrict__ v0,
                                                        float *__restrict__ v1, float *__restrict__ v2, long long plane) {
    constexpr int PER_WAVE = 64 / SEG_LANES;
    typedef ItemShape Sh;
    typedef Sh::V V;
    constexpr int SEG_VOX = SEG_LANES * 4;             // voxels of a segment
    __shared__ uint2 xchg[4 * PER_WAVE * SEG_VOX];  // per wave: PER_WAVE segments x SEG_VOX voxels x {depth bits, rgb}
    const int lane = threadIdx.x & 63;
    const int seg = lane / SEG_LANES, sl = lane % SEG_LANES;
    const unsigned n_items = *n_items_ptr;
    const unsigned n_trips = (n_items + PER_WAVE - 1) / PER_WAVE;
    const unsigned stride = gridDim.x * 4;
    const float trunc_rcp = refined_rcp(p.trunc);
    unsigned n_upd = 0;
    for (unsigned trip = blockIdx.x * 4 + (threadIdx.x >> 6); trip < n_trips; trip += stride) {
        const unsigned ii = trip * PER_WAVE + (unsigned)seg;
        WorkItem item;
        item.xy = item.zz = 0u;  // interval end 0: dead lanes
        if (ii < n_items) item = items[ii];
        const int x = (int)(item.xy & 0xffffu), y = (int)(item.xy >> 16);
        const int zseg = (int)(item.zz & 0xffffu);
        const int zb = zseg + sl * VPT;  // volume role: this lane's first voxel
        const bool live = sl * VPT < item_nlive(item.zz);
        const int nrow = p.Z - zb;  // voxels of this quad inside the row (>= 4 except in the last quad of a row with Z % 4 != 0)
        const long long idx = ((long long)x * p.Y + y) * p.Z + zb;
        float t[VPT], w[VPT], c[VPT];  // ACCUM: planes 0..2
        float c3[VPT], c4[VPT];        // ACCUM: planes 3..4
        // row constants, in the contract's operation order
        const float tx = (p.ox + (float)(x + p.x_off) * p.vs) - p.T[0];
        const float ty = (p.oy + (float)y * p.vs) - p.T[1];
        const float ax = p.R[0] * tx + p.R[3] * ty;
        const float ay = p.R[1] * tx + p.R[4] * ty;
        const float az = p.R[2] * tx + p.R[5] * ty;
        V cam_z[Sh::NG];
        float depth_v[VPT];
        unsigned rgb[VPT];
        {
            // gather role: voxels zseg + SEG_LANES k + sl, k = 0 .. 3 (packed pairs k = {0, 1}, {2, 3})
            uint2 tex[4];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                V cz;
                int pix[Sh::NV];
                voxel_pixels<RM, V, Sh::NV, SEG_LANES>(p, ax, ay, az, zseg + sl + 2 * SEG_LANES * g, cz, pix);
#pragma unroll
                for (int i = 0; i < Sh::NV; ++i) {
                    uint2 tx2 = p.frame[max(pix[i], 0)];
                    if (pix[i] < 0) tx2.x = 0u;  // outside the image / behind the camera: rejected by the depth test
                    tex[g * Sh::NV + i] = tx2;
                }
            }
            uint2 *mine = xchg + ((threadIdx.x >> 6) * PER_WAVE + seg) * SEG_VOX;
#pragma unroll
            for (int k = 0; k < 4; ++k) mine[SEG_LANES * k + sl] = tex[k];
            // same wave writes and reads: LDS operations of a wave complete in issue order, no workgroup barrier
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint4 lo = *reinterpret_cast<const uint4 *>(mine + 4 * sl);
            const uint4 hi = *reinterpret_cast<const uint4 *>(mine + 4 * sl + 2);
            depth_v[0] = __uint_as_float(lo.x), rgb[0] = lo.y;
            depth_v[1] = __uint_as_float(lo.z), rgb[1] = lo.w;
            depth_v[2] = __uint_as_float(hi.x), rgb[2] = hi.y;
            depth_v[3] = __uint_as_float(hi.z), rgb[3] = hi.w;
            __builtin_amdgcn_wave_barrier();  // the next trip's writes stay behind these reads
#pragma unroll
            for (int g = 0; g < Sh::NG; ++g) cam_z[g] = voxel_cam_z<V, Sh::NV>(p, az, zb + g * Sh::NV);
        }
        // inclusion tests
        V dist[Sh::NG];
        bool ok[VPT];
        const bool any = depth_tests<V>(p, trunc_rcp, depth_v, cam_z, live, nrow, dist, ok);
        // (measured and rejected: issuing these loads for all live lanes BEFORE the geometry, beside the texel gathers, to take one
        // round trip out of the wave's dependency chain -- 95 vs 92 us: the bytes loaded for lanes that then fail the test cost more)
        if (any) {
            vol_load4(t, v0 + idx, nrow);
            if (!ACCUM) {
                vol_load4(w, v1 + idx, nrow);
                vol_load4(c, v2 + idx, nrow);
            } else {
                vol_load4(w, v0 + plane + idx, nrow);
                vol_load4(c, v0 + 2 * plane + idx, nrow);
                vol_load4(c3, v0 + 3 * plane + idx, nrow);
                vol_load4(c4, v0 + 4 * plane + idx, nrow);
            }
            if (!ACCUM) {
#pragma unroll
                for (int g = 0; g < Sh::NG; ++g) {
                    V tv, wv, cv, rv, gv, bv;
                    unsigned rg[Sh::NV];
                    bool okg[Sh::NV];
#pragma unroll
                    for (int i = 0; i < Sh::NV; ++i) {
                        const int j = g * Sh::NV + i;
                        v_set(tv, i, t[j]);
                        v_set(wv, i, w[j]);
                        v_set(cv, i, c[j]);
                        rg[i] = rgb[j];
                        okg[i] = ok[j];
                    }
                    unpack_colour<V, Sh::NV>(cv, rv, gv, bv);
                    // (kernel-uniform branches)
                    if (RM == 1 && p.fast_colour)
                        update_voxels<1, V, Sh::NV, true, true>(tv, wv, rv, gv, bv, dist[g], rg, okg, p.obs_w);
                    else if (p.obs_w == 1.0f)
                        update_voxels<RM, V, Sh::NV, true, false>(tv, wv, rv, gv, bv, dist[g], rg, okg, p.obs_w);
                    else
                        update_voxels<RM, V, Sh::NV, false, false>(tv, wv, rv, gv, bv, dist[g], rg, okg, p.obs_w);
                    const V cn = pack_colour(rv, gv, bv);
#pragma unroll
                    for (int i = 0; i < Sh::NV; ++i) {
                        const int j = g * Sh::NV + i;
                        t[j] = v_get(tv, i);
                        w[j] = v_get(wv, i);
                        if (ok[j]) c[j] = v_get(cn, i);  // (a voxel that does not update keeps its stored colour bits)
                        if (COUNT && ok[j]) ++n_upd;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < VPT; ++j)
                    if (ok[j]) {
                        const float d = v_get(dist[j / Sh::NV], j % Sh::NV);
                        t[j] = t[j] + p.obs_w * d;
                        w[j] = w[j] + p.obs_w;
                        c[j] = c[j] + p.obs_w * (float)(rgb[j] & 255u);
                        c3[j] = c3[j] + p.obs_w * (float)((rgb[j] >> 8) & 255u);
                        c4[j] = c4[j] + p.obs_w * (float)((rgb[j] >> 16) & 255u);
                        if (COUNT) ++n_upd;
                    }
            }
            vol_store4(v0 + idx, t, nrow);
            if (!ACCUM) {
                vol_store4(v1 + idx, w, nrow);
                vol_store4(v2 + idx, c, nrow);
            } else {
                vol_store4(v0 + plane + idx, w, nrow);
                vol_store4(v0 + 2 * plane + idx, c, nrow);
                vol_store4(v0 + 3 * plane + idx, c3, nrow);
                vol_store4(v0 + 4 * plane + idx, c4, nrow);
            }
        }
    }
    if (COUNT) {
        // one atomic per WORKGROUP, spread over COUNT_SLOTS counters on separate 128-byte lines (summed on read-back): one same-address
        // atomic per wave cost 863 us per frame at 512^3 against ~90 us for the sweep itself (profiles/r02_kernel_stats.csv)
        __shared__ unsigned wave_cnt[4];
        for (int off = 32; off > 0; off >>= 1) n_upd += __shfl_xor((int)n_upd, off);
        if (lane == 0) wave_cnt[threadIdx.x >> 6] = n_upd;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
            if (total) atomicAdd(p.n_updated + (size_t)(blockIdx.x % COUNT_SLOTS) * COUNT_STRIDE, (unsigned long long)total);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// SEVERAL FRAMES PER LAUNCH (hive_tsdf_integrate_batch on device frames).  Consecutive frames of a sequence see almost the same part
// of the volume, and a voxel's update depends on its own state only: a trip loads its voxels ONCE, applies frames 0 .. nf - 1 to the
// registers in sequence order -- the same operations in the same order as nf launches, so the result is bit-identical -- and stores
// them once.  Per frame the gather role, the LDS exchange and the tests are those of integrate_kernel; the volume is loaded by a lane
// at the first frame that updates one of its voxels.  The work list is built over the UNION of the frames' clipped intervals (every
// voxel still runs every frame's exact tests).  What this buys: the volume traffic of nf frames for the price of one, the step
// becomes bound by the per-frame arithmetic (the single-frame kernel keeps its SIMDs 56 % busy).  (Tried and taken out: giving each XCD
// its own contiguous eighth of the work list, so that the texels of the four frames stay in its L2 -- 67 instead of 60 us per frame: the
// eighths are not equally expensive.)
#ifndef HIVE_TSDF_MAXF
#define HIVE_TSDF_MAXF 4
#endif
constexpr int MAXF = HIVE_TSDF_MAXF;
static_assert(MAXF <= 4, "WorkItem::zz holds a 4-bit frame mask");
struct MultiParams {
    FrameParams f[MAXF];  // the per-frame fields (R, T, frame / depth / rgb, tile_max) differ; the rest is the same in all
    int nf;
    int frame_skip;  // 1: a wave skips the frames whose clip excludes all of its segments (work item masks)
    int xcd_split;   // 1: the (band-sorted) work list's eighths go to the eight XCDs; 0: one grid-stride sweep over the whole list
    int bins_x;      // the sort key's image tiling: NBINS = (NBINS / bins_x) rows x bins_x columns of tiles of the first frame (1: full-width bands)
    unsigned *clear_next;  // the other scalar block: MS_CLEAR words to zero for the next sweep
    int lanes_along_x;     // work-list kernel: 1 = its lanes (neighbouring rows) run along x, 0 = along y
    int quad_interleave;   // work-list kernel: 1 = the segments of four neighbouring rows are interleaved
};

// scalar block of a fused sweep (two alternate, hive_ctx::d_scalars + MS_BASE + which * MS_STRIDE): [MS_NITEMS] work-list length, [MS_HIST ..] items
// per image band, [MS_CURSOR ..] the sort's per-band write cursors; the sweep's first kernel clears the OTHER block's first MS_CLEAR words
constexpr int MS_BASE = 2304, MS_STRIDE = 512, MS_NITEMS = MAXF, MS_HIST = 8, MS_CURSOR = MS_HIST + NBINS, MS_CLEAR = MS_CURSOR + NBINS;

__global__ __launch_bounds__(1024) void build_worklist_multi_kernel(MultiParams mp, WorkItem *__restrict__ items, unsigned *n_items, unsigned *hist) {
    __shared__ unsigned wave_sum[16];
    __shared__ unsigned block_base;
    __shared__ float dil[MAXF][MAX_TILES];
    __shared__ unsigned gmax[MAXF];
    __shared__ float gsum[MAXF];
    __shared__ unsigned bin_count[NBINS];
    constexpr int CHUNK = SEG_LANES * VPT;
    const FrameParams &p = mp.f[0];
    if (threadIdx.x < MAXF) gmax[threadIdx.x] = 0u, gsum[threadIdx.x] = 0.f;
    if (threadIdx.x < NBINS) bin_count[threadIdx.x] = 0u;
    __syncthreads();
    for (int f = 0; f < mp.nf; ++f) load_dilated_tiles(mp.f[f].tile_max, p.tiles_x, p.tiles_y, dil[f], &gmax[f], &gsum[f], threadIdx.x, 1024);
    __syncthreads();
    // Lanes run along the volume axis (x or y) that lies most ACROSS the camera's vertical: neighbouring lanes' rows then project to
    // horizontally neighbouring pixels -- the same 64-byte lines of the row-major texel planes -- and the work list interleaves the
    // segments of four neighbouring rows (below), so that a wave's four segments gather from the same lines.
    const long long row = (long long)blockIdx.x * 1024 + threadIdx.x;
    unsigned n_chunks = 0;
    int zstart = 0, z1 = 0, x = 0, y = 0;
    int fz0[MAXF], fz1[MAXF];  // the frames' own intervals (empty: 0, 0)
    float ax0 = 0.f, ay0 = 0.f, az0 = 0.f;  // row constants of the sweep's first frame (image tile of a segment)
    auto frames_of = [&](int zs) {  // frames whose own interval meets [zs, zs + CHUNK): the others cannot update a voxel of this segment
        unsigned mask = 0;
#pragma unroll
        for (int f = 0; f < MAXF; ++f) mask |= (fz1[f] > zs && fz0[f] < zs + CHUNK) ? (1u << f) : 0u;
        return mask;
    };
    if (row < (long long)p.X * p.Y) {
        if (mp.lanes_along_x) {
            y = (int)(row / p.X);
            x = (int)(row % p.X);
        } else {
            x = (int)(row / p.Y);
            y = (int)(row % p.Y);
        }
        int lo = p.Z, hi = 0;
#pragma unroll
        for (int f = 0; f < MAXF; ++f) {
            fz0[f] = fz1[f] = 0;
            if (f < mp.nf) {
                const FrameParams &q = mp.f[f];
                const float tx = (q.ox + (float)(x + q.x_off) * q.vs) - q.T[0];
                const float ty = (q.oy + (float)y * q.vs) - q.T[1];
                const float ax = q.R[0] * tx + q.R[3] * ty;
                const float ay = q.R[1] * tx + q.R[4] * ty;
                const float az = q.R[2] * tx + q.R[5] * ty;
                if (f == 0) ax0 = ax, ay0 = ay, az0 = az;
                const bool row_far = p.row_far == 2 || (p.row_far == 1 && row_far_pays(gsum[f], gmax[f], p.tiles_x * p.tiles_y));  // (workgroup-uniform)
                const RowClip clip = clip_row(q, ax, ay, az, __uint_as_float(gmax[f]), row_far ? dil[f] : nullptr);
                if (clip.z1 > clip.z0) {
                    fz0[f] = clip.z0, fz1[f] = clip.z1;
                    lo = min(lo, clip.z0);
                    hi = max(hi, clip.z1);
                }
            }
        }
        if (hi > lo) {
            zstart = (lo / VPT) * VPT;
            z1 = hi;
            // (a segment in a gap between the frames' intervals is in no frame's clip: it is not emitted)
            for (int zs = zstart; zs < z1; zs += CHUNK) n_chunks += frames_of(zs) ? 1u : 0u;
        }
    }
    const unsigned first = worklist_slots(n_chunks, n_items, wave_sum, &block_base);
    // Quad interleave: the four lanes 4 q .. 4 q + 3 (four neighbouring rows) emit segment e of lane 0, of lane 1, ... then segment e + 1:
    // item (lane j, e) goes to quad_first + sum_j' min(n_j', e) + #{j' < j : n_j' > e}.  A trip of the sweep takes four consecutive items.
    const int lane = threadIdx.x & 63;
    const unsigned quad_first = (unsigned)__shfl((int)first, lane & ~3);
    unsigned nq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) nq[j] = (unsigned)__shfl((int)n_chunks, (lane & ~3) + j);
    unsigned e = 0;
    if (n_chunks)
        for (int zs = zstart; zs < z1; zs += CHUNK) {
            const unsigned mask = frames_of(zs);
            if (!mask) continue;
            unsigned slot = quad_first;
            if (mp.quad_interleave) {
#pragma unroll
                for (int j = 0; j < 4; ++j) slot += min(nq[j], e) + ((j < (lane & 3) && nq[j] > e) ? 1u : 0u);
            } else {
                slot = first + e;
            }
            ++e;
            // image band: the row v of the first frame the segment's middle projects to (clamped; any value is valid -- it only orders the list)
            const float tz = (p.oz + (float)(zs + CHUNK / 2) * p.vs) - p.T[2];
            const float cz = az0 + p.R[8] * tz;
            const float v = cz > 1.0e-6f ? p.fy * ((ay0 + p.R[7] * tz) / cz) + p.cy : 0.f;
            // (round 5) the key is a 2-D image TILE where bins_x > 1: bins_y x bins_x tiles, row-major.  A 64-voxel segment is a radial image run of up to a few
            // hundred pixels; with full-width bands (17 rows at 1080p) it crosses up to 16 of them and an XCD's gathers range over all of those stripes of four
            // frames -- 16 MB against a 4 MB L2 (counters at 1920 x 1080 into 1024^3: L2 hit rate 46 %, FETCH_SIZE 4.7 GB per launch against 2.4 GB that must move;
            // at 640 x 480: 83 %).  A compact tile keeps what the XCD is gathering from at any one time to a few tiles of each frame.
            const int bins_y = NBINS / mp.bins_x;
            int bin = min(max((int)(v * ((float)bins_y / (float)p.H)), 0), bins_y - 1);
            if (mp.bins_x > 1) {
                const float u = cz > 1.0e-6f ? p.fx * ((ax0 + p.R[6] * tz) / cz) + p.cx : 0.f;
                bin = bin * mp.bins_x + min(max((int)(u * ((float)mp.bins_x / (float)p.W)), 0), mp.bins_x - 1);
            }
            atomicAdd(&bin_count[bin], 1u);
            WorkItem it;
            it.xy = (unsigned)x | ((unsigned)y << 16);
            it.zz = (unsigned)zs | ((unsigned)(min(z1 - zs, CHUNK) - 1) << ITEM_NLIVE_SHIFT) | (mask << ITEM_MASK_SHIFT) | ((unsigned)bin << ITEM_BIN_SHIFT);
            items[slot] = it;
        }
    __syncthreads();
    if (threadIdx.x < NBINS && bin_count[threadIdx.x]) atomicAdd(hist + threadIdx.x, bin_count[threadIdx.x]);
}

// Counting sort of the work list by image band (64 bands of the first frame's rows): the sorted list's eighths go to the eight XCDs
// (integrate_multi_kernel), so that an XCD's waves gather from one stripe of each frame at a time -- four 2.46 MB frames of texels do
// not fit a 4 MB L2 beside the volume stream when every XCD walks the whole image (round 3: L2 misses of the gathers several times
// the volume traffic; with all four frames reading ONE frame's texels the launch took 186 instead of 222 us).  One workgroup per
// 1024 consecutive items; inside a wave the items of a band keep their order (consecutive segments of a row stay neighbours).
__global__ __launch_bounds__(1024) void sort_worklist_kernel(const WorkItem *__restrict__ tmp, WorkItem *__restrict__ out, const unsigned *__restrict__ n_items_ptr,
                                                             const unsigned *__restrict__ hist, unsigned *cursor) {
    __shared__ unsigned base[NBINS], cnt[NBINS], gofs[NBINS];
    const unsigned n = *n_items_ptr;
    const unsigned i0 = blockIdx.x * 1024u;
    if (i0 >= n) return;
    if (threadIdx.x < NBINS) cnt[threadIdx.x] = 0u;
    if (threadIdx.x == 0) {
        unsigned a = 0;
        for (int b = 0; b < NBINS; ++b) {
            base[b] = a;
            a += hist[b];
        }
    }
    __syncthreads();
    const unsigned i = i0 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    WorkItem it;
    it.xy = it.zz = 0u;
    unsigned bin = 0xffffffffu;
    if (i < n) {
        it = tmp[i];
        bin = it.zz >> ITEM_BIN_SHIFT;
    }
    unsigned rank = 0, wave_base = 0;
    unsigned long long todo = __ballot(i < n);
    while (todo) {  // (wave-uniform: one round per distinct band of the wave's items)
        const int leader = __ffsll((long long)todo) - 1;
        const unsigned b = (unsigned)__shfl((int)bin, leader);
        const unsigned long long m = __ballot(bin == b);
        unsigned wb = 0;
        if (lane == leader) wb = atomicAdd(&cnt[b], (unsigned)__popcll(m));
        wb = (unsigned)__shfl((int)wb, leader);
        if (bin == b) {
            rank = (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            wave_base = wb;
        }
        todo &= ~m;
    }
    __syncthreads();
    if (threadIdx.x < NBINS && cnt[threadIdx.x]) gofs[threadIdx.x] = atomicAdd(cursor + threadIdx.x, cnt[threadIdx.x]);
    __syncthreads();
    if (i < n) out[base[bin] + gofs[bin] + wave_base + rank] = it;
}

// UPD: 0 = any observation weight, 1 = obs_weight == 1, 2 = obs_weight == 1 with the division-free colour update (update_voxels FASTC)
#ifdef HIVE_TSDF_WAVES
#define HIVE_TSDF_OCC __attribute__((amdgpu_waves_per_eu(HIVE_TSDF_WAVES, HIVE_TSDF_WAVES)))
#else
#define HIVE_TSDF_OCC
#endif
// The work list is sorted by image band (sort_worklist_kernel) and the XCDs take the sorted list's eighths: workgroups with the same
// blockIdx % 8 share an XCD (and its L2), so each XCD's gathers stay within one stripe of each frame.
// (Round 4, measured and taken out: gather instruction k covering the 64 consecutive voxels of segment k -- one pixel run per instruction
// instead of four runs of 16 voxels, i.e. 12 fewer distinct 64-byte lines per trip and frame -- with segment k's row constants from readlane:
// 210 vs 190 us per launch on the room scene, 198 vs 184 on the bench scene: the four rows' constants per frame cost more than the lines save.)
template <int RM, int UPD>
__global__ __launch_bounds__(256) HIVE_TSDF_OCC void integrate_multi_kernel(MultiParams mp, const WorkItem *__restrict__ items, const unsigned *__restrict__ n_items_ptr,
                                                              float *__restrict__ v0, float *__restrict__ v1, float *__restrict__ v2) {
    constexpr int PER_WAVE = 64 / SEG_LANES, SEG_VOX = SEG_LANES * 4;
    typedef ItemShape Sh;
    typedef Sh::V V;
    __shared__ uint2 xchg[4 * PER_WAVE * SEG_VOX];
    const int lane = threadIdx.x & 63;
    const int seg = lane / SEG_LANES, sl = lane % SEG_LANES;
    const unsigned n_items = *n_items_ptr;
    const unsigned n_trips = (n_items + PER_WAVE - 1) / PER_WAVE;
    // XCD x = blockIdx % 8 sweeps trips [x per_xcd, (x + 1) per_xcd) with its gridDim / 8 workgroups (the host launches a multiple of 8)
    const unsigned per_xcd = mp.xcd_split ? (n_trips + 7u) / 8u : n_trips;
    const unsigned xcd = mp.xcd_split ? (blockIdx.x & 7u) : 0u;
    const unsigned local_block = mp.xcd_split ? (blockIdx.x >> 3) : blockIdx.x;
    const unsigned stride = (mp.xcd_split ? (gridDim.x >> 3) : gridDim.x) * 4;
    const unsigned trip_end = min(n_trips, (xcd + 1u) * per_xcd);
    const FrameParams &p0 = mp.f[0];
    const float trunc_rcp = refined_rcp(p0.trunc);
    // the NEXT sweep's scalar block (work-list length, band histogram, sort cursors) is cleared here instead of by a memset launch per
    // sweep: the two blocks alternate, and every kernel of the sweep that used that block last has finished (stream order)
    if (blockIdx.x == 0 && (int)threadIdx.x < MS_CLEAR) mp.clear_next[threadIdx.x] = 0u;
    for (unsigned trip = xcd * per_xcd + local_block * 4 + (threadIdx.x >> 6); trip < trip_end; trip += stride) {
        const unsigned ii = trip * PER_WAVE + (unsigned)seg;
        WorkItem item;
        item.xy = item.zz = 0u;
        if (ii < n_items) item = items[ii];
        const int x = (int)(item.xy & 0xffffu), y = (int)(item.xy >> 16);
        const int zseg = (int)(item.zz & 0xffffu);
        const int zb = zseg + sl * VPT;
        const bool live = sl * VPT < item_nlive(item.zz);
        const int nrow = p0.Z - zb;  // voxels of this quad inside the row (< 4 only in the last quad of a row with Z % 4 != 0)
        const long long idx = ((long long)x * p0.Y + y) * p0.Z + zb;
        // frames that can update a voxel of ANY of the wave's segments (build_worklist_multi_kernel): the others are skipped by the
        // whole wave -- no projection, no gather, no tests.  (A skipped frame's own clip excludes every voxel of the segments, and the
        // clip is conservative: skipping cannot change a result.)
        unsigned wave_frames = (item.zz >> ITEM_MASK_SHIFT) & 15u;
#pragma unroll
        for (int o = SEG_LANES; o < 64; o <<= 1) wave_frames |= (unsigned)__shfl_xor((int)wave_frames, o);
        wave_frames = mp.frame_skip ? (unsigned)__builtin_amdgcn_readfirstlane((int)wave_frames) : 0xffu;
        float t[VPT], w[VPT], cr[VPT], cg[VPT], cb[VPT];  // the colour as three channels while the voxels are in registers
        bool loaded = false;
        uint2 *mine = xchg + ((threadIdx.x >> 6) * PER_WAVE + seg) * SEG_VOX;
#pragma unroll 1  // one frame's parameters in scalar registers at a time (unrolled, the four sets spill 160 SGPRs)
        for (int f = 0; f < mp.nf; ++f) {
            if (!((wave_frames >> f) & 1u)) continue;  // (wave-uniform)
            const FrameParams &p = mp.f[f];
            const float tx = (p.ox + (float)(x + p.x_off) * p.vs) - p.T[0];
            const float ty = (p.oy + (float)y * p.vs) - p.T[1];
            const float ax = p.R[0] * tx + p.R[3] * ty;
            const float ay = p.R[1] * tx + p.R[4] * ty;
            const float az = p.R[2] * tx + p.R[5] * ty;
            // gather role: voxels zseg + SEG_LANES k + sl, k = 0 .. 3
            // (Round 4, measured and taken out: depth and colour as separate planes, the 4-byte depth gathered first, the depth tests run
            // in the gather role and the colour fetched for the passing voxels only -- the second dependent round trip costs more than
            // the lines it saves: 292 vs 224 us per four-frame launch on the room scene, 261 vs 219 on the bench scene.)
            uint2 tex[4];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                V cz;
                int pix[Sh::NV];
                voxel_pixels<RM, V, Sh::NV, SEG_LANES>(p, ax, ay, az, zseg + sl + 2 * SEG_LANES * g, cz, pix);
#pragma unroll
                for (int i = 0; i < Sh::NV; ++i) {
                    uint2 tx2 = p.frame[max(pix[i], 0)];
                    if (pix[i] < 0) tx2.x = 0u;
                    tex[g * Sh::NV + i] = tx2;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) mine[SEG_LANES * k + sl] = tex[k];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint4 lo = *reinterpret_cast<const uint4 *>(mine + 4 * sl);
            const uint4 hi = *reinterpret_cast<const uint4 *>(mine + 4 * sl + 2);
            const float depth_v[VPT] = {__uint_as_float(lo.x), __uint_as_float(lo.z), __uint_as_float(hi.x), __uint_as_float(hi.z)};
            const unsigned rgb[VPT] = {lo.y, lo.w, hi.y, hi.w};
            __builtin_amdgcn_wave_barrier();  // the next frame's writes stay behind these reads
            V cam_z[Sh::NG], dist[Sh::NG];
#pragma unroll
            for (int g = 0; g < Sh::NG; ++g) cam_z[g] = voxel_cam_z<V, Sh::NV>(p, az, zb + g * Sh::NV);
            bool ok[VPT];
            const bool any = depth_tests<V>(p, trunc_rcp, depth_v, cam_z, live, nrow, dist, ok);
            if (any) {
                if (!loaded) {  // the lane's first frame with an update: its voxels come in now and stay in registers
                    float c[VPT];
                    vol_load4(t, v0 + idx, nrow);
                    vol_load4(w, v1 + idx, nrow);
                    vol_load4(c, v2 + idx, nrow);
#pragma unroll
                    for (int j = 0; j < VPT; ++j) unpack_colour<float, 1>(c[j], cr[j], cg[j], cb[j]);
                    loaded = true;
                }
#pragma unroll
                for (int g = 0; g < Sh::NG; ++g) {
                    V tv, wv, rv, gv, bv;
                    unsigned rg[Sh::NV];
                    bool okg[Sh::NV];
#pragma unroll
                    for (int i = 0; i < Sh::NV; ++i) {
                        const int j = g * Sh::NV + i;
                        v_set(tv, i, t[j]);
                        v_set(wv, i, w[j]);
                        v_set(rv, i, cr[j]);
                        v_set(gv, i, cg[j]);
                        v_set(bv, i, cb[j]);
                        rg[i] = rgb[j];
                        okg[i] = ok[j];
                    }
                    update_voxels<RM, V, Sh::NV, UPD >= 1, UPD == 2>(tv, wv, rv, gv, bv, dist[g], rg, okg, p.obs_w);
#pragma unroll
                    for (int i = 0; i < Sh::NV; ++i) {
                        const int j = g * Sh::NV + i;
                        t[j] = v_get(tv, i);
                        w[j] = v_get(wv, i);
                        cr[j] = v_get(rv, i);
                        cg[j] = v_get(gv, i);
                        cb[j] = v_get(bv, i);
                    }
                }
            }
        }
        if (loaded) {
            // (a voxel of the quad that no frame updated gets its colour back as pack(unpack(c)) = c: the plane holds integers below 2^24)
            float c[VPT];
#pragma unroll
            for (int j = 0; j < VPT; ++j) c[j] = pack_colour(cr[j], cg[j], cb[j]);
            vol_store4(v0 + idx, t, nrow);
            vol_store4(v1 + idx, w, nrow);
            vol_store4(v2 + idx, c, nrow);
        }
    }
}

__global__ __launch_bounds__(256) void fill3_kernel(float *__restrict__ a, float *__restrict__ b, float *__restrict__ c,
                                                    long long n, float va, float vb, float vc) {
    const long long stride = (long long)gridDim.x * 256 * 4;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n && (((uintptr_t)(a + i) | (uintptr_t)(b + i) | (uintptr_t)(c + i)) & 15) == 0) {
            *reinterpret_cast<float4 *>(a + i) = make_float4(va, va, va, va);
            *reinterpret_cast<float4 *>(b + i) = make_float4(vb, vb, vb, vb);
            *reinterpret_cast<float4 *>(c + i) = make_float4(vc, vc, vc, vc);
        } else {
            for (long long j = i; j < n && j < i + 4; ++j) {
                a[j] = va;
                b[j] = vb;
                c[j] = vc;
            }
        }
    }
}

// acc: 5 planes of `plane` floats each, of which elements [0, n) are folded into tsdf / weight / color [0, n)
template <int RM>
__global__ __launch_bounds__(256) void finalize_kernel(const float *__restrict__ acc, long long plane, long long n, float *__restrict__ tsdf,
                                                       float *__restrict__ weight, float *__restrict__ color) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float w = acc[1 * plane + i];
        float t = 1.0f, c = 0.0f;
        if (w > 0.0f) {
            t = acc[0 * plane + i] / w;
            const float r = fminf(hive_round<RM>(acc[2 * plane + i] / w), 255.0f);
            const float g = fminf(hive_round<RM>(acc[3 * plane + i] / w), 255.0f);
            const float b = fminf(hive_round<RM>(acc[4 * plane + i] / w), 255.0f);
            c = b * 65536.0f + g * 256.0f + r;
        }
        tsdf[i] = t;
        weight[i] = w;
        color[i] = c;
    }
}

// running-average volumes -> accumulator planes [num, w, r, g, b] = [tsdf * w, w, r * w, g * w, b * w]: a rank that fused
// its own frames with the ordinary integrate kernel contributes exactly these sums to the all-reduce
__global__ __launch_bounds__(256) void to_accum_kernel(const float *__restrict__ tsdf, const float *__restrict__ weight,
                                                       const float *__restrict__ color, long long n, float *__restrict__ acc) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float w = weight[i];
        const unsigned c = (unsigned)color[i];
        acc[0 * n + i] = tsdf[i] * w;  // w == 0 (never observed): tsdf is 1, the sum is 0
        acc[1 * n + i] = w;
        acc[2 * n + i] = (float)(c & 255u) * w;
        acc[3 * n + i] = (float)((c >> 8) & 255u) * w;
        acc[4 * n + i] = (float)(c >> 16) * w;
    }
}

// the same sums in the layout of ONE reduce-scatter: acc [world][5][chunk], voxel i in piece i / chunk at offset i % chunk; zeros past n
__global__ __launch_bounds__(256) void to_accum_sharded_kernel(const float *__restrict__ tsdf, const float *__restrict__ weight,
                                                               const float *__restrict__ color, long long n, long long chunk, long long total,
                                                               float *__restrict__ acc) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const long long r = i / chunk, j = i - r * chunk;
        float *o = acc + r * 5 * chunk + j;
        float w = 0.f, t = 0.f;
        unsigned c = 0u;
        if (i < n) {
            w = weight[i];
            t = tsdf[i];
            c = (unsigned)color[i];
        }
        o[0] = t * w;
        o[chunk] = w;
        o[2 * chunk] = (float)(c & 255u) * w;
        o[3 * chunk] = (float)((c >> 8) & 255u) * w;
        o[4 * chunk] = (float)(c >> 16) * w;
    }
}

// ---------------------------------------------------------------------------------------------
static int fill_volume(hive_tsdf *v) {
    hive_ctx *ctx = v->ctx;
    const int blocks = (int)std::min<long long>((v->n / 4 + 255) / 256 + 1, 256 * 16);
    hipLaunchKernelGGL(fill3_kernel, dim3(blocks), dim3(256), 0, ctx->stream, v->d_tsdf, v->d_weight, v->d_color,
                       (long long)v->n, 1.0f, 0.0f, 0.0f);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

// the scalar block of the frame in flight: d_scalars[0..7] or d_scalars[48..55]
static inline unsigned *tsdf_scalars(hive_ctx *ctx) { return ctx->d_scalars + (ctx->tsdf_scalars ? 48 : 0); }

// tiles of (32 << shift)^2 pixels, the smallest shift with at most MAX_TILES tiles (the work-list kernels keep MAXF tables in LDS)
struct TileGrid {
    int shift, tiles_x, tiles_y;
};
static TileGrid tile_grid(int H, int W) {
    TileGrid g;
    for (g.shift = 0;; ++g.shift) {
        const int ts = 32 << g.shift;
        g.tiles_x = (W + ts - 1) / ts;
        g.tiles_y = (H + ts - 1) / ts;
        if ((long long)g.tiles_x * g.tiles_y <= MAX_TILES) return g;
    }
}
static bool env_flag(const char *name, bool dflt) {  // tuning switches, read per call (A/B runs toggle them inside one process)
    const char *e = getenv(name);
    return e ? atoi(e) != 0 : dflt;
}

static int prepare_frame(hive_tsdf *v, const uint8_t *color, const float *depth, int H, int W, int mem,
                         const uint8_t **d_color, const float **d_depth) {
    hive_ctx *ctx = v->ctx;
    const size_t npx = (size_t)H * W;
    if (mem == HIVE_MEM_HOST) {
        const size_t depth_bytes = npx * sizeof(float), color_bytes = npx * 3;
        const size_t color_off = (depth_bytes + 255) & ~(size_t)255;
        int rc = hive_reserve_device(ctx, &ctx->d_in, &ctx->in_bytes, color_off + color_bytes);
        if (rc) return rc;
        if ((rc = hive_upload(ctx, ctx->d_in, depth, depth_bytes))) return rc;
        if ((rc = hive_upload(ctx, (char *)ctx->d_in + color_off, color, color_bytes))) return rc;
        *d_depth = (const float *)ctx->d_in;
        *d_color = (const uint8_t *)ctx->d_in + color_off;
    } else {
        *d_depth = depth;
        *d_color = color;
    }
    const TileGrid tg = tile_grid(H, W);
    const size_t tex_bytes = (npx * sizeof(uint2) + 255) & ~(size_t)255;
    int rc = hive_reserve_device(ctx, &ctx->d_frame, &ctx->frame_bytes, tex_bytes + MAX_TILES * sizeof(unsigned));
    if (rc) return rc;
    // scalar block of this frame: [4] work-list length (update counters: d_scalars + 128).  Two blocks alternate (both zero after
    // hive_ctx_create); the prep kernel of a frame clears the block of the next one.
    ctx->tsdf_scalars ^= 1;
    unsigned *next = ctx->d_scalars + (ctx->tsdf_scalars ? 0 : 48);
    unsigned *tiles = (unsigned *)((char *)ctx->d_frame + tex_bytes);
    const bool vec = W % 4 == 0 && ((uintptr_t)*d_depth % 16 == 0) && ((uintptr_t)*d_color % 4 == 0);
    const dim3 grid((unsigned)(tg.tiles_x * tg.tiles_y), 1);
    if (vec)
        hipLaunchKernelGGL(prep_frame_kernel<true>, grid, dim3(256), 0, ctx->stream, *d_depth, *d_color, H, W, tg.shift, tg.tiles_x, (uint2 *)ctx->d_frame, tiles,
                           MAX_TILES, next, 8);
    else
        hipLaunchKernelGGL(prep_frame_kernel<false>, grid, dim3(256), 0, ctx->stream, *d_depth, *d_color, H, W, tg.shift, tg.tiles_x, (uint2 *)ctx->d_frame, tiles,
                           MAX_TILES, next, 8);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

static void fill_frame_params(hive_tsdf *v, int H, int W, const float K[9], const double pose[16], float obs_weight, FrameParams &p) {
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) p.R[3 * r + c] = (float)pose[4 * r + c];
        p.T[r] = (float)pose[4 * r + 3];
    }
    p.fx = K[0];
    p.fy = K[4];
    p.cx = K[2];
    p.cy = K[5];
    p.ox = v->origin[0];
    p.oy = v->origin[1];
    p.oz = v->origin[2];
    p.vs = v->voxel_size;
    p.trunc = v->trunc;
    p.obs_w = obs_weight;
    p.X = (int)v->dim[0];
    p.x_off = (int)v->x_off;
    p.Y = (int)v->dim[1];
    p.Z = (int)v->dim[2];
    p.H = H;
    p.W = W;
    const TileGrid tg = tile_grid(H, W);
    p.tile_shift = tg.shift;
    p.tiles_x = tg.tiles_x;
    p.tiles_y = tg.tiles_y;
    {
        const char *e = getenv("HIVE_TSDF_ROW_FAR");  // tuning: 0 never / 1 adaptive (default) / 2 always
        p.row_far = e ? std::min(2, std::max(0, atoi(e))) : 1;
    }
    p.frame = nullptr;
    p.tile_max = nullptr;
    // the division-free colour update: the roundf contract, unit observation weight, and a volume whose weights are all integers
    // below 65534 -- true as long as every integrate since the last reset had obs_weight == 1 (hive_tsdf::unit_weights / unit_frames)
    p.fast_colour = (v->round_mode == HIVE_ROUND_HALF_AWAY && obs_weight == 1.0f && v->unit_weights && env_flag("HIVE_TSDF_FAST_COLOUR", true)) ? 1 : 0;
    p.n_updated = nullptr;
}

// The fused sweep's inputs: the {depth, rgb} texels and the tile maxima of every frame (prep_frame_kernel; one launch for a whole batch of frames).
struct PreparedFrames {
    const uint2 *texels;       // [nf][H*W]
    const unsigned *tile_max;  // [nf][tile_stride], tiles of (32 << tile_grid(H, W).shift)^2 pixels
    int tile_stride;
};
// One prep launch for frames [0, n) of a device-resident batch into hive_ctx::d_batch (texels, then the tile maxima).
static int prepare_batch(hive_tsdf *v, int n, const uint8_t *color, const float *depth, int H, int W, PreparedFrames *out) {
    hive_ctx *ctx = v->ctx;
    const size_t npx = (size_t)H * W;
    const TileGrid tg = tile_grid(H, W);
    const size_t tex_bytes = ((size_t)n * npx * sizeof(uint2) + 255) & ~(size_t)255;
    int rc = hive_reserve_device(ctx, &ctx->d_batch, &ctx->batch_bytes, tex_bytes + (size_t)n * MAX_TILES * sizeof(unsigned));
    if (rc) return rc;
    unsigned *d_tiles = (unsigned *)((char *)ctx->d_batch + tex_bytes);
    // the 4-pixels-per-lane form needs every frame's depth 16-byte and colour 4-byte aligned
    const bool vec = W % 4 == 0 && (uintptr_t)depth % 16 == 0 && (uintptr_t)color % 4 == 0;
    const dim3 grid((unsigned)(tg.tiles_x * tg.tiles_y), (unsigned)n);
    if (vec)
        hipLaunchKernelGGL(prep_frame_kernel<true>, grid, dim3(256), 0, ctx->stream, depth, color, H, W, tg.shift, tg.tiles_x, (uint2 *)ctx->d_batch, d_tiles, MAX_TILES,
                           (unsigned *)nullptr, 0);
    else
        hipLaunchKernelGGL(prep_frame_kernel<false>, grid, dim3(256), 0, ctx->stream, depth, color, H, W, tg.shift, tg.tiles_x, (uint2 *)ctx->d_batch, d_tiles, MAX_TILES,
                           (unsigned *)nullptr, 0);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    out->texels = (const uint2 *)ctx->d_batch;
    out->tile_max = d_tiles;
    out->tile_stride = MAX_TILES;
    return HIVE_OK;
}

// nf (2 .. MAXF) prepared frames in ONE sweep (integrate_multi_kernel): one work list over the union of the frames' clips, sorted by image
// band, one integrate launch.
static int launch_integrate_multi(hive_tsdf *v, int nf, int H, int W, const float K[9], const double *poses, float obs_weight, const PreparedFrames &prepared) {
    hive_ctx *ctx = v->ctx;
    const size_t npx = (size_t)H * W;
    int rc = HIVE_OK;
    // scalar block of this sweep (MS_* above).  Two blocks alternate (both zero after hive_ctx_create); this sweep's integrate kernel clears the
    // other one, which the previous sweep used (stream order) -- no memset launch
    static_assert(MS_CLEAR <= 256 && MS_CLEAR <= MS_STRIDE, "integrate_multi_kernel clears the next block with one workgroup of 256 threads");
    ctx->tsdf_multi_scalars ^= 1;
    unsigned *sc = ctx->d_scalars + MS_BASE + (ctx->tsdf_multi_scalars ? MS_STRIDE : 0);
    unsigned *idle_block = ctx->d_scalars + MS_BASE + (ctx->tsdf_multi_scalars ? 0 : MS_STRIDE);
    const uint2 *texels = prepared.texels;
    const unsigned *tiles = prepared.tile_max;
    const int tile_stride = prepared.tile_stride;
    const bool sorted = env_flag("HIVE_TSDF_SORT", true);  // work list sorted by image band, its eighths to the eight XCDs
    MultiParams mp;
    mp.nf = nf;
    mp.frame_skip = env_flag("HIVE_TSDF_FRAME_SKIP", true) ? 1 : 0;
    mp.xcd_split = sorted ? 1 : 0;
    mp.clear_next = idle_block;
    // the camera's "down" axis in world coordinates is column 1 of the pose's rotation: lanes run along the volume axis it has less of
    const double *pose0 = poses;
    const bool auto_x = fabs(pose0[0 * 4 + 1]) <= fabs(pose0[1 * 4 + 1]);
    const char *lanes_env = getenv("HIVE_TSDF_LANES");  // tuning: "x" / "y" force the lane axis
    mp.lanes_along_x = lanes_env ? (lanes_env[0] == 'x') : (auto_x ? 1 : 0);
    mp.quad_interleave = env_flag("HIVE_TSDF_QUAD", true) ? 1 : 0;
    {
        const char *e = getenv("HIVE_TSDF_BINS_X");  // tuning: columns of the sort key's image tiling (1 = full-width bands, the rule of round 4)
        int bx = e ? atoi(e) : 1;
        if (bx != 1 && bx != 2 && bx != 4 && bx != 8 && bx != 16) bx = 1;
        mp.bins_x = bx;
    }
    for (int f = 0; f < nf; ++f) {
        fill_frame_params(v, H, W, K, poses + 16 * (size_t)f, obs_weight, mp.f[f]);
        mp.f[f].frame = texels + (size_t)f * npx;
        mp.f[f].tile_max = tiles + (size_t)f * tile_stride;
#ifdef HIVE_TSDF_TUNING  // tuning builds only (make tsdf_variants): a TIMING experiment with WRONG results -- every frame gathers from frame 0's texels
        if (env_flag("HIVE_TSDF_TIMING_SAME_TEXELS", false)) mp.f[f].frame = texels;
#endif
    }
    for (int f = nf; f < MAXF; ++f) mp.f[f] = mp.f[0];
    const FrameParams &p = mp.f[0];
    const long long rows = (long long)p.X * p.Y;
    const long long seg = SEG_LANES * 4;
    const size_t max_items = (size_t)rows * (size_t)((p.Z + seg - 1) / seg);
    // two lists: as built (row order), and sorted by image band
    if ((rc = hive_reserve_device(ctx, &ctx->d_scratch, &ctx->scratch_bytes, 2 * max_items * sizeof(WorkItem)))) return rc;
    WorkItem *built = (WorkItem *)ctx->d_scratch, *items = sorted ? built + max_items : built;
    unsigned *n_items = sc + MS_NITEMS;
    v->last_n_items = n_items;
    hipLaunchKernelGGL(build_worklist_multi_kernel, dim3((unsigned)((rows + 1023) / 1024)), dim3(1024), 0, ctx->stream, mp, built, n_items, sc + MS_HIST);
    if (sorted)
        hipLaunchKernelGGL(sort_worklist_kernel, dim3((unsigned)((max_items + 1023) / 1024)), dim3(1024), 0, ctx->stream, (const WorkItem *)built, items,
                           (const unsigned *)n_items, (const unsigned *)(sc + MS_HIST), sc + MS_CURSOR);
    const long long max_trips = (long long)((max_items + 64 / SEG_LANES - 1) / (64 / SEG_LANES));
    const long long blocks = std::min<long long>((long long)ctx->num_cus * 8 * HIVE_GRID_MULT, (max_trips + 3) / 4);
    const dim3 grid((unsigned)((blocks + 7) / 8 * 8)), block(256);  // (a multiple of 8: the XCDs' shares)
    if ((rc = hive_time_begin(ctx))) return rc;
    const int upd = p.fast_colour ? 2 : (obs_weight == 1.0f ? 1 : 0);
#define HIVE_LAUNCH(RM, UPD) \
    hipLaunchKernelGGL((integrate_multi_kernel<RM, UPD>), grid, block, 0, ctx->stream, mp, (const WorkItem *)items, (const unsigned *)n_items, v->d_tsdf, v->d_weight, v->d_color)
    switch ((v->round_mode ? 3 : 0) + upd) {
        case 0: HIVE_LAUNCH(0, 0); break;
        case 1: HIVE_LAUNCH(0, 1); break;
        case 3: HIVE_LAUNCH(1, 0); break;
        case 4: HIVE_LAUNCH(1, 1); break;
        case 5: HIVE_LAUNCH(1, 2); break;
        default: return hive_fail(ctx, HIVE_ERR_STATE, "integrate: no kernel for round mode %d / update form %d", v->round_mode, upd);
    }
#undef HIVE_LAUNCH
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    ++v->launches_seen;
    return hive_time_end(ctx);
}

template <bool ACCUM>
static int launch_integrate(hive_tsdf *v, float *accum, int H, int W, const float K[9], const double pose[16],
                            float obs_weight, bool count) {
    hive_ctx *ctx = v->ctx;
    FrameParams p;
    fill_frame_params(v, H, W, K, pose, obs_weight, p);
    p.frame = (const uint2 *)ctx->d_frame;
    p.tile_max = (const unsigned *)((const char *)ctx->d_frame + (((size_t)H * W * sizeof(uint2) + 255) & ~(size_t)255));
    p.n_updated = (unsigned long long *)(ctx->d_scalars + 128);
    if (count) HIVE_CHECK_HIP(ctx, hipMemsetAsync(p.n_updated, 0, COUNT_SLOTS * COUNT_STRIDE * sizeof(unsigned long long), ctx->stream));
    const long long rows = (long long)p.X * p.Y;
    float *a0 = ACCUM ? accum : v->d_tsdf;
    // work list: at most ceil(Z / segment) items per row
    const long long seg = SEG_LANES * VPT;
    const size_t max_items = (size_t)rows * (size_t)((p.Z + seg - 1) / seg);
    int rc = hive_reserve_device(ctx, &ctx->d_scratch, &ctx->scratch_bytes, max_items * sizeof(WorkItem));
    if (rc) return rc;
    WorkItem *items = (WorkItem *)ctx->d_scratch;
    unsigned *n_items = tsdf_scalars(ctx) + 4;
    v->last_n_items = n_items;
    const dim3 wl_grid((unsigned)((rows + 1023) / 1024));
    hipLaunchKernelGGL(build_worklist_kernel, wl_grid, dim3(1024), 0, ctx->stream, p, items, n_items);
    // grid-stride sweep: HIVE_GRID_MULT (16) x the resident workgroup count (8 workgroups of 4 waves per CU), so that the
    // dispatcher evens out trips of unequal cost (room scene, 512^3: x2 97, x4 90-94, x8 91, x16 87, x32 95 us; at x16 the
    // 32768 workgroups hold ~0.5 trips each: most waves run exactly one); a wave takes 64 / SEG_LANES items per trip
    const long long max_trips = (long long)((max_items + 64 / SEG_LANES - 1) / (64 / SEG_LANES));
    const dim3 grid((unsigned)std::min<long long>((long long)ctx->num_cus * 8 * HIVE_GRID_MULT, (max_trips + 3) / 4)), block(256);
    if ((rc = hive_time_begin(ctx))) return rc;
#define HIVE_LAUNCH(RM, CNT)                                                                                      \
    hipLaunchKernelGGL((integrate_kernel<RM, CNT, ACCUM>), grid, block, 0, ctx->stream, p, items, n_items, a0, \
                       v->d_weight, v->d_color, (long long)v->n)
    switch ((v->round_mode ? 2 : 0) | (count ? 1 : 0)) {
        case 0: HIVE_LAUNCH(0, false); break;
        case 1: HIVE_LAUNCH(0, true); break;
        case 2: HIVE_LAUNCH(1, false); break;
        case 3: HIVE_LAUNCH(1, true); break;
    }
#undef HIVE_LAUNCH
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    ++v->launches_seen;
    return hive_time_end(ctx);
}

// bookkeeping behind hive_tsdf::unit_weights: n more observations of weight obs_w are about to be integrated
static void note_observations(hive_tsdf *v, float obs_w, int64_t n) {
    if (obs_w != 1.0f) v->unit_weights = false;
    v->unit_frames += n;
    v->frames_seen += n;
    if (v->unit_frames > 65533) v->unit_weights = false;  // (w + 1 stays below 65536: the bound of the division-free colour update)
}

static double voxel_extent(const hive_tsdf *v) {  // longest side of the whole grid in metres
    const double gx = (double)v->grid_dim0, gy = (double)v->dim[1], gz = (double)v->dim[2];
    return std::max(gx, std::max(gy, gz)) * (double)v->voxel_size;
}

static int check_frame_args(hive_tsdf *vol, const void *color, const void *depth, int H, int W, const float *K,
                            const double *pose, int mem) {
    if (!vol) return hive_fail(nullptr, HIVE_ERR_INVALID, "vol is NULL");
    hive_ctx *ctx = vol->ctx;
    HIVE_REQUIRE(ctx, color && depth && K && pose, "integrate: NULL argument");
    HIVE_REQUIRE(ctx, H > 0 && W > 0 && H < (1 << 23) && W < (1 << 23) && (long long)H * W < (1ll << 30), "integrate: bad image size %dx%d", H, W);
    HIVE_REQUIRE(ctx, mem == HIVE_MEM_HOST || mem == HIVE_MEM_DEVICE, "integrate: bad mem kind %d", mem);
    HIVE_REQUIRE(ctx, K[0] != 0.f && K[4] != 0.f, "integrate: singular intrinsics");
    return HIVE_OK;
}

extern "C" {

int hive_tsdf_dims(const double vol_bnds[6], double voxel_size, int64_t vol_dim[3]) {
    if (!vol_bnds || !vol_dim || !(voxel_size > 0)) return hive_fail(nullptr, HIVE_ERR_INVALID, "hive_tsdf_dims: bad argument");
    for (int a = 0; a < 3; ++a) vol_dim[a] = (int64_t)ceil((vol_bnds[2 * a + 1] - vol_bnds[2 * a]) / voxel_size);
    return HIVE_OK;
}

int hive_tsdf_create(hive_ctx *ctx, const double vol_bnds[6], double voxel_size, float *d_tsdf, float *d_weight,
                     float *d_color, hive_tsdf **out) {
    return hive_tsdf_create_slab(ctx, vol_bnds, voxel_size, 0, -1, d_tsdf, d_weight, d_color, out);
}

int hive_tsdf_create_slab(hive_ctx *ctx, const double vol_bnds[6], double voxel_size, int64_t x_begin, int64_t x_end, float *d_tsdf,
                          float *d_weight, float *d_color, hive_tsdf **out) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, out && vol_bnds, "hive_tsdf_create: NULL argument");
    HIVE_REQUIRE(ctx, voxel_size > 0, "hive_tsdf_create: voxel_size must be positive, got %g", voxel_size);
    *out = nullptr;
    int64_t dim[3];
    hive_tsdf_dims(vol_bnds, voxel_size, dim);
    HIVE_REQUIRE(ctx, dim[0] > 0 && dim[1] > 0 && dim[2] > 0, "hive_tsdf_create: empty volume %lld x %lld x %lld",
                 (long long)dim[0], (long long)dim[1], (long long)dim[2]);
    HIVE_REQUIRE(ctx, dim[0] < 65536 && dim[1] < 65536 && dim[2] < 65536, "hive_tsdf_create: a volume dimension exceeds 65535");
    const int64_t gdim0 = dim[0];
    if (x_end < 0) x_end = gdim0;  // the whole grid
    HIVE_REQUIRE(ctx, 0 <= x_begin && x_begin < x_end && x_end <= gdim0, "hive_tsdf_create_slab: x range [%lld, %lld) outside [0, %lld)",
                 (long long)x_begin, (long long)x_end, (long long)gdim0);
    dim[0] = x_end - x_begin;
    const bool external = d_tsdf || d_weight || d_color;
    HIVE_REQUIRE(ctx, !external || (d_tsdf && d_weight && d_color), "hive_tsdf_create: pass all three volume pointers or none");
    hive_tsdf *v = new hive_tsdf();
    v->ctx = ctx;
    v->x_off = x_begin;
    v->grid_dim0 = gdim0;
    for (int a = 0; a < 3; ++a) {
        v->dim[a] = dim[a];
        v->bnds[2 * a] = vol_bnds[2 * a];
        v->bnds[2 * a + 1] = vol_bnds[2 * a] + (double)(a == 0 ? gdim0 : dim[a]) * voxel_size;  // as the reference library adjusts them
        v->origin[a] = (float)vol_bnds[2 * a];
    }
    v->n = dim[0] * dim[1] * dim[2];
    v->voxel_size = (float)voxel_size;
    v->trunc = (float)(5.0 * voxel_size);
    v->round_mode = ctx->round_mode;  // inherited at creation; hive_tsdf_set_round_mode changes it for this volume only
    if (external) {
        v->d_tsdf = d_tsdf;
        v->d_weight = d_weight;
        v->d_color = d_color;
    } else {
        v->owns = true;
        hipError_t e = hipSetDevice(ctx->device);
        if (e == hipSuccess) e = hipMalloc((void **)&v->d_tsdf, v->n * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void **)&v->d_weight, v->n * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void **)&v->d_color, v->n * sizeof(float));
        if (e != hipSuccess) {
            int rc = hive_fail(ctx, HIVE_ERR_NOMEM, "hive_tsdf_create: allocating 3 x %lld floats failed: %s", (long long)v->n,
                               hipGetErrorString(e));
            hive_tsdf_destroy(v);
            return rc;
        }
    }
    int rc = fill_volume(v);
    if (rc) {
        hive_tsdf_destroy(v);
        return rc;
    }
    *out = v;
    return HIVE_OK;
}

int hive_tsdf_destroy(hive_tsdf *v) {
    HIVE_ENTER(v ? v->ctx : nullptr);
    if (!v) return HIVE_OK;
    (void)hipStreamSynchronize(v->ctx->stream);
    hive_tsdf_free_mesh(v);
    if (v->d_vbase) (void)hipFree(v->d_vbase);
    if (v->d_blk) (void)hipFree(v->d_blk);
    if (v->owns) {
        if (v->d_tsdf) (void)hipFree(v->d_tsdf);
        if (v->d_weight) (void)hipFree(v->d_weight);
        if (v->d_color) (void)hipFree(v->d_color);
    }
    delete v;
    return HIVE_OK;
}

int hive_tsdf_set_round_mode(hive_tsdf *v, int mode) {
    if (!v) return hive_fail(nullptr, HIVE_ERR_INVALID, "vol is NULL");
    HIVE_REQUIRE(v->ctx, mode == HIVE_ROUND_HALF_EVEN || mode == HIVE_ROUND_HALF_AWAY, "round mode must be 0 or 1, got %d", mode);
    v->round_mode = mode;
    return HIVE_OK;
}

int hive_tsdf_reset(hive_tsdf *v) {
    HIVE_ENTER(v ? v->ctx : nullptr);
    if (!v) return hive_fail(nullptr, HIVE_ERR_INVALID, "vol is NULL");
    v->unit_weights = true;
    v->unit_frames = 0;
    v->frames_seen = v->launches_seen = 0;
    return fill_volume(v);
}

int hive_tsdf_planes_modified(hive_tsdf *v) {
    if (!v) return hive_fail(nullptr, HIVE_ERR_INVALID, "vol is NULL");
    v->unit_weights = false;
    v->n_verts = v->n_faces = -1;
    return HIVE_OK;
}

int hive_tsdf_info(hive_tsdf *v, int64_t vol_dim[3], float origin[3], double vol_bnds[6], float *voxel_size,
                   float *trunc_margin) {
    HIVE_ENTER(v ? v->ctx : nullptr);
    if (!v) return hive_fail(nullptr, HIVE_ERR_INVALID, "vol is NULL");
    for (int a = 0; a < 3; ++a) {
        if (vol_dim) vol_dim[a] = v->dim[a];
        if (origin) origin[a] = v->origin[a];
    }
    if (vol_bnds) memcpy(vol_bnds, v->bnds, sizeof(v->bnds));
    if (voxel_size) *voxel_size = v->voxel_size;
    if (trunc_margin) *trunc_margin = v->trunc;
    return HIVE_OK;
}

int hive_tsdf_stats(hive_tsdf *v, int64_t *frames, int64_t *launches) {
    if (!v) return hive_fail(nullptr, HIVE_ERR_INVALID, "vol is NULL");
    if (frames) *frames = v->frames_seen;
    if (launches) *launches = v->launches_seen;
    return HIVE_OK;
}

int hive_tsdf_slab_info(hive_tsdf *v, int64_t *x_begin, int64_t *grid_dim_x) {
    if (!v) return hive_fail(nullptr, HIVE_ERR_INVALID, "vol is NULL");
    if (x_begin) *x_begin = v->x_off;
    if (grid_dim_x) *grid_dim_x = v->grid_dim0;
    return HIVE_OK;
}

int hive_tsdf_device_ptrs(hive_tsdf *v, float **d_tsdf, float **d_weight, float **d_color) {
    HIVE_ENTER(v ? v->ctx : nullptr);
    if (!v) return hive_fail(nullptr, HIVE_ERR_INVALID, "vol is NULL");
    if (d_tsdf) *d_tsdf = v->d_tsdf;
    if (d_weight) *d_weight = v->d_weight;
    if (d_color) *d_color = v->d_color;
    return HIVE_OK;
}

int hive_tsdf_integrate(hive_tsdf *vol, const uint8_t *color, const float *depth, int H, int W, const float K[9],
                        const double cam_pose[16], float obs_weight, int mem, uint64_t *n_updated) {
    HIVE_ENTER(vol ? vol->ctx : nullptr);
    int rc = check_frame_args(vol, color, depth, H, W, K, cam_pose, mem);
    if (rc) return rc;
    hive_ctx *ctx = vol->ctx;
    const uint8_t *d_color;
    const float *d_depth;
    note_observations(vol, obs_weight, 1);
    if ((rc = prepare_frame(vol, color, depth, H, W, mem, &d_color, &d_depth))) return rc;
    if ((rc = launch_integrate<false>(vol, nullptr, H, W, K, cam_pose, obs_weight, n_updated != nullptr))) return rc;
    vol->n_verts = vol->n_faces = -1;
    if (n_updated) {
        unsigned long long slots[COUNT_SLOTS * COUNT_STRIDE];
        HIVE_CHECK_HIP(ctx, hipMemcpyAsync(slots, ctx->d_scalars + 128, sizeof(slots), hipMemcpyDeviceToHost, ctx->stream));
        HIVE_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        unsigned long long n = 0;
        for (int i = 0; i < COUNT_SLOTS; ++i) n += slots[i * COUNT_STRIDE];
        *n_updated = n;
    }
    return HIVE_OK;
}

int hive_tsdf_integrate_batch(hive_tsdf *vol, int n, const uint8_t *color, const float *depth, int H, int W,
                              const float K[9], const double *cam_poses, float obs_weight, int mem) {
    HIVE_ENTER(vol ? vol->ctx : nullptr);
    int rc = check_frame_args(vol, color, depth, H, W, K, cam_poses, mem);
    if (rc) return rc;
    HIVE_REQUIRE(vol->ctx, n >= 0, "integrate_batch: n must be >= 0");
    note_observations(vol, obs_weight, n);
    const size_t npx = (size_t)H * W;
    // device-resident frames on the vector path: groups of up to MAXF consecutive frames per sweep (bit-identical to one sweep each)
    static const char *frames_env = getenv("HIVE_TSDF_FRAMES_PER_LAUNCH");  // "1": the single-frame kernel (tuning / A-B)
    const int group = frames_env ? std::max(1, std::min(MAXF, atoi(frames_env))) : MAXF;
    const bool multi = mem == HIVE_MEM_DEVICE && group > 1;  // any volume shape, any image size (rows of any length take the 16-byte path)
    // Fusing pays when the frames look at (almost) the same voxels -- consecutive frames of a video; frames far apart share little and
    // every voxel of the union still runs every frame's tests.  Measured on the room scene (640 x 480 into 512^3, us per frame: alone |
    // pairs | fours): 2.4 degrees apart 90 | 70 | 62; 12 degrees 90 | - | 75; 15 degrees 85 | 77 | 82; 20 degrees 85 | 81 | 93; 30 degrees
    // 85 | 80 | 96; 45 degrees 85 | 85 | 122.  A group therefore grows only while the optical axis stays within 36 degrees of its FIRST
    // frame's (and the camera within a quarter of the volume's longest side).
    const double max_side = voxel_extent(vol);
    static const char *cos_env = getenv("HIVE_TSDF_FUSE_COS");  // tuning: cosine of the largest angle to the group's first frame
    const double fuse_cos = cos_env ? atof(cos_env) : 0.809017;  // cos 36 deg
    static const char *dist_env = getenv("HIVE_TSDF_FUSE_DIST");  // tuning: camera distance to the group's first frame, in longest volume sides
    const double fuse_dist = dist_env ? atof(dist_env) : 0.25;
    auto fusable = [&](int a, int b) {
        const double *pa = cam_poses + 16 * (size_t)a, *pb = cam_poses + 16 * (size_t)b;
        const double dot = pa[2] * pb[2] + pa[6] * pb[6] + pa[10] * pb[10];
        const double na = sqrt(pa[2] * pa[2] + pa[6] * pa[6] + pa[10] * pa[10]), nb = sqrt(pb[2] * pb[2] + pb[6] * pb[6] + pb[10] * pb[10]);
        const double dx = pa[3] - pb[3], dy = pa[7] - pb[7], dz = pa[11] - pb[11];
        return dot >= fuse_cos * na * nb && dx * dx + dy * dy + dz * dz <= fuse_dist * fuse_dist * max_side * max_side;
    };
    vol->last_groups.clear();
    // One prep launch (texels + tile maxima) serves up to PREP_CHUNK frames of the batch: at 4 frames per sweep a prep per sweep was 27 small
    // launches per 107-frame step.  [prep_lo, prep_hi): the frames the current chunk holds.
    constexpr int PREP_CHUNK = 128;
    PreparedFrames chunk{nullptr, nullptr, 0};
    int prep_lo = 0, prep_hi = 0;
    int f = 0;
    while (f < n) {
        int nf = 1;
        if (multi)
            while (nf < group && f + nf < n && fusable(f, f + nf)) ++nf;
        if (nf > 1) {
            if (f + nf > prep_hi) {
                prep_lo = f;
                prep_hi = std::min(n, f + PREP_CHUNK);
                if ((rc = prepare_batch(vol, prep_hi - prep_lo, color + (size_t)prep_lo * npx * 3, depth + (size_t)prep_lo * npx, H, W, &chunk))) return rc;
            }
            PreparedFrames mine = chunk;
            mine.texels += (size_t)(f - prep_lo) * npx;
            mine.tile_max += (size_t)(f - prep_lo) * chunk.tile_stride;
            if ((rc = launch_integrate_multi(vol, nf, H, W, K, cam_poses + 16 * (size_t)f, obs_weight, mine))) return rc;
        } else {
            const uint8_t *d_color;
            const float *d_depth;
            if ((rc = prepare_frame(vol, color + f * npx * 3, depth + f * npx, H, W, mem, &d_color, &d_depth))) return rc;
            if ((rc = launch_integrate<false>(vol, nullptr, H, W, K, cam_poses + 16 * (size_t)f, obs_weight, false))) return rc;
        }
        vol->last_groups.push_back(nf);
        f += nf;
    }
    vol->n_verts = vol->n_faces = -1;
    return HIVE_OK;
}

int hive_tsdf_last_batch_groups(hive_tsdf *vol, int *sizes, int capacity, int *n_groups) {
    if (!vol) return hive_fail(nullptr, HIVE_ERR_INVALID, "vol is NULL");
    HIVE_REQUIRE(vol->ctx, n_groups && capacity >= 0 && (sizes || capacity == 0), "last_batch_groups: bad arguments");
    *n_groups = (int)vol->last_groups.size();
    for (int i = 0; i < capacity && i < *n_groups; ++i) sizes[i] = vol->last_groups[i];
    return HIVE_OK;
}

int hive_tsdf_last_sweep_items(hive_tsdf *vol, uint64_t *n_items, int *segment_voxels) {
    HIVE_ENTER(vol ? vol->ctx : nullptr);
    if (!vol) return hive_fail(nullptr, HIVE_ERR_INVALID, "vol is NULL");
    hive_ctx *ctx = vol->ctx;
    HIVE_REQUIRE(ctx, n_items, "last_sweep_items: NULL argument");
    HIVE_REQUIRE(ctx, vol->last_n_items, "last_sweep_items: no sweep has run on this volume");
    unsigned n = 0;
    HIVE_CHECK_HIP(ctx, hipMemcpyAsync(&n, vol->last_n_items, sizeof(n), hipMemcpyDeviceToHost, ctx->stream));
    HIVE_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_items = n;
    if (segment_voxels) *segment_voxels = SEG_LANES * VPT;
    return HIVE_OK;
}

int hive_tsdf_get_volume(hive_tsdf *v, float *h_tsdf, float *h_color, float *h_weight) {
    HIVE_ENTER(v ? v->ctx : nullptr);
    if (!v) return hive_fail(nullptr, HIVE_ERR_INVALID, "vol is NULL");
    hive_ctx *ctx = v->ctx;
    const size_t bytes = (size_t)v->n * sizeof(float);
    if (h_tsdf) HIVE_CHECK_HIP(ctx, hipMemcpyAsync(h_tsdf, v->d_tsdf, bytes, hipMemcpyDefault, ctx->stream));
    if (h_color) HIVE_CHECK_HIP(ctx, hipMemcpyAsync(h_color, v->d_color, bytes, hipMemcpyDefault, ctx->stream));
    if (h_weight) HIVE_CHECK_HIP(ctx, hipMemcpyAsync(h_weight, v->d_weight, bytes, hipMemcpyDefault, ctx->stream));
    HIVE_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HIVE_OK;
}

int hive_tsdf_set_volume(hive_tsdf *v, const float *h_tsdf, const float *h_color, const float *h_weight) {
    HIVE_ENTER(v ? v->ctx : nullptr);
    if (!v) return hive_fail(nullptr, HIVE_ERR_INVALID, "vol is NULL");
    hive_ctx *ctx = v->ctx;
    const size_t bytes = (size_t)v->n * sizeof(float);
    if (h_tsdf) HIVE_CHECK_HIP(ctx, hipMemcpyAsync(v->d_tsdf, h_tsdf, bytes, hipMemcpyDefault, ctx->stream));
    if (h_color) HIVE_CHECK_HIP(ctx, hipMemcpyAsync(v->d_color, h_color, bytes, hipMemcpyDefault, ctx->stream));
    if (h_weight) HIVE_CHECK_HIP(ctx, hipMemcpyAsync(v->d_weight, h_weight, bytes, hipMemcpyDefault, ctx->stream));
    HIVE_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    v->n_verts = v->n_faces = -1;
    v->unit_weights = false;
    return HIVE_OK;
}

int hive_tsdf_accum_reset(hive_tsdf *v, float *d_accum) {
    HIVE_ENTER(v ? v->ctx : nullptr);
    if (!v) return hive_fail(nullptr, HIVE_ERR_INVALID, "vol is NULL");
    HIVE_REQUIRE(v->ctx, d_accum, "accum_reset: d_accum is NULL");
    HIVE_CHECK_HIP(v->ctx, hipMemsetAsync(d_accum, 0, 5 * (size_t)v->n * sizeof(float), v->ctx->stream));
    return HIVE_OK;
}

int hive_tsdf_accum_integrate(hive_tsdf *vol, float *d_accum, const uint8_t *color, const float *depth, int H, int W,
                              const float K[9], const double cam_pose[16], float obs_weight, int mem) {
    HIVE_ENTER(vol ? vol->ctx : nullptr);
    int rc = check_frame_args(vol, color, depth, H, W, K, cam_pose, mem);
    if (rc) return rc;
    HIVE_REQUIRE(vol->ctx, d_accum, "accum_integrate: d_accum is NULL");
    const uint8_t *d_color;
    const float *d_depth;
    if ((rc = prepare_frame(vol, color, depth, H, W, mem, &d_color, &d_depth))) return rc;
    ++vol->frames_seen;  // (the volume's own planes are untouched: unit_weights / unit_frames do not move)
    return launch_integrate<true>(vol, d_accum, H, W, K, cam_pose, obs_weight, false);
}

int hive_tsdf_accum_from_volume(hive_tsdf *v, float *d_accum) {
    HIVE_ENTER(v ? v->ctx : nullptr);
    if (!v) return hive_fail(nullptr, HIVE_ERR_INVALID, "vol is NULL");
    hive_ctx *ctx = v->ctx;
    HIVE_REQUIRE(ctx, d_accum, "accum_from_volume: d_accum is NULL");
    const int blocks = (int)std::min<long long>((v->n + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(to_accum_kernel, dim3(blocks), dim3(256), 0, ctx->stream, v->d_tsdf, v->d_weight, v->d_color, (long long)v->n, d_accum);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

int hive_tsdf_accum_from_volume_sharded(hive_tsdf *v, float *d_out, int world, int64_t chunk) {
    HIVE_ENTER(v ? v->ctx : nullptr);
    if (!v) return hive_fail(nullptr, HIVE_ERR_INVALID, "vol is NULL");
    hive_ctx *ctx = v->ctx;
    HIVE_REQUIRE(ctx, d_out && world > 0 && chunk > 0 && (int64_t)world * chunk >= v->n, "accum_from_volume_sharded: need world * chunk >= %lld voxels (world %d, chunk %lld)",
                 (long long)v->n, world, (long long)chunk);
    const long long total = (long long)world * chunk;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(to_accum_sharded_kernel, dim3(blocks), dim3(256), 0, ctx->stream, v->d_tsdf, v->d_weight, v->d_color, (long long)v->n, (long long)chunk, total, d_out);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

int hive_tsdf_set_volume_range(hive_tsdf *v, int64_t first, int64_t count, const float *d_tsdf, const float *d_weight, const float *d_color) {
    HIVE_ENTER(v ? v->ctx : nullptr);
    if (!v) return hive_fail(nullptr, HIVE_ERR_INVALID, "vol is NULL");
    hive_ctx *ctx = v->ctx;
    HIVE_REQUIRE(ctx, first >= 0 && count >= 0 && first + count <= v->n, "set_volume_range: [%lld, %lld) outside the %lld voxels", (long long)first,
                 (long long)(first + count), (long long)v->n);
    if (count == 0) return HIVE_OK;
    const size_t bytes = (size_t)count * sizeof(float);
    if (d_tsdf) HIVE_CHECK_HIP(ctx, hipMemcpyAsync(v->d_tsdf + first, d_tsdf, bytes, hipMemcpyDefault, ctx->stream));
    if (d_weight) HIVE_CHECK_HIP(ctx, hipMemcpyAsync(v->d_weight + first, d_weight, bytes, hipMemcpyDefault, ctx->stream));
    if (d_color) HIVE_CHECK_HIP(ctx, hipMemcpyAsync(v->d_color + first, d_color, bytes, hipMemcpyDefault, ctx->stream));
    v->n_verts = v->n_faces = -1;
    v->unit_weights = false;
    return HIVE_OK;
}

int hive_tsdf_accum_finalize(hive_tsdf *v, const float *d_accum) {
    if (!v) return hive_fail(nullptr, HIVE_ERR_INVALID, "vol is NULL");
    int rc = hive_tsdf_accum_finalize_to(v, d_accum, v->n, v->n, v->d_tsdf, v->d_weight, v->d_color);
    v->n_verts = v->n_faces = -1;
    v->unit_weights = false;
    return rc;
}

int hive_tsdf_accum_finalize_to(hive_tsdf *v, const float *d_accum, int64_t plane_stride, int64_t count, float *d_tsdf, float *d_weight,
                                float *d_color) {
    HIVE_ENTER(v ? v->ctx : nullptr);
    if (!v) return hive_fail(nullptr, HIVE_ERR_INVALID, "vol is NULL");
    hive_ctx *ctx = v->ctx;
    HIVE_REQUIRE(ctx, d_accum && d_tsdf && d_weight && d_color, "accum_finalize: NULL argument");
    HIVE_REQUIRE(ctx, count >= 0 && plane_stride >= count, "accum_finalize_to: %lld voxels, plane stride %lld", (long long)count, (long long)plane_stride);
    if (count == 0) return HIVE_OK;
    const int blocks = (int)std::min<long long>((count + 255) / 256, 256 * 32);
    if (v->round_mode)
        hipLaunchKernelGGL(finalize_kernel<1>, dim3(blocks), dim3(256), 0, ctx->stream, d_accum, (long long)plane_stride, (long long)count, d_tsdf, d_weight, d_color);
    else
        hipLaunchKernelGGL(finalize_kernel<0>, dim3(blocks), dim3(256), 0, ctx->stream, d_accum, (long long)plane_stride, (long long)count, d_tsdf, d_weight, d_color);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

}  // extern "C"
