// Internal declarations shared by the translation units of libhive_mi355x.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hive_mi355x.h"

#define HIVE_WAVE 64

struct hive_staging_slot {
    void *pinned = nullptr;
    size_t bytes = 0;
    hipEvent_t done = nullptr;
    bool in_flight = false;
};

struct hive_ctx {
    int device = 0;
    int num_cus = 256;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    int round_mode = HIVE_ROUND_HALF_EVEN;
    std::string last_error;

    // host -> device staging ring (pinned), so that a HOST call may return before the copy lands
    static constexpr int kSlots = 4;
    hive_staging_slot slots[kSlots];
    int next_slot = 0;

    // generic device scratch (grown on demand, never shrunk)
    void *d_scratch = nullptr;
    size_t scratch_bytes = 0;
    // packed frame {depth bits, rgb} + per-frame scalars
    void *d_frame = nullptr;
    size_t frame_bytes = 0;
    void *d_batch = nullptr;  // texels + tile maxima of a prepared batch of frames (tsdf.hip prepare_batch)
    size_t batch_bytes = 0;
    void *d_in = nullptr;  // device copy of host inputs
    size_t in_bytes = 0;
    int tsdf_scalars = 0;           // which of the two TSDF scalar blocks the frame in flight uses (tsdf.hip prepare_frame)
    int tsdf_multi_scalars = 0;     // the same for the multi-frame sweep's blocks (d_scalars + 64 / + 80)
    unsigned *d_scalars = nullptr;  // [0]=max depth bits, [2..3]=u64 counter, ...
    void *h_pinned_small = nullptr;  // 256 bytes of pinned host memory for small read-backs (mesh totals)
    void *d_zeros = nullptr;        // 256 bytes of zeros: the source of padding taps in the implicit-GEMM convolutions
    // split-K of the MFMA tile kernels at small batches (mfma_pipe.hpp splitk_*): f32 partial tiles + one arrival counter per tile (zero between launches)
    void *d_splitk = nullptr;
    size_t splitk_bytes = 0;
    unsigned *d_splitk_count = nullptr;
    void *d_gram = nullptr;  // partial Gram matrices / sums / quadratic forms of gram.hip (GroupNorm statistics of 1 x 1 convolutions)
    size_t gram_bytes = 0;

    // hive_ctx_set_deterministic: no split-K, no Gram-matrix GroupNorm statistics (the two paths whose use depends on the batch size and the CU count)
    bool deterministic = false;
    // launches of the small-launch paths since creation / the last reset (hive_ctx_launch_stats): split-K items, four-stage-ring kernels
    int64_t n_splitk_launches = 0, n_deep_ring_launches = 0;

    // HIP-event timing of the dominant kernel
    bool timing = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    hipEvent_t last_start = nullptr, last_stop = nullptr;
};

int hive_fail(hive_ctx *ctx, int code, const char *fmt, ...);

// Every extern "C" entry point runs with its context's device current and puts the caller's device back on
// return: a context (or a volume / ViT engine built on it) may belong to a GPU other than the calling
// thread's current one -- kernels, copies and allocations must land on the context's GPU, and the caller's
// (PyTorch's) current device must not change behind its back.
struct hive_device_guard {
    int prev = -1;
    bool switched = false;
    explicit hive_device_guard(const hive_ctx *ctx) {
        if (!ctx) return;
        if (hipGetDevice(&prev) == hipSuccess && prev != ctx->device) switched = hipSetDevice(ctx->device) == hipSuccess;
    }
    ~hive_device_guard() {
        if (switched) (void)hipSetDevice(prev);
    }
    hive_device_guard(const hive_device_guard &) = delete;
    hive_device_guard &operator=(const hive_device_guard &) = delete;
};
#define HIVE_ENTER(ctx) hive_device_guard _hive_device_guard(ctx)
void hive_set_global_error(const char *msg);

#define HIVE_CHECK_HIP(ctx, expr)                                                                        \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess)                                                                            \
            return hive_fail((ctx), _e == hipErrorOutOfMemory ? HIVE_ERR_NOMEM : HIVE_ERR_DEVICE,        \
                             "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define HIVE_REQUIRE(ctx, cond, ...)                                   \
    do {                                                               \
        if (!(cond)) return hive_fail((ctx), HIVE_ERR_INVALID, __VA_ARGS__); \
    } while (0)

// grows *ptr to at least `bytes` of device memory
int hive_reserve_device(hive_ctx *ctx, void **ptr, size_t *cur, size_t bytes);
// copies `bytes` from host memory to device memory `dst` through the pinned ring; returns once the
// host buffer may be reused by the caller
int hive_upload(hive_ctx *ctx, void *dst, const void *src, size_t bytes);
// dpt_ops.hip: (mean, rstd) per (sample, group) from the per-tile channel sums a GN convolution epilogue left (conv.hip)
int hive_gn_finalize_tiles(hive_ctx *ctx, const float *d_partial, int N, int HW, int C, int G, int tile_rows, float eps, float *d_stats);
// split-K workspace of at least `bytes` and the (zeroed) arrival counters (HIVE_SPLITK_TILES of them)
constexpr int HIVE_SPLITK_TILES = 4096;
int hive_splitk_workspace(hive_ctx *ctx, size_t bytes, void **ws, unsigned **count);
// gram.hip: (mean, rstd)[N][G] of GroupNorm(conv1x1(x, w)) from the Gram matrices of x and the tables hive_gn_gram_prepare made of w
int hive_gram_gn_stats(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, int C_in, int C_out, int stride, int Ho, int Wo, int G, const float *d_tables,
                       float eps, float *d_stats, float *d_S_out, float *d_s_out);
// event helpers for kernel timing
int hive_time_begin(hive_ctx *ctx);
int hive_time_end(hive_ctx *ctx);

struct hive_tsdf {
    hive_ctx *ctx = nullptr;
    int64_t dim[3] = {0, 0, 0};
    int64_t n = 0;
    int64_t x_off = 0;      // x-slab of a larger scene grid: this volume holds grid voxels x_off <= x < x_off + dim[0]
    int64_t grid_dim0 = 0;  // x dimension of the whole grid
    double bnds[6] = {0, 0, 0, 0, 0, 0};
    float origin[3] = {0, 0, 0};
    float voxel_size = 0.f;
    float trunc = 0.f;
    float *d_tsdf = nullptr, *d_weight = nullptr, *d_color = nullptr;
    bool owns = false;
    // pixel / colour rounding of integrate, finalize and the mesh colour lookup of THIS volume (hive_round_mode):
    // a property of the volume, not of the context, so that two volumes of one thread may differ
    int round_mode = HIVE_ROUND_HALF_EVEN;
    // frames per sweep of the most recent hive_tsdf_integrate_batch (1 = the single-frame kernel), in launch order
    std::vector<int> last_groups;
    // true while every weight of the volume is a whole number of unit observations: set by create / reset, kept by integrate calls with
    // obs_weight == 1 (at most 65533 frames: unit_frames), cleared by anything else that writes the planes (other observation weights,
    // set_volume / set_volume_range / accum_finalize, hive_tsdf_planes_modified).  Selects the division-free colour update (tsdf.hip).
    bool unit_weights = true;
    int64_t unit_frames = 0;
    // frames handed to the integrate entry points / integrate launches since creation or the last reset (hive_tsdf_stats)
    int64_t frames_seen = 0, launches_seen = 0;
    // device word holding the work-list length of the most recent sweep (valid until the next sweep but one on this context)
    const unsigned *last_n_items = nullptr;
    // mesh extraction results (device)
    int64_t n_verts = -1, n_faces = -1;
    int64_t cap_verts = 0, cap_faces = 0;  // capacity of the result arrays below (kept between extractions, grown on demand)
    float *d_verts = nullptr, *d_norms = nullptr, *d_verts_vox = nullptr;
    int32_t *d_faces = nullptr;
    uint8_t *d_vcolors = nullptr;
    // mesh scratch
    uint32_t *d_vbase = nullptr;  // per voxel: first vertex id | axis mask << 29
    size_t vbase_bytes = 0;
    uint32_t *d_blk = nullptr;  // per block counts / offsets
    size_t blk_bytes = 0;
};

void hive_tsdf_free_mesh(hive_tsdf *vol);

// rounding shared by device code: RM = 0 half-even (np.round), 1 half-away (roundf)
template <int RM>
__device__ __forceinline__ float hive_round(float x) {
    if (RM == 0) return rintf(x);
    return roundf(x);
}
