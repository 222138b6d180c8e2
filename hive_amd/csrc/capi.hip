// Context, error reporting, staging and timing plumbing of libhive_mi355x.so.
#include <algorithm>

#include "hive_internal.hpp"

static thread_local std::string g_global_error;

void hive_set_global_error(const char *msg) { g_global_error = msg ? msg : ""; }

int hive_fail(hive_ctx *ctx, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->last_error = buf;
    else
        g_global_error = buf;
    return code;
}

int hive_reserve_device(hive_ctx *ctx, void **ptr, size_t *cur, size_t bytes) {
    if (*cur >= bytes && *ptr) return HIVE_OK;
    if (*ptr) {
        HIVE_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        HIVE_CHECK_HIP(ctx, hipFree(*ptr));
        *ptr = nullptr;
        *cur = 0;
    }
    HIVE_CHECK_HIP(ctx, hipMalloc(ptr, bytes));
    *cur = bytes;
    return HIVE_OK;
}

int hive_splitk_workspace(hive_ctx *ctx, size_t bytes, void **ws, unsigned **count) {
    if (!ctx->d_splitk_count) {
        HIVE_CHECK_HIP(ctx, hipMalloc((void **)&ctx->d_splitk_count, HIVE_SPLITK_TILES * sizeof(unsigned)));
        HIVE_CHECK_HIP(ctx, hipMemsetAsync(ctx->d_splitk_count, 0, HIVE_SPLITK_TILES * sizeof(unsigned), ctx->stream));  // (the last arriver of a tile puts its counter back to 0)
    }
    int rc = hive_reserve_device(ctx, &ctx->d_splitk, &ctx->splitk_bytes, std::max(bytes, (size_t)32 << 20));
    if (rc) return rc;
    *ws = ctx->d_splitk;
    *count = ctx->d_splitk_count;
    return HIVE_OK;
}

int hive_upload(hive_ctx *ctx, void *dst, const void *src, size_t bytes) {
    hive_staging_slot &s = ctx->slots[ctx->next_slot];
    ctx->next_slot = (ctx->next_slot + 1) % hive_ctx::kSlots;
    if (s.in_flight) {
        HIVE_CHECK_HIP(ctx, hipEventSynchronize(s.done));
        s.in_flight = false;
    }
    if (s.bytes < bytes) {
        if (s.pinned) HIVE_CHECK_HIP(ctx, hipHostFree(s.pinned));
        s.pinned = nullptr;
        s.bytes = 0;
        HIVE_CHECK_HIP(ctx, hipHostMalloc(&s.pinned, bytes, hipHostMallocDefault));
        s.bytes = bytes;
    }
    if (!s.done) HIVE_CHECK_HIP(ctx, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    memcpy(s.pinned, src, bytes);
    HIVE_CHECK_HIP(ctx, hipMemcpyAsync(dst, s.pinned, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIVE_CHECK_HIP(ctx, hipEventRecord(s.done, ctx->stream));
    s.in_flight = true;
    return HIVE_OK;
}

static int next_event(hive_ctx *ctx, hipEvent_t *ev) {
    if (ctx->ev_used == ctx->ev_pool.size()) {
        hipEvent_t e;
        HIVE_CHECK_HIP(ctx, hipEventCreate(&e));
        ctx->ev_pool.push_back(e);
    }
    *ev = ctx->ev_pool[ctx->ev_used++];
    return HIVE_OK;
}

int hive_time_begin(hive_ctx *ctx) {
    if (!ctx->timing) return HIVE_OK;
    int rc = next_event(ctx, &ctx->last_start);
    if (rc) return rc;
    HIVE_CHECK_HIP(ctx, hipEventRecord(ctx->last_start, ctx->stream));
    return HIVE_OK;
}

int hive_time_end(hive_ctx *ctx) {
    if (!ctx->timing) return HIVE_OK;
    int rc = next_event(ctx, &ctx->last_stop);
    if (rc) return rc;
    HIVE_CHECK_HIP(ctx, hipEventRecord(ctx->last_stop, ctx->stream));
    return HIVE_OK;
}

extern "C" {

int hive_abi_version(void) { return HIVE_ABI_VERSION; }

int hive_ctx_create(int device_id, void *stream, hive_ctx **out) {
    if (!out) return hive_fail(nullptr, HIVE_ERR_INVALID, "hive_ctx_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return hive_fail(nullptr, HIVE_ERR_DEVICE,
                         "hive_ctx_create: no HIP device available (%s); libhive_mi355x has no CPU fallback",
                         hipGetErrorString(e));
    if (device_id < 0 || device_id >= count)
        return hive_fail(nullptr, HIVE_ERR_INVALID, "hive_ctx_create: device %d out of range [0,%d)", device_id, count);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) return hive_fail(nullptr, HIVE_ERR_DEVICE, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return hive_fail(nullptr, HIVE_ERR_DEVICE, "hive_ctx_create: device %d is %s; this library is built for gfx950 only",
                         device_id, prop.gcnArchName);
    hive_ctx *ctx = new hive_ctx();
    ctx->device = device_id;
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIVE_ENTER(ctx);  // device_id current for the allocations below; the caller's device is restored on return
    int current = -1;
    e = hipGetDevice(&current);
    if (e == hipSuccess && current != device_id) e = hipErrorInvalidDevice;  // the guard could not switch
    if (e == hipSuccess) {
        if (stream == HIVE_STREAM_OWN) {
            e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
            ctx->owns_stream = true;
        } else if (stream == HIVE_STREAM_OWN_LOW) {  // lowest dispatch priority: work that should only fill what other streams leave idle
            int least = 0, greatest = 0;
            e = hipDeviceGetStreamPriorityRange(&least, &greatest);
            if (e == hipSuccess) e = hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, least);
            ctx->owns_stream = true;
        } else {
            ctx->stream = (hipStream_t)stream;  // NULL = the default stream
        }
    }
    // [0, 128): scalar blocks of the kernels; [128, 128 + 2048): HIVE_COUNT_SLOTS update counters, one per 128-byte line (tsdf.hip);
    // [2304, 2304 + 2 x 512): the two scalar blocks of the fused sweep (tsdf.hip MS_BASE)
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_scalars, 4096 * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemset(ctx->d_scalars, 0, 4096 * sizeof(unsigned));  // the TSDF scalar blocks start cleared (tsdf.hip prepare_frame)
    if (e == hipSuccess) e = hipMalloc(&ctx->d_zeros, 256);
    if (e == hipSuccess) e = hipMemset(ctx->d_zeros, 0, 256);
    if (e != hipSuccess) {
        int rc = hive_fail(nullptr, HIVE_ERR_DEVICE, "hive_ctx_create: %s", hipGetErrorString(e));
        delete ctx;
        return rc;
    }
    *out = ctx;
    return HIVE_OK;
}

int hive_ctx_destroy(hive_ctx *ctx) {
    if (!ctx) return HIVE_OK;
    {
        HIVE_ENTER(ctx);
        (void)hipStreamSynchronize(ctx->stream);
    for (auto &s : ctx->slots) {
        if (s.pinned) (void)hipHostFree(s.pinned);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    for (auto ev : ctx->ev_pool) (void)hipEventDestroy(ev);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->d_frame) (void)hipFree(ctx->d_frame);
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    if (ctx->d_gram) (void)hipFree(ctx->d_gram);
    if (ctx->d_splitk) (void)hipFree(ctx->d_splitk);
    if (ctx->d_splitk_count) (void)hipFree(ctx->d_splitk_count);
    if (ctx->d_in) (void)hipFree(ctx->d_in);
    if (ctx->d_scalars) (void)hipFree(ctx->d_scalars);
    if (ctx->h_pinned_small) (void)hipHostFree(ctx->h_pinned_small);
        if (ctx->d_zeros) (void)hipFree(ctx->d_zeros);
        if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
    return HIVE_OK;
}

int hive_ctx_release_stream(hive_ctx *ctx) {
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    ctx->owns_stream = false;  // hive_ctx_destroy leaves the stream alive: somebody else (a torch ExternalStream, a caching allocator's recorded uses) refers to it
    return HIVE_OK;
}

int hive_ctx_set_stream(hive_ctx *ctx, void *stream) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, stream != HIVE_STREAM_OWN && stream != HIVE_STREAM_OWN_LOW, "hive_ctx_set_stream: pass a hipStream_t (NULL = the default stream)");
    hipStream_t next = (hipStream_t)stream;
    if (next == ctx->stream) return HIVE_OK;
    // work already queued on the old stream may still use the context's scratch buffers: order the new stream behind it
    hipEvent_t ev;
    HIVE_CHECK_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(next, ev, 0);
    (void)hipEventDestroy(ev);
    HIVE_CHECK_HIP(ctx, e);
    if (ctx->owns_stream) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamDestroy(ctx->stream);
        ctx->owns_stream = false;
    }
    ctx->stream = next;
    return HIVE_OK;
}

int hive_ctx_get_stream(hive_ctx *ctx, void **stream) {
    if (!ctx || !stream) return hive_fail(ctx, HIVE_ERR_INVALID, "hive_ctx_get_stream: NULL argument");
    *stream = (void *)ctx->stream;
    return HIVE_OK;
}

int hive_ctx_synchronize(hive_ctx *ctx) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HIVE_OK;
}

const char *hive_last_error(hive_ctx *ctx) { return ctx ? ctx->last_error.c_str() : g_global_error.c_str(); }

int hive_ctx_set_round_mode(hive_ctx *ctx, int mode) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, mode == HIVE_ROUND_HALF_EVEN || mode == HIVE_ROUND_HALF_AWAY, "round mode must be 0 or 1, got %d", mode);
    ctx->round_mode = mode;
    return HIVE_OK;
}

int hive_ctx_set_deterministic(hive_ctx *ctx, int enabled) {
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    ctx->deterministic = enabled != 0;
    return HIVE_OK;
}

int hive_ctx_launch_stats(hive_ctx *ctx, int64_t *splitk_launches, int64_t *deep_ring_launches, int reset) {
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    if (splitk_launches) *splitk_launches = ctx->n_splitk_launches;
    if (deep_ring_launches) *deep_ring_launches = ctx->n_deep_ring_launches;
    if (reset) ctx->n_splitk_launches = ctx->n_deep_ring_launches = 0;
    return HIVE_OK;
}

int hive_ctx_set_timing(hive_ctx *ctx, int enabled) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    ctx->timing = enabled != 0;
    ctx->ev_used = 0;
    ctx->last_start = ctx->last_stop = nullptr;
    return HIVE_OK;
}

int hive_ctx_last_kernel_ms(hive_ctx *ctx, float *ms) {
    HIVE_ENTER(ctx);
    if (!ctx || !ms) return hive_fail(ctx, HIVE_ERR_INVALID, "NULL argument");
    HIVE_REQUIRE(ctx, ctx->last_start && ctx->last_stop, "no timed kernel has been launched on this context");
    HIVE_CHECK_HIP(ctx, hipEventSynchronize(ctx->last_stop));
    HIVE_CHECK_HIP(ctx, hipEventElapsedTime(ms, ctx->last_start, ctx->last_stop));
    return HIVE_OK;
}

int hive_ctx_kernel_time_total(hive_ctx *ctx, int *n_launches, float *total_ms) {
    HIVE_ENTER(ctx);
    if (!ctx || !n_launches || !total_ms) return hive_fail(ctx, HIVE_ERR_INVALID, "NULL argument");
    HIVE_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float total = 0.f;
    int n = 0;
    for (size_t i = 0; i + 1 < ctx->ev_used; i += 2) {
        float ms = 0.f;
        HIVE_CHECK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_pool[i], ctx->ev_pool[i + 1]));
        total += ms;
        ++n;
    }
    *n_launches = n;
    *total_ms = total;
    ctx->ev_used = 0;
    return HIVE_OK;
}

}  // extern "C"
