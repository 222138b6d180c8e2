/*
 * hive_mi355x.h -- C ABI of libhive_mi355x.so: the MI355X (gfx950) implementation of HIVE's
 * per-frame dense-compute path (depth-map -> TSDF volume fusion, view frusta, depth-to-point
 * unprojection / projection, mask dilation, marching cubes, DPT ViT blocks).
 *
 * HIVE itself has no FFI: its boundary for this path is a set of plain Python call
 * signatures.  Every entry point below cites the reference call site
 * (file:line under /root/reference) whose arithmetic it replaces; INTEGRATION.md shows
 * the ctypes binding a HIVE maintainer would add.
 *
 * Conventions
 *   - every function returns 0 (HIVE_OK) or a negative hive_status; the message for the
 *     last failure on a context is returned by hive_last_error(ctx) (ctx may be NULL for
 *     failures of hive_ctx_create itself: thread-local).
 *   - plain pointers and sizes only; `mem` says whether image/point pointers are host
 *     (HIVE_MEM_HOST: copied through a pinned staging buffer, the call returns after the
 *     copy so the caller may overwrite its arrays -- hive/fusion.py:121 mutates depth_im
 *     right before integrate) or device memory (HIVE_MEM_DEVICE: used in place, stream
 *     ordered on the context's stream).
 *   - one hive_ctx per (GPU, stream); contexts are independent, so Python threads may own
 *     separate contexts (the reference calls the geometric functions from a ThreadPool,
 *     hive/pipeline.py:491).  A single context is not re-entrant.
 *   - volumes are three float32 arrays [X][Y][Z] (z fastest), as in the reference library:
 *     tsdf (init 1), weight (init 0), colour packed b*65536 + g*256 + r (init 0).
 *   - there is no CPU fallback: every compute entry point needs a gfx950 device.
 */
#ifndef HIVE_MI355X_H
#define HIVE_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIVE_ABI_VERSION 2

typedef enum hive_status {
    HIVE_OK = 0,
    HIVE_ERR_INVALID = -1,   /* bad argument (shape, NULL, size)                 */
    HIVE_ERR_DEVICE = -2,    /* HIP runtime failure; message has the hipError_t  */
    HIVE_ERR_NOMEM = -3,     /* device / pinned allocation failed                */
    HIVE_ERR_EMPTY = -4,     /* no surface: tsdf has no zero crossing (the reference raises
                                ValueError there, scripts/experiments.py:165-168) */
    HIVE_ERR_STATE = -5      /* call sequence error (e.g. copy_mesh before extract) */
} hive_status;

typedef enum hive_mem_kind { HIVE_MEM_HOST = 0, HIVE_MEM_DEVICE = 1 } hive_mem_kind;

/* Rounding of real-valued pixel coordinates / colours to integers.
 * HALF_EVEN = numpy's np.round (the reference library's CPU path and hive/geometric.py:176);
 * HALF_AWAY = C roundf (the reference library's CUDA kernel).  SURVEY.md §7(b). */
typedef enum hive_round_mode { HIVE_ROUND_HALF_EVEN = 0, HIVE_ROUND_HALF_AWAY = 1 } hive_round_mode;

typedef struct hive_ctx hive_ctx;
typedef struct hive_tsdf hive_tsdf;

/* ---- library / context ------------------------------------------------------------ */
int hive_abi_version(void);
/* `stream` is the hipStream_t every kernel and copy of the context is issued on, e.g.
 * torch.cuda.current_stream().cuda_stream.  NULL is HIP's default (null) stream -- which is what that
 * expression returns for PyTorch's default stream -- and HIVE_STREAM_OWN makes the context create a
 * non-blocking stream of its own. */
#define HIVE_STREAM_OWN ((void *)(intptr_t)-1)
/* ... and HIVE_STREAM_OWN_LOW one of the LOWEST dispatch priority (hipStreamCreateWithPriority): for work that should fill what the
 * other streams leave idle -- the TSDF sweeps of a batch under the next batch's network (hive_amd.depth.DepthFusionStream(overlap=True)). */
#define HIVE_STREAM_OWN_LOW ((void *)(intptr_t)-2)
int hive_ctx_create(int device_id, void *stream, hive_ctx **out);
int hive_ctx_destroy(hive_ctx *ctx);
/* The hipStream_t the context issues on (e.g. to wrap a context-owned stream in torch.cuda.ExternalStream). */
int hive_ctx_get_stream(hive_ctx *ctx, void **stream);
int hive_ctx_synchronize(hive_ctx *ctx);
const char *hive_last_error(hive_ctx *ctx);
/* Give up ownership of a private stream (HIVE_STREAM_OWN / _OWN_LOW): hive_ctx_destroy will then NOT destroy it.  For callers that hand the stream to a
 * framework which keeps referring to it after the context is gone -- PyTorch's caching allocator records an event on every stream a tensor was used on
 * (`Tensor.record_stream`) when the tensor is freed, possibly long after the context died, and process groups cache streams too; recording on a destroyed
 * stream crashes the process (found by tests/test_distributed_gpu.py::test_rccl_merge_of_a_volume_on_the_overlap_side_stream).  The Python binding calls
 * this as soon as it wraps the stream for torch (`Context.torch_stream()`): such a stream lives until the process ends. */
int hive_ctx_release_stream(hive_ctx *ctx);
/* Re-binds the context to another hipStream_t (NULL = the default stream).  The new stream is ordered behind
 * everything already queued on the old one.  The Python binding calls this when torch's current stream changes
 * (e.g. inside `with torch.cuda.stream(s)`), so that hive kernels stay ordered with the surrounding torch ops. */
int hive_ctx_set_stream(hive_ctx *ctx, void *stream);
/* Rounding used by hive_project / hive_project_bbox, and the mode a volume created afterwards on this context starts with. */
int hive_ctx_set_round_mode(hive_ctx *ctx, int mode);
/* Deterministic mode (off by default): the network kernels stop using the three paths whose USE depends on the batch size and on the device's CU count --
 * split-K of long K loops at small launches (another, fixed, order of float32 additions), the attention kernel's two-way split of the keys at small grids
 * (likewise) and the Gram-matrix form of the GroupNorm statistics of the bottlenecks' 1 x 1 convolutions (statistics of the exact products instead of the
 * rounded outputs) -- so that a frame's depth map no longer changes
 * with the size of the batch it was in beyond what is stated next.  What remains: a GroupNorm's per-tile partial sums are cut at 128- / 256-pixel
 * tile boundaries counted from the START OF THE BATCH, so a frame's (mean, rstd) still depend on its position in the batch by float32 re-association
 * (~1e-7 relative).  Frame-sharded multi-GPU runs that want one-GPU depth maps to the last bit must also keep the per-rank batch composition. */
int hive_ctx_set_deterministic(hive_ctx *ctx, int enabled);
/* Launch counters of the small-launch paths since creation (or the last call with reset != 0): GEMM / convolution launches that split their K loop,
 * and launches of the four-stage-ring kernels (csrc/mfma_pipe.hpp).  Diagnostic: the one-frame parity tests assert with it that those paths ran. */
int hive_ctx_launch_stats(hive_ctx *ctx, int64_t *splitk_launches, int64_t *deep_ring_launches, int reset);
/* HIP-event timing of the most recent kernel launched by a *_timed call on this context. */
int hive_ctx_set_timing(hive_ctx *ctx, int enabled);
int hive_ctx_last_kernel_ms(hive_ctx *ctx, float *ms);
/* Sum and count of the HIP-event durations of every dominant-kernel launch (integrate) made on
 * this context since timing was enabled or since the previous call; synchronises the stream. */
int hive_ctx_kernel_time_total(hive_ctx *ctx, int *n_launches, float *total_ms);

/* ---- TSDF volume: replaces third_party/tsdf_fusion_python `fusion.TSDFVolume` ------ */
/* fusion.TSDFVolume(vol_bnds, voxel_size)           -- hive/fusion.py:104
 * vol_bnds is row-major [3][2] = {xmin,xmax, ymin,ymax, zmin,zmax} (float64, as hive/fusion.py:48).
 * vol_dim = ceil((max-min)/voxel_size); origin = float32(min); trunc = 5*voxel_size.
 * If d_tsdf/d_weight/d_color are non-NULL the volume lives in caller-owned device memory
 * (e.g. torch tensors) of vol_dim[0]*vol_dim[1]*vol_dim[2] floats each; otherwise the
 * library allocates.  The volume is initialised (1,0,0) either way. */
int hive_tsdf_dims(const double vol_bnds[6], double voxel_size, int64_t vol_dim[3]);
int hive_tsdf_create(hive_ctx *ctx, const double vol_bnds[6], double voxel_size,
                     float *d_tsdf, float *d_weight, float *d_color, hive_tsdf **out);
/* An x-slab of the same scene grid: the volume holds grid voxels x_begin <= x < x_end (all y, z), dims (x_end - x_begin, Y, Z),
 * and every voxel keeps the world position it has in the whole grid (origin + index * voxel_size with the GRID index, bit for
 * bit) -- so integrating a frame into the slabs of a partition gives exactly the slices of integrating it into the whole
 * volume.  For the bit-exact multi-GPU mode (frames all-gathered, volume sharded in x-slabs).  x_end < 0 = the
 * whole grid.  hive_tsdf_info reports the slab's dims and the grid's origin / bounds; extract_mesh needs a whole volume. */
int hive_tsdf_create_slab(hive_ctx *ctx, const double vol_bnds[6], double voxel_size, int64_t x_begin, int64_t x_end,
                          float *d_tsdf, float *d_weight, float *d_color, hive_tsdf **out);
int hive_tsdf_slab_info(hive_tsdf *vol, int64_t *x_begin, int64_t *grid_dim_x);
int hive_tsdf_destroy(hive_tsdf *vol);
/* Rounding of THIS volume's pixel projection and colour average (integrate, accum_finalize): the reference library has two
 * arithmetic paths -- its CUDA kernel rounds with roundf (HIVE_ROUND_HALF_AWAY; what `TSDFVolume(..., use_gpu=True)` runs when
 * pycuda is installed, as in the reference's Docker image, requirements.txt:14) and its numpy path with np.round
 * (HIVE_ROUND_HALF_EVEN; `use_gpu=False`, BASELINE config 1). */
int hive_tsdf_set_round_mode(hive_tsdf *vol, int mode);
int hive_tsdf_reset(hive_tsdf *vol);
/* Tell the library that the caller wrote the volume's planes itself (caller-owned storage of hive_tsdf_create): cached mesh results
 * and the fast paths that rely on what the library knows about the planes' contents (every weight a whole number of unit
 * observations since the last reset -- the division-free colour update of the sweep) are dropped until the next hive_tsdf_reset. */
int hive_tsdf_planes_modified(hive_tsdf *vol);
int hive_tsdf_info(hive_tsdf *vol, int64_t vol_dim[3], float origin[3], double vol_bnds[6],
                   float *voxel_size, float *trunc_margin);
/* device pointers of the three volumes (for RCCL collectives / zero-copy views) */
int hive_tsdf_device_ptrs(hive_tsdf *vol, float **d_tsdf, float **d_weight, float **d_color);

/* TSDFVolume.integrate(color_im, depth_im, cam_intr, cam_pose, obs_weight) -- hive/fusion.py:124
 * color u8 [H][W][3] RGB, depth f32 [H][W] metres (0 = invalid), K f32 row-major 3x3,
 * cam_pose f64 row-major 4x4 camera-to-world.  n_updated (optional, host) receives the number
 * of voxels written by this call (forces a stream sync). */
int hive_tsdf_integrate(hive_tsdf *vol, const uint8_t *color, const float *depth, int H, int W,
                        const float K[9], const double cam_pose[16], float obs_weight,
                        int mem, uint64_t *n_updated);
/* n frames back to back (frame f at color + f*H*W*3, depth + f*H*W, poses + f*16), in order:
 * identical results (bit for bit) to n calls of hive_tsdf_integrate.  Device-resident frames are swept up to four
 * consecutive frames per launch: a voxel is loaded once, the frames are applied to it in order, it is stored once. */
int hive_tsdf_integrate_batch(hive_tsdf *vol, int n, const uint8_t *color, const float *depth,
                              int H, int W, const float K[9], const double *cam_poses,
                              float obs_weight, int mem);
/* How the most recent hive_tsdf_integrate_batch on this volume grouped its frames: *n_groups launches, sizes[i] frames in
 * sweep i (1 = the single-frame kernel), in order; at most `capacity` sizes are written.  Diagnostic: the parity tests assert
 * with it that the fused sweep really ran (the semantics are those of hive/fusion.py:113-124's serial loop either way). */
int hive_tsdf_last_batch_groups(hive_tsdf *vol, int *sizes, int capacity, int *n_groups);
/* Length of the work list the most recent sweep on this volume ran over (segments of *segment_voxels consecutive z voxels
 * that survive the per-row frustum / depth clip); forces a stream sync.  Diagnostic: n_items * segment_voxels against the voxels
 * a sweep updates is the share of its arithmetic that can change a voxel (bench.py reports it). */
int hive_tsdf_last_sweep_items(hive_tsdf *vol, uint64_t *n_items, int *segment_voxels);
/* What the volume has been given since its creation / last hive_tsdf_reset: frames handed to integrate / integrate_batch /
 * accum_integrate (`frames`) and the integrate launches that applied them (`launches`: one per single-frame kernel, one per fused
 * sweep).  Host-side bookkeeping, no stream sync.  bench.py prints it next to the weight plane's sum as the timed job's proof of work
 * (one TSDFVolume.integrate call of hive/fusion.py:124 = one frame here). */
int hive_tsdf_stats(hive_tsdf *vol, int64_t *frames, int64_t *launches);
/* TSDFVolume.get_volume(): copies tsdf and colour (and weight) out (any may be NULL).  The destinations / sources of
 * get_volume / set_volume may be host OR device memory (unified addressing decides the copy direction). */
int hive_tsdf_get_volume(hive_tsdf *vol, float *h_tsdf, float *h_color, float *h_weight);
int hive_tsdf_set_volume(hive_tsdf *vol, const float *h_tsdf, const float *h_color, const float *h_weight);
/* Overwrite voxels [first, first + count) of the three planes from device (or host) arrays of `count` floats each (any may be
 * NULL): how the pieces of an all-gather (a rank's share of the merged volume, or an x-slab) are put in place. */
int hive_tsdf_set_volume_range(hive_tsdf *vol, int64_t first, int64_t count, const float *d_tsdf, const float *d_weight, const float *d_color);

/* TSDFVolume.get_mesh() -- hive/fusion.py:127.  Two steps because the sizes are data dependent:
 * extract counts and builds the mesh on the device, copy_mesh copies it out.
 * verts f32 [nv][3] world coordinates, faces i32 [nf][3], norms f32 [nv][3], colors u8 [nv][3] (RGB).
 * Vertex order: ascending (voxel linear index, axis); face order: ascending cell linear index,
 * then table order.  Returns HIVE_ERR_EMPTY when there is no zero crossing. */
int hive_tsdf_extract_mesh(hive_tsdf *vol, int64_t *n_verts, int64_t *n_faces);
int hive_tsdf_copy_mesh(hive_tsdf *vol, float *verts, int32_t *faces, float *norms, uint8_t *colors);
/* vertices of the same extraction in voxel units (before x voxel_size + origin), f32 [nv][3] */
int hive_tsdf_copy_mesh_voxel_coords(hive_tsdf *vol, float *verts_vox);

/* Alternative to accum_integrate for a rank that fused its own frames with hive_tsdf_integrate: convert its
 * running-average volumes into the same sums, planes = [tsdf * w, w, r * w, g * w, b * w], ready for the all-reduce.
 * (The per-rank colours are already rounded per frame, so the merged colour can differ from the sequential one by the
 * same +-2 levels as with accum_integrate; tsdf and weight merge exactly up to float re-association.) */
int hive_tsdf_accum_from_volume(hive_tsdf *vol, float *d_accum);
/* The same sums laid out for ONE reduce-scatter over `world` ranks: d_out is float [world][5][chunk]; voxel i goes to piece
 * i / chunk, offset i % chunk (world * chunk >= N; the tail past N is written as zeros).  Rank r's reduce-scatter output is then
 * its [5][chunk] share, the input of hive_tsdf_accum_finalize_to with plane_stride = chunk. */
int hive_tsdf_accum_from_volume_sharded(hive_tsdf *vol, float *d_out, int world, int64_t chunk);
/* Frame-sharded fusion (BASELINE.json north_star): a rank accumulates
 * num = sum(w_i*dist_i), w = sum(w_i), rgb = sum(w_i*c_i) for its frames into 5 float planes
 * [5][X][Y][Z]; planes are summed across ranks (RCCL all-reduce by the caller), then folded
 * into the volume.  d_accum must hold 5*N floats of device memory, zeroed by accum_reset. */
int hive_tsdf_accum_reset(hive_tsdf *vol, float *d_accum);
int hive_tsdf_accum_integrate(hive_tsdf *vol, float *d_accum, const uint8_t *color, const float *depth,
                              int H, int W, const float K[9], const double cam_pose[16],
                              float obs_weight, int mem);
int hive_tsdf_accum_finalize(hive_tsdf *vol, const float *d_accum);
/* The same arithmetic (and the volume's round mode) for `count` voxels of 5 planes of `plane_stride` floats each, into explicit
 * device arrays d_tsdf / d_weight / d_color [count]: what a rank runs on ITS share of the voxels after a reduce-scatter of the
 * planes; the three result arrays are then all-gathered (hive_amd.distributed.fuse_sharded). */
int hive_tsdf_accum_finalize_to(hive_tsdf *vol, const float *d_accum, int64_t plane_stride, int64_t count, float *d_tsdf,
                                float *d_weight, float *d_color);

/* ---- fusion.get_view_frustum(depth_im, cam_intr, cam_pose) -- hive/fusion.py:59 ------ */
/* out: float64 row-major [3][5] (apex + 4 corners at max(depth)), world coordinates */
int hive_view_frustum(hive_ctx *ctx, const float *depth, int H, int W, const float K[9],
                      const double cam_pose[16], int mem, double out[15]);

/* The bounds pass of adjust_voxel_size (hive/fusion.py:53-61) for a whole frame set at once: n depth maps [n][H][W] (already in HBM
 * when mem = HIVE_MEM_DEVICE), poses f64 [n][16] -> out f64 [n][3][5]; ONE reduction launch and ONE read-back of n maxima instead
 * of n launches + n synchronisations.  Same values as n calls of hive_view_frustum. */
int hive_view_frustum_batch(hive_ctx *ctx, const float *depth, int n, int H, int W, const float K[9],
                            const double *cam_poses, int mem, double *out);

/* ---- hive/geometric.py ------------------------------------------------------------- */
/* point_cloud_from_depth(depth, mask, K, R, t)  -- hive/geometric.py:107-126 (+ image2world :183-206)
 * valid = mask & (depth > 0); points in row-major (v,u) order; X = R^T (d * Kinv [u,v,1]^T - t).
 * Kinv = np.linalg.inv(K) evaluated by the caller in K's dtype (geometric.py:203) and widened to f64.
 * mask u8 [H][W] (non-zero = keep; NULL = all), R f64[9], t f64[3] (world-to-camera),
 * out_xyz f64 [capacity][3] (host or device per `mem`), *n = number of points written.
 * rgb (optional u8 [H][W][3]) / out_rgba (u8 [capacity][4], alpha 255): point_cloud_from_rgbd :129-152 */
int hive_unproject(hive_ctx *ctx, const float *depth, const uint8_t *mask, const uint8_t *rgb,
                   int H, int W, const double Kinv[9], const double R[9], const double t[3],
                   int mem, double *out_xyz, uint8_t *out_rgba, int64_t capacity, int64_t *n);
/* image2world(points, depth, K, R, t, scale_factor) -- hive/geometric.py:183-206, for an explicit
 * list of pixel coordinates: uv f64 [n][2], depth f64 [n] -> out_xyz f64 [n][3] */
int hive_image2world(hive_ctx *ctx, const double *uv, const double *depth, int64_t n, const double Kinv[9],
                     const double R[9], const double t[3], double scale_factor, int mem, double *out_xyz);
/* world2image(points, K, R, t, scale_factor, dtype) -- hive/geometric.py:155-180
 * points f64 [n][3]; out_uv_i32 (rounded per ctx round mode, default half-even = np.round) or
 * out_uv_f64 (exactly one non-NULL); out_depth f64 [n] */
int hive_project(hive_ctx *ctx, const double *points, int64_t n, const double K[9], const double R[9],
                 const double t[3], double scale_factor, int mem,
                 int32_t *out_uv_i32, double *out_uv_f64, double *out_depth);

/* world2image + the visibility reduction of HiveDataset.select_key_frames -- hive/io.py:1161-1175:
 * project points (f64 [n][3]), round half-to-even, keep pixels inside [0,W) x [0,H) and return
 * out = {min u, max u, min v, max v, count} (count == 0: the extrema are INT_MAX / INT_MIN). */
int hive_project_bbox(hive_ctx *ctx, const double *points, int64_t n, const double K[9], const double R[9],
                      const double t[3], int W, int H, int mem, int32_t out[5]);

/* ---- foreground per-frame meshes: Pipeline._create_scene's inner loops -- hive/pipeline.py:340-483 ------------------ */
/* _triangulate_faces (:651-667) + _filter_faces (:670-694) in one pass over the frame: the valid pixels (mask & depth > 0) of
 * a depth map are a lattice, so the triangulation is the implicit one of the pixel grid (two triangles per fully valid 2 x 2
 * block, one per block with three valid corners) and only faces whose edges are all <= max_pixel_distance pixels long and span
 * <= max_depth_distance metres are emitted (hive/options.py:271-286: defaults 2 and 0.1).  out_faces i32 [capacity][3] index the
 * rows of hive_unproject's output for the same depth / mask (valid pixels in row-major order), wound like the reference's
 * reversed Delaunay simplices; face order: row-major by block.  *n_faces = number of faces found (may exceed capacity: call
 * with capacity 0 to size the buffer); *n_vertices (optional) = number of valid pixels. */
int hive_grid_mesh(hive_ctx *ctx, const float *depth, const uint8_t *mask, int H, int W, double max_pixel_distance,
                   double max_depth_distance, int mem, int32_t *out_faces, int64_t capacity, int64_t *n_faces,
                   int64_t *n_vertices);
/* One object of one frame in ONE call (the body of process_frame's loop, hive/pipeline.py:383-461, without the CPU-library stages between -- decimation, connected
 * components, billboard): point_cloud_from_depth (:386; hive_unproject's rows) -> _triangulate_faces + _filter_faces (:402-408; hive_grid_mesh's faces) ->
 * _get_mesh_texture_and_uv (:453; hive_texture_window's uv and crop box), everything device-resident: d_depth f32 [H][W], d_mask u8 [H][W] (NULL = all pixels) in;
 * d_vertices f64 [vertex_capacity][3], d_faces i32 [face_capacity][3], d_uv i32 [vertex_capacity][2] out (H W and 4 H W are always enough).  Kinv = inverse
 * intrinsics as the caller computes it (numpy.linalg.inv, as hive/geometric.py:201), K, R, t the camera as for hive_unproject / hive_project.  Seven launches and
 * ONE read-back (through pinned memory): *n_vertices, *n_faces, bbox = {min_u, min_v, max_u + 1, max_v + 1} (the texture is image[min_v:max_v, min_u:max_u]).
 * Results are bit-identical to the three separate entry points on the same inputs. */
int hive_fg_frame_mesh(hive_ctx *ctx, const float *d_depth, const uint8_t *d_mask, int H, int W, const double Kinv[9], const double K[9], const double R[9],
                       const double t[3], double max_pixel_distance, double max_depth_distance, double *d_vertices, int64_t vertex_capacity, int32_t *d_faces,
                       int64_t face_capacity, int32_t *d_uv, int64_t *n_vertices, int64_t *n_faces, int32_t bbox[4]);
/* _filter_faces (:670-694) for an explicit face list of any triangulation: points2d i32 [n][2] (u, v), depth f32 [n], faces i32
 * [F][3] -> the faces whose three edges pass both limits, order preserved, into out_faces (capacity F). */
int hive_filter_faces(hive_ctx *ctx, const int32_t *points2d, const float *depth, int64_t n_points, const int32_t *faces,
                      int64_t n_faces_in, double max_pixel_distance, double max_depth_distance, int mem, int32_t *out_faces,
                      int64_t *n_faces_out);
/* _get_mesh_texture_and_uv (:782-808): uv = world2image(vertices) in its default int32 form (np.round); bbox = {min_u, min_v,
 * max_u + 1, max_v + 1} -- the crop `image[min_v:max_v, min_u:max_u]` that becomes the texture; out_uv i32 [n][2] = uv - (min_u, min_v). */
int hive_texture_window(hive_ctx *ctx, const double *points, int64_t n, const double K[9], const double R[9], const double t[3],
                        double scale_factor, int mem, int32_t *out_uv, int32_t bbox[4]);

/* ---- dilate_mask(mask, MaskDilationOptions(num_iterations)) -- hive/image_processing.py:30-45 */
/* 3x3 rectangular structuring element applied `iterations` times == one (2*it+1)^2 box max
 * (cv2.dilate border = no contribution from outside).  mask/out u8 [H][W], non-zero = set. */
int hive_dilate_mask(hive_ctx *ctx, const uint8_t *mask, int H, int W, int iterations, int mem,
                     uint8_t *out);

/* The same with the caller's structuring element (hive/options.py:245-268: `MaskDilationOptions(num_iterations, dilation_filter)`):
 * se u8 [kh][kw] (host memory, 1x1 .. 32x32, non-zero = member, at least one member), cv2.dilate's definition -- anchor at
 * (kw / 2, kh / 2), taps outside the image ignored, the pass repeated `iterations` times (0 = copy).  A full rectangle of odd sides
 * takes the separable box-max path; anything else is iterated literally. */
int hive_dilate_mask_se(hive_ctx *ctx, const uint8_t *mask, int H, int W, const uint8_t *se, int kh, int kw, int iterations, int mem,
                        uint8_t *out);

/* Masking of the depth maps of a frame set on the device (all pointers device memory; d_out may alias d_depth):
 *   mode 0 -- background volume, hive/fusion.py:118-121: `mask = dilate_mask(mask, iterations); depth[mask > 0] = 0`;
 *   mode 1 -- foreground volume (BASELINE config 5: the complement): depth is kept where the UNDILATED mask is set, 0 elsewhere.
 * d_mask u8 [n][H][W] instance ids (hive/io.py:214-218: 0 = background, 1..k = objects); instance_id > 0 restricts "set" to that
 * object, 0 = any object. */
int hive_depth_apply_mask_se(hive_ctx *ctx, const float *d_depth, const uint8_t *d_mask, int n, int H, int W, const uint8_t *se, int kh,
                             int kw, int iterations, int mode, int instance_id, float *d_out); /* the same, any structuring element (host) */
int hive_depth_apply_mask(hive_ctx *ctx, const float *d_depth, const uint8_t *d_mask, int n, int H, int W, int iterations,
                          int mode, int instance_id, float *d_out);
/* The loader's depth transform on the device (hive/io.py:1032-1039): uint16 millimetres -> float32 metres
 * (`depth_scale * mm`), `> max_depth -> 0`. */
int hive_depth_mm_to_m(hive_ctx *ctx, const uint16_t *d_mm, int64_t n, float depth_scale, float max_depth, float *d_out);

/* ---- depth hand-off DPT -> TSDF ----------------------------------------------------- */
/* dataset_adaptors.py:1432-1433 (x1000 -> uint16 truncation) then io.py:1032-1039
 * (x 1/1000 as float32, > max_depth -> 0), optional mask (non-zero -> depth 0, fusion.py:121).
 * in: f32 or f16/bf16 depth in metres [H][W] on the device; out_mm (optional u16) and out_m (f32). */
typedef enum hive_dtype { HIVE_F32 = 0, HIVE_F16 = 1, HIVE_BF16 = 2 } hive_dtype;
int hive_depth_quantize(hive_ctx *ctx, const void *d_depth, int dtype, int H, int W,
                        float depth_scale, float max_depth, const uint8_t *d_mask,
                        uint16_t *d_out_mm, float *d_out_m);

/* ---- DPT ViT encoder blocks: replaces the transformer blocks of dpt.models.DPTDepthModel.forward
 *      (timm vit_base_resnet50_384) -- hive/dataset_adaptors.py:1366-1374,1419 --------------------- */
/* All pointers are device memory.  Activations / weights are the 16-bit type `dtype` names -- HIVE_BF16, or HIVE_F16: what the
 * reference runs (`model.half()` and fp16 samples, hive/dataset_adaptors.py:1394-1401, 1415-1417); same kernels, same rates
 * (v_mfma_f32_16x16x32_{bf16,f16}) -- with [out][in] row-major weights, as nn.Linear stores them; biases and LayerNorm affine
 * parameters are float32.  Below "16-bit" stands for that type. */
typedef struct hive_vit hive_vit;
typedef struct hive_vit_block_weights {
    const void *ln1_g, *ln1_b;   /* f32 [D]            norm1            */
    const void *qkv_w, *qkv_b;   /* 16-bit [3D][D], f32 [3D] attn.qkv   */
    const void *proj_w, *proj_b; /* 16-bit [D][D], f32 [D]   attn.proj  */
    const void *ln2_g, *ln2_b;   /* f32 [D]            norm2            */
    const void *fc1_w, *fc1_b;   /* 16-bit [F][D], f32 [F]   mlp.fc1    */
    const void *fc2_w, *fc2_b;   /* 16-bit [D][F], f32 [D]   mlp.fc2    */
} hive_vit_block_weights;
/* head dim must be 64 (D = 64 * heads), D a multiple of 256 and <= 1024, F a multiple of 128. */
int hive_vit_create(hive_ctx *ctx, int dtype, int depth, int dim, int heads, int mlp_dim, float ln_eps,
                    const hive_vit_block_weights *blocks, hive_vit **out);
/* WEIGHT SNAPSHOT CONTRACT.  hive_vit_create copies what the folded LayerNorm needs -- gamma o W of qkv / fc1 and the constant rows c1 = sum_k (gamma o W)[n][k],
 * c2 = sum_k beta[k] W[n][k] + bias[n] -- into private buffers; proj / fc2 and every bias other than those two are read through the caller's pointers at
 * every forward.  A caller that overwrites weights IN PLACE after create must call hive_vit_weights_modified (re-folds on the context's stream, ordered
 * behind the writes there) before the next forward, or gets old LayerNorm / qkv / fc1 with new proj / fc2.  (hive_dpt: hive_dpt_weights_modified.) */
int hive_vit_weights_modified(hive_vit *vit);
int hive_vit_destroy(hive_vit *vit);
/* x 16-bit [B][N][D] -> runs all blocks; after block tap_blocks[t] its output is copied to tap_out[t]
 * (16-bit [B][N][D]).  x = x + proj(attn(LN1 x)); x = x + fc2(gelu(fc1(LN2 x))). */
int hive_vit_forward(hive_vit *vit, const void *x, int B, int N, const int *tap_blocks, int n_taps,
                     void *const *tap_out);
/* the individual kernels (used by hive_vit_forward; exposed for the numerics tests) */
int hive_vit_layernorm(hive_ctx *ctx, const void *x, int dtype, const float *gamma, const float *beta, void *out,
                       int M, int D, float eps);
/* C = epilogue(A[M][K] W[N][K]^T + bias): epilogue 0 = none, 1 = GELU (erf), 2 = + residual[M][N] */
int hive_vit_linear(hive_ctx *ctx, const void *A, int dtype, const void *W, const float *bias, const void *residual,
                    void *C, int M, int N, int K, int epilogue);
/* qkv projection of x [B*Np][D] (Np a multiple of 64): q|k -> qk [B*Np][2D], v -> vT [B][H][64][Np].  Both are operands
 * private to hive_vit_attention: q is stored multiplied by head_dim^-0.5 * log2(e) (in f32, before the one rounding to 16 bits), so
 * the attention's score products are base-2 exponents; vT is laid out for its fragment reads: along its last axis the token quads 4..7 and 8..11 of every group of 16 are swapped
 * (token t is stored at t with bits 2 and 3 exchanged), which makes the attention's V fragments single 16-byte LDS reads. */
int hive_vit_qkv(hive_ctx *ctx, const void *x, int dtype, const void *W, const float *bias, void *qk, void *vT,
                 int B, int Np, int D, int H);
/* softmax(q k^T / 8) v over the first N keys of each image -> out [B*Np][D]; Np = N rounded up to a multiple of 64; qk, vT as
 * hive_vit_qkv writes them (q pre-scaled: the kernel evaluates exp2(q' k^T - max)).  Where the launch leaves a CU one workgroup at most (one or two 480 x 640
 * frames) each workgroup splits its keys over two groups of waves and merges the halves: another order of float32 additions than at larger batches
 * (hive_ctx_set_deterministic keeps the one-chain form everywhere). */
int hive_vit_attention(hive_ctx *ctx, const void *qk, int dtype, const void *vT, void *out, int B, int N, int Np,
                       int D, int H);

/* ---- DPT pre/post-processing around the network (device pointers) -------------------------------- */
/* uint8 RGB values -> ((x / 255) - mean) / std as f16 / bf16, same element order (a [B][H][W][3] frame
 * batch becomes the channels-last network input) -- hive/dataset_adaptors.py:1407-1417 */
int hive_dpt_preprocess(hive_ctx *ctx, const uint8_t *d_rgb, int64_t n_values, float mean, float std,
                        int dtype, void *d_out);
/* The same for frames that are not the network's size -- hive/dataset_adaptors.py:1376-1389: `dpt.transforms.Resize(net_w, net_h, ...,
 * image_interpolation_method=cv2.INTER_CUBIC)` applied to `image / 255.0`, then NormalizeImage / PrepareForNet / the 16-bit cast:
 * d_rgb u8 [B][H][W][3] -> d_out [B][out_h][out_w][3] in `dtype` (HIVE_F16 / HIVE_BF16; HIVE_F32 for checks).  The output size is the
 * caller's (hive_amd.dpt.transforms.Resize.get_size restates the reference's sizing rule); the arithmetic is cv2.resize's INTER_CUBIC:
 * a = -0.75 cubic, four taps per axis at floor((d + 0.5) * src / dst - 0.5) - 1 .. + 2, indices clamped to the image, no anti-aliasing,
 * float32 weights, rows first.  (cv2 is not available to this build: parity with cv2 itself is unpinned; see csrc/resize.hip.) */
int hive_dpt_resize_preprocess(hive_ctx *ctx, const uint8_t *d_rgb, int B, int H, int W, int out_h, int out_w, float mean, float std,
                               int dtype, void *d_out);
/* hive/dataset_adaptors.py:1421-1426 `torch.nn.functional.interpolate(prediction, size=frame.shape[:2], mode="nearest")` fused with the
 * hand-off of :1432-1433 / hive/io.py:1032-1039: d_depth f32 [B][h][w] -> [B][H][W] outputs (any may be NULL, not all): d_out_depth f32,
 * d_out_mm = uint16(depth * 1000), d_out_m = depth_scale * mm with > max_depth -> 0.  Source index = min((int)floorf(dst * (float)src / dst_size), src - 1). */
int hive_depth_resize_nearest(hive_ctx *ctx, const float *d_depth, int B, int h, int w, int H, int W, float depth_scale, float max_depth,
                              float *d_out_depth, uint16_t *d_out_mm, float *d_out_m);
/* Last layer of the depth head, fused, in float32: 1x1 conv C -> 1 on the channels-last f16 / bf16 map
 * d_feat [n_px][C] (weights / bias on the host), ReLU if non_negative, depth = 1 / max(scale x + shift, 1e-8)
 * if invert (DPTDepthModel.forward).  h_pre_bias (optional, host, [C]) and pre_relu apply the bias and ReLU of the
 * convolution that produced d_feat on the fly (head: conv 128->32 + bias, ReLU, conv 32->1), saving two passes over
 * the largest activation of the network.  Optional outputs: d_depth f32 metres; and the PNG hand-off
 * d_out_mm = uint16(depth * 1000), d_out_m = depth_scale * mm with > max_depth -> 0
 * (hive/dataset_adaptors.py:1432-1433, hive/io.py:1032-1039). */
int hive_dpt_head_tail(hive_ctx *ctx, const void *d_feat, int dtype, int64_t n_px, int C,
                       const float *h_pre_bias, int pre_relu, const float *h_weight, float bias,
                       int non_negative, int invert, float scale, float shift, float *d_depth,
                       float depth_scale, float max_depth, uint16_t *d_out_mm, float *d_out_m);
/* The second half of the depth head as one kernel (f16 / bf16): Interpolate(x2, bilinear, align_corners=True) ->
 * Conv3x3(C_in=128 -> C_mid=32) + bias -> ReLU -> Conv1x1(32 -> 1) + bias -> [ReLU] -> [1 / max(scale x + shift, 1e-8)]
 * -> optional uint16-mm hand-off, i.e. scratch.output_conv[1:] of DPTDepthModel plus the tail above
 * (hive/dataset_adaptors.py:1419, 1432-1433).  d_x: channels-last [N][H][W][C_in] (output of output_conv[0]);
 * d_b0 (optional, device, f32 [C_in]): bias of output_conv[0], added to d_x on load (x + b rounded to the tensor dtype first, as the
 * separate bias add rounds it), so that convolution can run without its bias pass;
 * d_w3: the 3x3 weights on the device as [ky][kx][C_mid][C_in]; h_b3 [C_mid], h_w1 [C_mid] on the host.
 * Outputs are [N][2H][2W]; any of them may be NULL (not all). */
int hive_dpt_head_fused(hive_ctx *ctx, const void *d_x, const float *d_b0, int dtype, int N, int H, int W, int C_in, int C_mid,
                        const void *d_w3, const float *h_b3, const float *h_w1, float b1, int non_negative, int invert,
                        float scale, float shift, float *d_depth, float depth_scale, float max_depth,
                        uint16_t *d_out_mm, float *d_out_m);

/* ---- fused channels-last glue of the DPT convolutional parts (device pointers, f16 / bf16) ----------- */
/* GroupNorm over [N][HW][C] (C a power of two, 8..2048) with G groups, affine gamma / beta [C] in the tensor
 * dtype, optional residual add (same shape) and ReLU:  out = relu?( gn(x) (+ residual) ).  timm GroupNormAct +
 * the bottleneck's `relu(norm3(x) + shortcut)`, reached from DPTDepthModel.forward (dataset_adaptors.py:1419). */
int hive_nhwc_group_norm(hive_ctx *ctx, const void *d_x, int dtype, int N, int HW, int C, int G,
                         const void *d_gamma, const void *d_beta, float eps, const void *d_residual,
                         int relu, void *d_out);
/* out = relu?( (x + bias[c]) (+ residual) (+ residual2) ) over [n_px][C] (C % 8 == 0; bias in the tensor dtype; out may
 * alias x): convolution bias + ReLU + skip adds of the RefineNet residual units / fusion blocks (isl-org/DPT
 * ResidualConvUnit_custom, FeatureFusionBlock_custom) in one pass.  d_out_relu (optional) additionally receives
 * relu(out): the input of the next residual unit's first convolution (its `relu(x)`), saving that unit a pass. */
int hive_nhwc_bias_act(hive_ctx *ctx, const void *d_x, int dtype, int64_t n_px, int C, const void *d_bias, int relu,
                       const void *d_residual, const void *d_residual2, void *d_out, void *d_out_relu);
/* bilinear x2, align_corners=True: [N][H][W][C] -> [N][2H][2W][C] (C % 8 == 0): the RefineNet fusion blocks'
 * and the depth head's `interpolate(scale_factor=2, mode="bilinear", align_corners=True)`.  d_bias (optional, [C], tensor
 * dtype) is added to the input on load, rounded to the tensor dtype first: the bias pass of the producing convolution. */
int hive_nhwc_upsample2x(hive_ctx *ctx, const void *d_in, const void *d_bias, int dtype, int N, int H, int W, int C, void *d_out);

/* 3 x 3 convolution, stride 1, padding 1, channels-last f16 / bf16, as an implicit GEMM on the matrix cores with the decoder's
 * element-wise tail fused:  out = relu?( conv(x, w) (+ bias) (+ residual) (+ residual2) ), and optionally out_relu = relu(out).
 * The 3 x 3 convolutions of isl-org/DPT's decoder reached from DPTDepthModel.forward (hive/dataset_adaptors.py:1419):
 * scratch.layer{1..4}_rn, ResidualConvUnit_custom.conv1 / conv2, scratch.output_conv[0].
 * d_x [N][H][W][C_in]; d_w [C_out][3][3][C_in] (= the [C_out][C_in][3][3] weight tensor in channels-last memory format);
 * d_bias [C_out] in the tensor dtype or NULL; residuals / outputs [N][H][W][C_out].  C_in % 64 == 0, C_out % 128 == 0;
 * the outputs must not alias the input. */
int hive_nhwc_conv3x3(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, int C_in, int C_out, const void *d_w,
                      const void *d_bias, int relu, const void *d_residual, const void *d_residual2, void *d_out,
                      void *d_out_relu);

/* The same implicit-GEMM kernel family for the other convolutions of the network (ResNetV2-50 stages of the hybrid backbone --
 * timm StdConv2dSame with its "SAME" padding --, the 1 x 1 projections of the reassemble stages and fusion blocks, the stride-2
 * 3 x 3 of act_postprocess4): kernel 1 or 3 (square), stride 1 or 2, explicit top / left padding (the bottom / right padding
 * is whatever H_out, W_out imply), C_in % 64 == 0, C_out % 64 == 0, d_w [C_out][kernel][kernel][C_in].  Same fused epilogue.
 * Input pixel of tap (ky, kx) for output (oy, ox): (oy * stride + ky - pad_top, ox * stride + kx - pad_left), zero outside. */
int hive_nhwc_conv(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, int C_in, int C_out, int kernel, int stride,
                   int pad_top, int pad_left, int H_out, int W_out, const void *d_w, const void *d_bias, int relu,
                   const void *d_residual, const void *d_residual2, void *d_out, void *d_out_relu);

/* The same convolution with the statistics pass of the GroupNorm that follows it (timm ResNetV2: `norm(conv(x))` behind every
 * convolution of the hybrid backbone) folded into the epilogue: per tile of *gn_tile_rows output pixels (all C_out channels) the sum
 * and the sum of squares of the stored outputs, per channel, split between the two samples the tile may touch, are left in
 * d_gn_partial (float, >= hive_nhwc_conv_gn_partial_floats(N * H_out * W_out, C_out) elements).  *gn_tile_rows = 0 when the map is
 * smaller than a tile or the epilogue is more than a store (relu / a residual / d_out_relu given): nothing was written and the
 * GroupNorm makes its own pass.  hive_nhwc_group_norm_stats = hive_nhwc_group_norm
 * taking its statistics from there (d_gn_partial NULL or gn_tile_rows 0: identical to hive_nhwc_group_norm). */
int64_t hive_nhwc_conv_gn_partial_floats(int64_t n_px, int C_out);
int hive_nhwc_conv_gn(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, int C_in, int C_out, int kernel, int stride,
                      int pad_top, int pad_left, int H_out, int W_out, const void *d_w, const void *d_bias, int relu,
                      const void *d_residual, const void *d_residual2, void *d_out, void *d_out_relu, void *d_gn_partial,
                      int64_t gn_partial_floats, int *gn_tile_rows);
int hive_nhwc_group_norm_stats(hive_ctx *ctx, const void *d_x, int dtype, int N, int HW, int C, int G, const void *d_gamma,
                               const void *d_beta, float eps, const void *d_residual, int relu, void *d_out,
                               const void *d_gn_partial, int gn_tile_rows);

/* d_out = relu?( GroupNorm_G(conv(x); gamma, beta, eps) (+ d_residual) ) with the convolution's own output never written: the
 * bottleneck tails of timm's ResNetV2 (`norm3(conv3(x))` + shortcut + ReLU, and `downsample.norm(downsample.conv(x))`).  The
 * convolution runs twice -- per-tile channel sums first, then normalisation in the epilogue -- which pays where C_out > C_in.
 * Results are bit-identical to hive_nhwc_conv_gn followed by hive_nhwc_group_norm_stats.  d_scratch: float,
 * >= hive_nhwc_conv_gn_partial_floats(N * H_out * W_out, C_out) + 2 * N * G elements.  *fused = 0: not eligible (needs
 * C_out % 256 == 0, (C_out / G) % 8 == 0, H_out * W_out >= 256), nothing was done and the caller runs the separate sequence. */
int hive_nhwc_conv_gn_apply(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, int C_in, int C_out, int kernel,
                            int stride, int pad_top, int pad_left, int H_out, int W_out, const void *d_w, int G,
                            const void *d_gamma, const void *d_beta, float eps, const void *d_residual, int relu, void *d_out,
                            void *d_scratch, int64_t scratch_floats, int *fused);

/* The same for 1 x 1 convolutions (the only kind timm's ResNetV2 uses here) with the GroupNorm's statistics taken from the INPUT instead of a first pass of
 * the convolution: y = W x is linear, so a group's sum is u_g . sum_p x_p and its sum of squares <G_g, sum_p x_p x_p^T> with u_g = sum_{c in g} w_c,
 * G_g = sum_{c in g} w_c w_c^T -- one read of the C_in-wide input and its Gram matrix on the matrix cores (csrc/gram.hip).  hive_gn_gram_prepare makes the
 * tables (hive_gn_gram_table_floats(C_in, G) floats) of a weight tensor [C_out][C_in] once; hive_nhwc_conv_gn_apply_gram then needs d_scratch >= 2 N G floats.
 * These are the statistics of the exact products, not of the 16-bit-rounded outputs: (mean, rstd) agree with the two-pass form to ~1e-4 relative, the
 * outputs to within one rounding in a few places.  *fused = 0: not eligible (hive_nhwc_conv_gn_apply's conditions, C_in in {64, 128, 256}, G % 4 == 0).
 * hive_gn_gram_stats: the statistics alone ([N][G][2] = mean, rstd) and, for tests, the partial Gram matrices [N][parts][C_in][C_in] / sums
 * [N][parts][C_in] (parts = hive_gn_gram_parts(...); d_S_out / d_s_out may be NULL).  Replaces nothing in the reference: an implementation choice behind
 * `norm3(conv3(x))` of /root/reference's DPT-Hybrid backbone (hive/dataset_adaptors.py:1419 runs it). */
int64_t hive_gn_gram_table_floats(int C_in, int G);
int hive_gn_gram_prepare(hive_ctx *ctx, const void *d_w, int dtype, int C_in, int C_out, int G, float *d_tables);
int hive_gn_gram_parts(hive_ctx *ctx, int N, int C_in, int H_out, int W_out);
int hive_gn_gram_stats(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, int C_in, int C_out, int stride, int H_out, int W_out, int G,
                       const float *d_tables, float eps, float *d_stats, float *d_S_out, float *d_s_out);
int hive_nhwc_conv_gn_apply_gram(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, int C_in, int C_out, int stride, int H_out, int W_out,
                                 const void *d_w, const float *d_tables, int G, const void *d_gamma, const void *d_beta, float eps,
                                 const void *d_residual, int relu, void *d_out, void *d_scratch, int64_t scratch_floats, int *fused);

/* DPT-Large (timm vit_large_patch16_384, `DPTDepthModel(backbone="vitl16_384")`) pieces that are not plain convolutions of
 * hive_nhwc_conv:
 *   hive_patch_rows: the patch embedding Conv2d(C, D, P, P) as a GEMM -- the channels-last frame d_x [N][H][W][C] is rearranged into
 *     d_out [N (H/P) (W/P)][P P C] ((ky, kx, c) order = the weight tensor in channels-last memory format); hive_vit_linear with that
 *     weight and the bias then yields the tokens.
 *   hive_nhwc_pixel_shuffle_bias: ConvTranspose2d(C, C, s, s) (reassemble stages 1 and 2) = a 1 x 1 convolution to s s C channels in
 *     (dy, dx, co) order (hive_nhwc_conv) followed by this scatter d_in [N H W][s s C] -> d_out [N][s H][s W][C] (+ d_bias[co]). */
int hive_patch_rows(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, int C, int patch, void *d_out);
int hive_nhwc_pixel_shuffle_bias(hive_ctx *ctx, const void *d_in, const void *d_bias, int dtype, int N, int H, int W, int C, int s,
                                 void *d_out);

/* ResNetV2 stem of the hybrid backbone (timm 0.5.4 ResNetV2.stem, reached from DPTDepthModel.forward): the 7 x 7 stride-2
 * weight-standardised convolution 3 -> 64 with TensorFlow "SAME" padding on the channels-last frame d_x [N][H][W][3] ->
 * d_out [N][ceil(H/2)][ceil(W/2)][64]; d_w = the standardised weights as [64][7][32] ((kx, c) of a kernel row padded from 21 to
 * 32 with zeros).  And MaxPool2dSame(3, 2): [N][H][W][C] -> [N][ceil(H/2)][ceil(W/2)][C], C % 8 == 0. */
int hive_resnet_stem_conv(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, const void *d_w, void *d_out);
int hive_nhwc_maxpool3x3s2(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, int C, void *d_out);
/* The stem as the network object runs it (same results as the three calls above with hive_nhwc_group_norm between them):
 * hive_resnet_stem_conv_gn = the convolution, leaving the statistics of the GroupNorm behind it as hive_nhwc_conv_gn does (per 256-pixel
 * tile; d_gn_partial >= hive_nhwc_conv_gn_partial_floats(N * H_out * W_out, 64) floats; *gn_tile_rows = 0 -- nothing written -- unless
 * H_out % 8 == 0 and W_out % 32 == 0); hive_nhwc_group_norm_relu_maxpool = MaxPool2dSame(3, 2)(relu(GroupNorm(x))) in one pass over
 * x [N][H][W][C] -> [N][ceil(H/2)][ceil(W/2)][C] (d_gn_partial / gn_tile_rows as hive_nhwc_group_norm_stats; NULL / 0: own statistics). */
/* t2 = conv2(relu(norm1(t))) of a 64-channel ResNetV2 bottleneck (timm Bottleneck: `x = norm1(conv1(x)); x = norm2(conv2(x))`) as one kernel:
 * d_x [N][H][W][64] is conv1's raw output with the sums its epilogue left (d_in_partial, in_tile_rows: hive_nhwc_conv_gn), d_gamma / d_beta
 * norm1's parameters, d_w conv2's standardised 3 x 3 weights [64][3][3][64]; d_out = conv2's raw output (bit-identical to
 * hive_nhwc_conv on the separately normalised tensor), and the sums of norm2 as hive_nhwc_conv_gn leaves them (d_gn_partial,
 * *gn_tile_rows; 0: none).  *fused = 0 and nothing done where it does not apply (C != 64, no sums from conv1): run the pair. */
int hive_bneck_gn_conv3x3(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, int C, const void *d_in_partial, int in_tile_rows,
                          const void *d_gamma, const void *d_beta, float eps, const void *d_w, void *d_out, void *d_gn_partial,
                          int64_t gn_partial_floats, int *gn_tile_rows, int *fused);
int hive_resnet_stem_conv_gn(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, const void *d_w, void *d_out, void *d_gn_partial,
                             int64_t gn_partial_floats, int *gn_tile_rows);
int hive_nhwc_group_norm_relu_maxpool(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, int C, int G, const void *d_gamma,
                                      const void *d_beta, float eps, void *d_out, const void *d_gn_partial, int gn_tile_rows);

/* ---- the whole DPT-Hybrid network behind one handle ------------------------------------------------------------------------
 * dpt.models.DPTDepthModel(path, scale, shift, invert, backbone="vitb_rn50_384", non_negative) + .forward(), with the frame
 * pre-processing and the depth hand-off around it, as HIVE's estimate_depth_dpt uses them (hive/dataset_adaptors.py:1366-1374,
 * 1407-1419, 1432-1433; hive/io.py:1032-1039): uint8 RGB frames in HBM -> depth maps in HBM.
 *
 * The weights come as a table of (name, device pointer).  Names are the parameter names of the published isl-org/DPT checkpoint
 * (`dpt_hybrid_nyu-2ce69ec7.pt`); every tensor is of the config's 16-bit `dtype` unless noted:
 *   pretrained.model.patch_embed.backbone.stem.conv.weight        [64][7][32]: STANDARDISED weights, (ky, (kx, c) padded 21 -> 32)
 *   ...backbone.stem.norm.{weight,bias}, ...stages.S.blocks.B.{norm1,norm2,norm3,downsample.norm}.{weight,bias}     [C]
 *   ...stages.S.blocks.B.{conv1,conv2,conv3,downsample.conv}.weight   [C_out][k][k][C_in]: STANDARDISED (timm StdConv2dSame, eps 1e-8)
 *   pretrained.model.patch_embed.proj.{weight [768][1][1][1024], bias [768]},  pretrained.model.cls_token [768]
 *   pretrained.model.blocks.I.{norm1,norm2}.{weight,bias} f32 [768]; attn.qkv.weight [2304][768], attn.qkv.bias f32; attn.proj.*;
 *   mlp.fc1.weight [3072][768], mlp.fc1.bias f32; mlp.fc2.weight [768][3072], mlp.fc2.bias f32            (I = 0 .. 11)
 *   pretrained.act_postprocess{3,4}.0.project.0.{weight [768][1536], bias f32 [768]}
 *   pretrained.act_postprocess3.3.*, act_postprocess4.3.* ([768][1][1][768] + bias), act_postprocess4.4.* ([768][3][3][768] + bias)
 *   scratch.layer{1..4}_rn.weight [256][3][3][C_in];  scratch.refinenet{1..4}.resConfUnit{1,2}.conv{1,2}.{weight [256][3][3][256], bias};
 *   scratch.refinenet{1..4}.out_conv.{weight [256][1][1][256], bias}
 *   scratch.output_conv.0.{weight [128][3][3][256], bias [128]}; scratch.output_conv.2.weight as
 *   [ky][kx][32][128]; the last two layers' host values go in the config (head_b3 = output_conv.2.bias, head_w1 / head_b1 = output_conv.4).
 * All convolution weights are [C_out][ky][kx][C_in] (= the PyTorch tensor in channels-last memory format).  The pointers must
 * stay valid for the life of the handle.
 * backbone 1 (DPT-Large, `dpt_large-midas-2f21e586.pt`: no ResNet; 24 blocks of width 1024, hooks 5 / 11 / 17 / 23) differs in:
 *   pretrained.model.patch_embed.proj.weight [1024][16][16][3] (the GEMM's [1024][768]), ...proj.bias.f32 (f32 [1024]), cls_token [1024];
 *   blocks.I.* with 768 -> 1024, 3072 -> 4096 (I = 0 .. 23); act_postprocess{1..4}.0.project.0.{weight [1024][2048], bias f32};
 *   act_postprocess1.3.* [256][1][1][1024], act_postprocess1.4.weight.rows [4 4 256][256] = ConvTranspose2d(256, 256, 4, 4).weight
 *   permuted to ((dy, dx, co), ci), act_postprocess1.4.bias; act_postprocess2.3.* [512][1][1][1024], act_postprocess2.4.weight.rows
 *   [2 2 512][512], .bias; act_postprocess3.3.* [1024][1][1][1024]; act_postprocess4.3.* likewise, act_postprocess4.4.* [1024][3][3][1024];
 *   scratch.layer{1..4}_rn.weight with C_in = 256 / 512 / 1024 / 1024. */
typedef struct hive_dpt hive_dpt;
typedef struct hive_dpt_tensor {
    const char *name;
    const void *data;
} hive_dpt_tensor;
typedef struct hive_dpt_config {
    int backbone;                 /* 0 = vitb_rn50_384 (DPT-Hybrid: the one HIVE instantiates), 1 = vitl16_384 (DPT-Large) */
    float scale, shift;           /* depth = 1 / max(scale * x + shift, 1e-8) when invert */
    int invert, non_negative;
    float gn_eps, ln_eps;         /* 1e-5 (GroupNorm), 1e-6 (timm ViT LayerNorm) */
    float head_b3[32], head_w1[32], head_b1;
    int dtype;                    /* HIVE_BF16 or HIVE_F16: the type of every 16-bit tensor of the table, of d_pos_embed and of the activations */
} hive_dpt_config;
int hive_dpt_create(hive_ctx *ctx, const hive_dpt_config *config, const hive_dpt_tensor *tensors, int n_tensors, hive_dpt **out);
/* d_rgb u8 [B][H][W][3] (H, W multiples of 32), d_pos_embed (the config's dtype) [(H/16)(W/16) + 1][768 | 1024] = the position embedding resized to this
 * token grid (dpt `_resize_pos_embed`: evaluated once per frame size by the host binding).  Outputs [B][H][W], any may be NULL (not
 * all): d_depth f32 metres; d_out_mm = uint16(depth * 1000); d_out_m = mm / 1000 with > max_depth -> 0. */
int hive_dpt_forward(hive_dpt *dpt, const uint8_t *d_rgb, int B, int H, int W, const void *d_pos_embed, float *d_depth, float max_depth,
                     uint16_t *d_out_mm, float *d_out_m);
/* Frames of ANY size (BASELINE config 4: 1920 x 1080): d_rgb u8 [B][frame_h][frame_w][3] enters through hive_dpt_resize_preprocess at the network's
 * net_h x net_w (multiples of 32; the reference's rule gives 864 x 480 for 1080p, hive/dataset_adaptors.py:1376-1389), the depth map leaves through
 * hive_depth_resize_nearest at the frame size (:1421-1426): outputs [B][frame_h][frame_w].  d_pos_embed is the embedding of the NETWORK's token grid
 * (net_h / 16)(net_w / 16) + 1.  With net == frame size this is hive_dpt_forward. */
int hive_dpt_forward_frames(hive_dpt *dpt, const uint8_t *d_rgb, int B, int frame_h, int frame_w, int net_h, int net_w, const void *d_pos_embed,
                            float *d_depth, float max_depth, uint16_t *d_out_mm, float *d_out_m);
/* Bytes of the activation arena (grown to the largest forward seen: the high-water mark of its maps, which are released behind
 * their last consumer -- ~10 GB for 96 frames of 480 x 640, where the maps total 38 GB). */
int hive_dpt_arena_bytes(hive_dpt *dpt, int64_t *bytes);
/* The table's tensors were overwritten in place: refresh what the object derived from them -- the ViT engine's folded LayerNorm weights (captured at
 * hive_dpt_create, see hive_vit_weights_modified) and the Gram tables of the bottlenecks' 1 x 1 convolutions (made on the first forward that needs them).
 * Everything else is read through the table's pointers at every forward.  Pointers must stay the same; new tensors need a new object. */
int hive_dpt_weights_modified(hive_dpt *dpt);
int hive_dpt_destroy(hive_dpt *dpt);

#ifdef __cplusplus
}
#endif
#endif /* HIVE_MI355X_H */
