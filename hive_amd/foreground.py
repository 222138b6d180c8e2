"""Foreground per-frame meshing inner loops of ``Pipeline._create_scene`` (/root/reference/hive/pipeline.py:340-483) on the
MI355X: the steps behind ``point_cloud_from_depth`` that are dense per-pixel work -- triangulation of the valid pixels, the
face filter, the texture window and UV coordinates.  Decimation (openmesh), connected-component clean-up and atlas packing
(trimesh / numpy glue) stay with the reference.

  ``grid_faces``                 fused ``_triangulate_faces`` + ``_filter_faces`` for one object mask of one frame
  ``triangulate_faces``          ``Pipeline._triangulate_faces(points)`` (:651-667) for lattice points
  ``filter_faces``               ``Pipeline._filter_faces(points2d, depth, faces, options)`` (:670-694), any face list
  ``get_mesh_texture_and_uv``    ``Pipeline._get_mesh_texture_and_uv(...)`` (:782-808)

The triangulation is the implicit one of the pixel grid (csrc/fgmesh.hip): unit squares and triangles of the valid pixels plus the
(sqrt 2, sqrt 2, 2) triangles with which a lattice Delaunay bridges one-pixel holes -- after the reference's filter (sides <= 2 pixels by
default) the face set equals the one the reference gets from Qhull, up to which diagonal splits four co-circular points (Qhull's
arbitrary choice); tests/test_fgmesh_gpu.py checks that against scipy.
"""
import ctypes

import numpy as np

from hive_amd import _lib
from hive_amd._lib import MEM_DEVICE, MEM_HOST, ptr
from hive_amd.options import MeshFilteringOptions
from hive_amd.utils import validate_camera_parameter_shapes, validate_shape


def _is_torch(x):
    return hasattr(x, "data_ptr")


def grid_faces(depth, mask, options: MeshFilteringOptions = None, ctx=None, return_vertex_count=False):
    """Faces of the valid pixels (``mask & (depth > 0)``) of one depth map, already filtered: int32 (F, 3) indices into the
    rows of ``point_cloud_from_depth(depth, mask, ...)``.  ``depth`` float32 (H, W) (numpy or device tensor), ``mask`` bool /
    uint8 (H, W) or None."""
    options = options or MeshFilteringOptions()
    ctx = ctx or _lib.default_context()
    if _is_torch(depth):
        import torch
        d = depth.contiguous()
        assert d.dtype == torch.float32
        m = None if mask is None else mask.to(torch.uint8).contiguous()
        mem = MEM_DEVICE
    else:
        d = np.ascontiguousarray(depth, dtype=np.float32)
        m = None if mask is None else np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8)
        mem = MEM_HOST
    assert d.ndim == 2 and (m is None or tuple(m.shape) == tuple(d.shape)), "depth (H, W) and mask (H, W)"
    h, w = (int(v) for v in d.shape)
    nf, nv = ctypes.c_int64(0), ctypes.c_int64(0)
    args = (ctx.handle, ptr(d), ptr(m), h, w, float(options.max_pixel_distance), float(options.max_depth_distance), mem)
    ctx.check(ctx.lib.hive_grid_mesh(*args, None, 0, ctypes.byref(nf), ctypes.byref(nv)))  # size the output
    if mem == MEM_DEVICE:
        import torch
        faces = torch.empty((nf.value, 3), dtype=torch.int32, device=d.device)
    else:
        faces = np.empty((nf.value, 3), np.int32)
    if nf.value:
        ctx.check(ctx.lib.hive_grid_mesh(*args, ptr(faces), nf.value, ctypes.byref(nf), ctypes.byref(nv)))
    return (faces, nv.value) if return_vertex_count else faces


def triangulate_faces(points, ctx=None):
    """``Pipeline._triangulate_faces``: faces of a set of 2D lattice points (N, 2) = (u, v) integer pixel coordinates in
    row-major order (what ``np.vstack((u, v)).T`` of ``valid.nonzero()`` gives, pipeline.py:392-395) -- no filtering."""
    validate_shape(points, 'points', expected_shape=(None, 2))
    pts = np.asarray(points)
    assert np.issubdtype(pts.dtype, np.integer) and len(pts) > 0, "lattice (integer pixel) points expected"
    u0, v0 = pts[:, 0].min(), pts[:, 1].min()
    w, h = int(pts[:, 0].max() - u0 + 1), int(pts[:, 1].max() - v0 + 1)
    grid = np.zeros((h, w), np.float32)
    grid[pts[:, 1] - v0, pts[:, 0] - u0] = 1.0
    order = np.lexsort((pts[:, 0], pts[:, 1]))
    assert np.array_equal(order, np.arange(len(pts))), "points must be in row-major (v, then u) order"
    return grid_faces(grid, None, MeshFilteringOptions(max_pixel_distance=np.inf, max_depth_distance=np.inf), ctx=ctx)


def filter_faces(points2d, depth, faces, options: MeshFilteringOptions, ctx=None):
    """``Pipeline._filter_faces``: the faces whose three edges are at most ``max_pixel_distance`` long in image space and span
    at most ``max_depth_distance`` of depth; order preserved.  Works on any triangulation of the points."""
    validate_shape(points2d, 'points2d', expected_shape=(None, 2))
    validate_shape(depth, 'depth', expected_shape=(points2d.shape[0],))
    validate_shape(faces, 'faces', expected_shape=(None, 3))
    ctx = ctx or _lib.default_context()
    p = np.ascontiguousarray(points2d, dtype=np.int32)
    d = np.ascontiguousarray(depth, dtype=np.float32)
    f = np.ascontiguousarray(faces, dtype=np.int32)
    out = np.empty_like(f)
    n = ctypes.c_int64(0)
    ctx.check(ctx.lib.hive_filter_faces(ctx.handle, ptr(p), ptr(d), len(p), ptr(f), len(f), float(options.max_pixel_distance),
                                        float(options.max_depth_distance), MEM_HOST, ptr(out), ctypes.byref(n)))
    return out[:n.value].astype(np.asarray(faces).dtype, copy=False)


def get_mesh_texture_and_uv(vertices, image, camera_matrix, rotation=np.eye(3), translation=np.zeros((3, 1)), scale_factor=1.0, ctx=None):
    """``Pipeline._get_mesh_texture_and_uv``: (cropped texture, UV coordinates relative to the crop's corner)."""
    validate_shape(vertices, 'vertices', expected_shape=(None, 3))
    validate_shape(image, 'image', expected_shape=(None, None, 3))
    validate_camera_parameter_shapes(camera_matrix, rotation, translation)
    ctx = ctx or _lib.default_context()
    pts = np.ascontiguousarray(vertices, dtype=np.float64)
    K = np.ascontiguousarray(camera_matrix, dtype=np.float64).reshape(3, 3)
    R = np.ascontiguousarray(rotation, dtype=np.float64).reshape(3, 3)
    t = np.ascontiguousarray(translation, dtype=np.float64).reshape(3)
    uv = np.empty((len(pts), 2), np.int32)  # world2image's default dtype: rounded pixel coordinates
    box = np.zeros(4, np.int32)
    ctx.check(ctx.lib.hive_texture_window(ctx.handle, ptr(pts), len(pts), ptr(K), ptr(R), ptr(t), float(scale_factor), MEM_HOST, ptr(uv), ptr(box)))
    min_u, min_v, max_u, max_v = (int(b) for b in box)
    texture = image[min_v:max_v, min_u:max_u, :].copy()
    return texture, uv


class FrameMeshBuffers:
    """Device buffers of worst-case size for ``frame_mesh`` (H W vertices, 4 H W faces), reused from frame to frame."""

    def __init__(self, height, width, device="cuda"):
        import torch
        n = int(height) * int(width)
        self.shape = (int(height), int(width))
        self.vertices = torch.empty((n, 3), dtype=torch.float64, device=device)
        self.faces = torch.empty((4 * n, 3), dtype=torch.int32, device=device)
        self.uv = torch.empty((n, 2), dtype=torch.int32, device=device)


def frame_mesh(depth, mask, image, camera_matrix, rotation=np.eye(3), translation=np.zeros((3, 1)), options: MeshFilteringOptions = None, ctx=None,
               buffers: FrameMeshBuffers = None):
    """One object of one frame, device-resident, in ONE library call (``hive_fg_frame_mesh``): what the loop body of ``process_frame``
    (/root/reference/hive/pipeline.py:383-461) computes between the binary mask and the texture atlas, minus its CPU-library stages (decimation, connected
    components, billboard) --

        vertices = point_cloud_from_depth(depth, mask, K, R, t)             (:386)
        faces    = _filter_faces(points2d, depth[valid], _triangulate_faces(points2d), options)   (:402-408)
        texture, uv = _get_mesh_texture_and_uv(vertices, rgb, K, R, t)      (:453)

    ``depth`` float32 (H, W), ``mask`` bool / uint8 (H, W) or None, ``image`` uint8 (H, W, 3): device tensors (numpy arrays are uploaded).  Returns a dict of
    device tensors -- ``vertices`` float64 (V, 3), ``faces`` int32 (F, 3), ``uv`` int32 (V, 2), ``texture`` uint8 crop -- and ``bbox`` (min_u, min_v, max_u, max_v);
    views into ``buffers`` (valid until its next use) when one is given.  Bit-identical to the three separate functions; seven launches and one read-back instead
    of sixteen launches, four read-backs and the host round trip of the vertices.  An object with no valid pixel gives V = F = 0 and ``texture`` None."""
    import torch
    options = options or MeshFilteringOptions()
    validate_camera_parameter_shapes(camera_matrix, rotation, translation)
    dev = depth.device if _is_torch(depth) else torch.device("cuda", torch.cuda.current_device())
    ctx = ctx or _lib.default_context(dev.index or 0)
    ctx.follow_torch_stream()
    as_dev = lambda a, dt: (a if _is_torch(a) else torch.from_numpy(np.ascontiguousarray(a))).to(device=dev, dtype=dt).contiguous()
    d = as_dev(depth, torch.float32)
    m = None if mask is None else as_dev(mask, torch.uint8)
    img = None if image is None else as_dev(image, torch.uint8)
    assert d.dim() == 2 and (m is None or tuple(m.shape) == tuple(d.shape)), "depth (H, W) and mask (H, W)"
    h, w = (int(v) for v in d.shape)
    buffers = buffers or FrameMeshBuffers(h, w, dev)
    assert buffers.shape == (h, w), "buffers were sized for another frame size"
    from hive_amd.geometric import _kinv
    K = np.ascontiguousarray(camera_matrix, dtype=np.float64).reshape(3, 3)
    Kinv = _kinv(np.asarray(camera_matrix).reshape(3, 3))  # as point_cloud_from_depth computes it: inverted in K's own dtype (geometric.py:203)
    R = np.ascontiguousarray(rotation, dtype=np.float64).reshape(3, 3)
    t = np.ascontiguousarray(translation, dtype=np.float64).reshape(3)
    nv, nf = ctypes.c_int64(0), ctypes.c_int64(0)
    box = np.zeros(4, np.int32)
    ctx.check(ctx.lib.hive_fg_frame_mesh(ctx.handle, ptr(d), ptr(m), h, w, ptr(Kinv), ptr(K), ptr(R), ptr(t), float(options.max_pixel_distance),
                                         float(options.max_depth_distance), ptr(buffers.vertices), buffers.vertices.shape[0], ptr(buffers.faces), buffers.faces.shape[0],
                                         ptr(buffers.uv), ctypes.byref(nv), ctypes.byref(nf), ptr(box)))
    min_u, min_v, max_u, max_v = (int(b) for b in box)
    texture = None
    if nv.value and img is not None:
        texture = img[max(min_v, 0):max_v, max(min_u, 0):max_u, :].clone()  # `image[min_v:max_v, min_u:max_u, :].copy()` (pipeline.py:805)
    return {"vertices": buffers.vertices[:nv.value], "faces": buffers.faces[:nf.value], "uv": buffers.uv[:nv.value], "texture": texture,
            "bbox": (min_u, min_v, max_u, max_v)}


def pack_textures_row(textures, uvs):
    """``Pipeline._pack_textures(textures_atlas, uvs_atlas, n_rows=1)`` (/root/reference/hive/pipeline.py:811-866) for the one form the reference calls: the objects' texture
    crops side by side in ONE row (top-aligned, zero below the shorter ones), every object's u shifted by the widths in front of it, then u / atlas width and
    v -> 1 - v / atlas height.  ``textures``: uint8 (h_i, w_i, 3) device tensors, ``uvs``: (V_i, 2) integer tensors relative to their crop.  Returns (atlas uint8
    (max h, sum w, 3), uv float64 (sum V, 2)) on the device."""
    import torch
    assert len(textures) == len(uvs) and len(textures) > 0
    height, width = max(int(t.shape[0]) for t in textures), sum(int(t.shape[1]) for t in textures)
    atlas = torch.zeros((height, width, 3), dtype=torch.uint8, device=textures[0].device)
    shifted, x = [], 0
    for tex, uv in zip(textures, uvs):
        h, w = int(tex.shape[0]), int(tex.shape[1])
        atlas[:h, x:x + w] = tex
        moved = uv.to(torch.float64)
        moved[:, 0] += x
        shifted.append(moved)
        x += w
    out = torch.cat(shifted)
    # (tensor / tensor: torch turns a division by a Python scalar into a multiplication by its reciprocal on the GPU -- one ulp off numpy's `/=` in places)
    size = torch.tensor([float(width), float(height)], dtype=torch.float64, device=out.device)
    out = out / size
    out[:, 1] = 1.0 - out[:, 1]
    return atlas, out


def process_frame(rgb, depth, mask_encoded, camera_matrix, pose, dilation_options=None, filtering_options: MeshFilteringOptions = None,
                  disable_coverage_constraint=False, ctx=None, buffers: FrameMeshBuffers = None):
    """The body of ``process_frame`` in ``Pipeline._create_scene`` (/root/reference/hive/pipeline.py:340-483) for the dynamic objects of ONE frame, device-resident:
    for every object id 1 .. mask_encoded.max(): the binary mask, dilated (``dilate_mask``, :369-370); skipped when it covers less than 1 % of the frame (:374-379); its
    mesh in one library call (``frame_mesh``: point cloud, triangulation + face filter, texture window); skipped with fewer than 9 vertices or no face (:388-391, 410-413);
    the objects' vertices stacked, their faces offset by the vertices in front, their textures packed in one atlas row (:463-468, 811-866).  The reference's CPU-library
    stages in between -- decimation (openmesh), connected components (trimesh), billboard -- are outside this build's scope and are NOT applied.

    ``rgb`` uint8 (H, W, 3+), ``depth`` float32 (H, W), ``mask_encoded`` uint8 (H, W) instance ids (0 = background), ``pose`` the frame's 4 x 4 world-to-camera transform
    (``dataset.camera_trajectory.to_homogenous_transforms()[index]``).  Numpy arrays or device tensors.  Returns None for a frame without a surviving object (the reference
    returns an empty ``trimesh.Trimesh()``), else a dict of device tensors: ``vertices`` float64 (V, 3), ``faces`` int64 (F, 3), ``uv`` float64 (V, 2) in atlas
    coordinates, ``texture`` uint8 atlas, and ``objects`` = the ids that survived."""
    import torch
    from hive_amd.options import MaskDilationOptions
    dilation_options = dilation_options or MaskDilationOptions()
    filtering_options = filtering_options or MeshFilteringOptions()
    dev = depth.device if _is_torch(depth) else torch.device("cuda", torch.cuda.current_device())
    ctx = ctx or _lib.default_context(dev.index or 0)
    as_dev = lambda a, dt: (a if _is_torch(a) else torch.from_numpy(np.ascontiguousarray(a))).to(device=dev, dtype=dt).contiguous()
    rgb_d = as_dev(rgb, torch.uint8)[:, :, :3].contiguous()  # (`rgb[:, :, :3]`, :355)
    depth_d = as_dev(depth, torch.float32)
    ids = as_dev(mask_encoded, torch.uint8)
    h, w = (int(v) for v in depth_d.shape)
    buffers = buffers or FrameMeshBuffers(h, w, dev)
    pose = np.asarray(pose, dtype=np.float64).reshape(4, 4)
    rotation, translation = pose[:3, :3], pose[:3, 3:4]  # get_pose_components
    se = dilation_options.structuring_element()
    n_objects = int(ids.max().item())
    vertices, faces, uvs, textures, kept = [], [], [], [], []
    vertex_count = 0
    for object_id in range(1, n_objects + 1):
        mask = (ids == object_id).to(torch.uint8)
        if dilation_options.num_iterations > 0:
            ctx.follow_torch_stream()
            grown = torch.empty_like(mask)
            ctx.check(ctx.lib.hive_dilate_mask_se(ctx.handle, ptr(mask), h, w, ptr(se), se.shape[0], se.shape[1], int(dilation_options.num_iterations), MEM_DEVICE, ptr(grown)))
            mask = grown
        if float(mask.float().mean().item()) < 0.01 and not disable_coverage_constraint:
            continue
        mesh = frame_mesh(depth_d, mask, rgb_d, camera_matrix, rotation, translation, filtering_options, ctx=ctx, buffers=buffers)
        if mesh["vertices"].shape[0] < 9 or mesh["faces"].shape[0] < 1:
            continue
        vertices.append(mesh["vertices"].clone())
        faces.append(mesh["faces"].to(torch.int64) + vertex_count)
        uvs.append(mesh["uv"].clone())
        textures.append(mesh["texture"])
        vertex_count += int(mesh["vertices"].shape[0])
        kept.append(object_id)
    if not kept:
        return None
    atlas, uv = pack_textures_row(textures, uvs)
    return {"vertices": torch.cat(vertices), "faces": torch.cat(faces), "uv": uv, "texture": atlas, "objects": kept}
