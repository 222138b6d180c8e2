"""MI355X drop-in for the TSDF side of HIVE's hot path.

Two layers, both mirroring the reference's names, argument meaning and error behaviour:

* ``TSDFVolume`` / ``get_view_frustum`` / ``rigid_transform`` -- the module object HIVE imports as
  ``from third_party.tsdf_fusion_python import fusion`` (/root/reference/hive/fusion.py:34; call
  sites :59, :104, :124, :127).  The volumes live in HBM; every method calls the HIP kernels in
  ``libhive_mi355x.so`` through its C ABI.
* ``adjust_voxel_size`` / ``tsdf_fusion`` -- the driver functions of /root/reference/hive/fusion.py:37-134.
"""
import ctypes
import logging
from typing import List, Optional, Tuple

import numpy as np

from hive_amd import _lib
from hive_amd._lib import MEM_DEVICE, MEM_HOST, ptr
from hive_amd.options import MaskDilationOptions


def rigid_transform(xyz, transform):
    """Applies a rigid transform to an (N, 3) pointcloud (helper of the reference library)."""
    xyz_h = np.hstack([xyz, np.ones((len(xyz), 1), dtype=np.float32)])
    xyz_t_h = np.dot(transform, xyz_h.T).T
    return xyz_t_h[:, :3]


def _is_torch(x):
    return hasattr(x, "data_ptr")


def get_view_frustum(depth_im, cam_intr, cam_pose, ctx=None):
    """Get corners of 3D camera view frustum of depth image -> (3, 5) float64
    (apex + 4 image corners at max(depth)); call site /root/reference/hive/fusion.py:59."""
    ctx = ctx or _lib.default_context()
    if _is_torch(depth_im):
        depth, mem = depth_im.contiguous(), MEM_DEVICE
        assert str(depth.dtype) == "torch.float32"
    else:
        depth, mem = np.ascontiguousarray(depth_im, dtype=np.float32), MEM_HOST
    assert depth.ndim == 2, "depth_im must be (H, W)"
    K = np.ascontiguousarray(cam_intr, dtype=np.float32).reshape(3, 3)
    pose = np.ascontiguousarray(cam_pose, dtype=np.float64).reshape(4, 4)
    out = np.empty((3, 5), np.float64)
    ctx.check(ctx.lib.hive_view_frustum(ctx.handle, ptr(depth), depth.shape[0], depth.shape[1], ptr(K), ptr(pose), mem, ptr(out)))
    return out


def volume_dims(vol_bnds, voxel_size):
    """vol_dim = ceil((max - min) / voxel_size) of the reference library, without creating a volume."""
    bnds = np.ascontiguousarray(vol_bnds, dtype=np.float64).reshape(3, 2)
    dim = np.zeros(3, np.int64)
    _lib.check(_lib.load().hive_tsdf_dims(ptr(bnds), float(voxel_size), ptr(dim)))
    return dim


class TSDFVolume:
    """Volumetric TSDF Fusion of RGB-D Images, resident in MI355X HBM.

    Same constructor and methods as the reference library's class (``use_gpu`` is accepted and
    ignored: there is only the GPU path, and it fails loudly without a device).
    """

    def __init__(self, vol_bnds, voxel_size, use_gpu=True, ctx=None, round_mode=None, storage=None, x_range=None):
        """
        :param vol_bnds: (3, 2) array of the xyz bounds (min/max) in metres.
        :param voxel_size: The volume discretisation in metres.
        :param use_gpu: everything runs on the MI355X either way; the flag selects WHICH of the reference
            library's two arithmetic paths is reproduced.  ``True`` (the reference's default, and what its
            Docker image runs: pycuda==2021.1, requirements.txt:14) = the CUDA kernel's ``roundf`` (ties away
            from zero) for the pixel projection and the colour average; ``False`` = the numpy path's
            ``np.round`` (ties to even; BASELINE config 1).  They differ only on exact ``.5`` ties.
        :param round_mode: explicit ``hive_amd._lib.ROUND_HALF_EVEN / ROUND_HALF_AWAY``; overrides ``use_gpu``.
        :param storage: optional 3-tuple of float32 torch tensors (tsdf, weight, colour), each with
            prod(vol_dim) elements, to keep the volume in caller-owned device memory.
        :param x_range: ``(x0, x1)``: hold only the x-slab ``x0 <= x < x1`` of the grid defined by ``vol_bnds`` (bit-exact
            multi-GPU mode, ``hive_amd.distributed.ExactSlabFusion``); ``vol_dim`` is then the slab's.
        """
        vol_bnds = np.asarray(vol_bnds, dtype=np.float64)
        assert vol_bnds.shape == (3, 2), "[!] `vol_bnds` should be of shape (3, 2)."
        self._ctx = ctx or _lib.default_context()
        lib = self._ctx.lib
        if round_mode is None:
            round_mode = _lib.ROUND_HALF_AWAY if use_gpu else _lib.ROUND_HALF_EVEN
        self.round_mode = int(round_mode)
        self._voxel_size = float(voxel_size)
        self._trunc_margin = 5 * self._voxel_size  # truncation on SDF
        self._color_const = 256 * 256
        self._storage = storage
        handle = ctypes.c_void_p()
        bnds = np.ascontiguousarray(vol_bnds)
        s = storage or (None, None, None)
        x0, x1 = (0, -1) if x_range is None else (int(x_range[0]), int(x_range[1]))
        self._ctx.check(lib.hive_tsdf_create_slab(self._ctx.handle, ptr(bnds), self._voxel_size, x0, x1, ptr(s[0]), ptr(s[1]), ptr(s[2]),
                                                  ctypes.byref(handle)))
        self._handle = handle
        self._ctx.check(lib.hive_tsdf_set_round_mode(handle, self.round_mode))  # per volume, not per context
        dim = np.zeros(3, np.int64)
        origin = np.zeros(3, np.float32)
        adj = np.zeros(6, np.float64)
        self._ctx.check(lib.hive_tsdf_info(handle, ptr(dim), ptr(origin), ptr(adj), None, None))
        self._vol_dim = dim.astype(int)
        self._vol_bnds = adj.reshape(3, 2)
        self._vol_origin = origin
        self.gpu_mode = True
        logging.debug("Voxel volume size: %d x %d x %d - # points: %d", *self._vol_dim, int(np.prod(self._vol_dim)))

    # -- properties kept for callers that poke at the reference object --------------------------
    @property
    def vol_dim(self):
        return self._vol_dim

    @property
    def num_voxels(self):
        return int(np.prod(self._vol_dim.astype(np.int64)))

    def _frame_args(self, color_im, depth_im):
        self._ctx.follow_torch_stream()  # device tensors are ordered on torch's current stream
        if _is_torch(depth_im):
            import torch
            assert _is_torch(color_im), "colour and depth must both be device tensors or both numpy arrays"
            depth = depth_im.contiguous()
            color = color_im.contiguous()
            assert depth.dtype == torch.float32 and color.dtype == torch.uint8
            return color, depth, MEM_DEVICE
        depth = np.ascontiguousarray(depth_im, dtype=np.float32)
        color = np.ascontiguousarray(color_im, dtype=np.uint8)
        return color, depth, MEM_HOST

    def integrate(self, color_im, depth_im, cam_intr, cam_pose, obs_weight=1., return_n_updated=False):
        """Integrate an RGB-D frame into the TSDF volume (call site /root/reference/hive/fusion.py:124).

        :param color_im: An RGB image of shape (H, W, 3), uint8 (numpy or a device tensor).
        :param depth_im: A depth image of shape (H, W), float32 metres, 0 = invalid.
        :param cam_intr: The camera intrinsics matrix of shape (3, 3).
        :param cam_pose: The camera pose (camera-to-world) of shape (4, 4).
        :param obs_weight: The weight to assign for the current observation.
        """
        color, depth, mem = self._frame_args(color_im, depth_im)
        assert depth.ndim == 2 and tuple(color.shape) == tuple(depth.shape) + (3,), "color_im must be (H, W, 3) matching depth_im (H, W)"
        K = np.ascontiguousarray(cam_intr, dtype=np.float32).reshape(3, 3)
        pose = np.ascontiguousarray(cam_pose, dtype=np.float64).reshape(4, 4)
        n = ctypes.c_uint64(0)
        self._ctx.check(self._ctx.lib.hive_tsdf_integrate(self._handle, ptr(color), ptr(depth), depth.shape[0], depth.shape[1],
                                                          ptr(K), ptr(pose), float(obs_weight), mem,
                                                          ctypes.byref(n) if return_n_updated else None))
        return n.value if return_n_updated else None

    def integrate_batch(self, color_ims, depth_ims, cam_intr, cam_poses, obs_weight=1.):
        """Integrate frames [T,H,W,3] / [T,H,W] / poses [T,4,4] in order (same result as T integrate calls)."""
        color, depth, mem = self._frame_args(color_ims, depth_ims)
        assert depth.ndim == 3 and tuple(color.shape) == tuple(depth.shape) + (3,)
        K = np.ascontiguousarray(cam_intr, dtype=np.float32).reshape(3, 3)
        poses = np.ascontiguousarray(cam_poses, dtype=np.float64).reshape(depth.shape[0], 4, 4)
        self._ctx.check(self._ctx.lib.hive_tsdf_integrate_batch(self._handle, depth.shape[0], ptr(color), ptr(depth), depth.shape[1],
                                                                depth.shape[2], ptr(K), ptr(poses), float(obs_weight), mem))

    def last_batch_groups(self):
        """Frames per sweep of the most recent ``integrate_batch`` (1 = single-frame kernel), in launch order."""
        sizes = (ctypes.c_int * 4096)()
        n = ctypes.c_int(0)
        self._ctx.check(self._ctx.lib.hive_tsdf_last_batch_groups(self._handle, ctypes.cast(sizes, ctypes.c_void_p), 4096, ctypes.byref(n)))
        return [int(sizes[i]) for i in range(min(n.value, 4096))]

    def planes_modified(self):
        """Tell the library that the caller wrote the volume's planes itself (caller-owned ``storage``): cached results and the fast paths that
        rely on what the library knows about their contents are dropped until the next ``reset``."""
        self._ctx.check(self._ctx.lib.hive_tsdf_planes_modified(self._handle))

    def last_sweep_voxels(self):
        """Voxels on the work list of the most recent sweep (segments x voxels per segment); forces a stream sync."""
        n, seg = ctypes.c_uint64(0), ctypes.c_int(0)
        self._ctx.check(self._ctx.lib.hive_tsdf_last_sweep_items(self._handle, ctypes.byref(n), ctypes.byref(seg)))
        return int(n.value) * int(seg.value)

    def stats(self):
        """(frames, launches) this volume has been given since its creation / last ``reset`` (``hive_tsdf_stats``; no sync)."""
        f, n = ctypes.c_int64(0), ctypes.c_int64(0)
        self._ctx.check(self._ctx.lib.hive_tsdf_stats(self._handle, ctypes.byref(f), ctypes.byref(n)))
        return int(f.value), int(n.value)

    def get_volume(self, with_weight=False):
        shape = tuple(int(v) for v in self._vol_dim)
        tsdf = np.empty(shape, np.float32)
        color = np.empty(shape, np.float32)
        weight = np.empty(shape, np.float32) if with_weight else None
        self._ctx.check(self._ctx.lib.hive_tsdf_get_volume(self._handle, ptr(tsdf), ptr(color), ptr(weight)))
        return (tsdf, color, weight) if with_weight else (tsdf, color)

    def set_volume(self, tsdf=None, color=None, weight=None):
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float32) for a in (tsdf, color, weight)]
        for a in arrs:
            assert a is None or a.size == self.num_voxels
        self._ctx.check(self._ctx.lib.hive_tsdf_set_volume(self._handle, ptr(arrs[0]), ptr(arrs[1]), ptr(arrs[2])))

    def reset(self):
        self._ctx.check(self._ctx.lib.hive_tsdf_reset(self._handle))

    def set_volume_device(self, tsdf=None, color=None, weight=None):
        """Overwrite the volume from float32 device tensors (device-to-device copies on the volume's stream)."""
        self._ctx.follow_torch_stream()
        arrs = [None if a is None else a.contiguous() for a in (tsdf, color, weight)]
        for a in arrs:
            assert a is None or (a.numel() == self.num_voxels and str(a.dtype) == "torch.float32" and a.is_cuda)
        self._ctx.check(self._ctx.lib.hive_tsdf_set_volume(self._handle, ptr(arrs[0]), ptr(arrs[1]), ptr(arrs[2])))

    def device_tensors(self):
        """Copies of (tsdf, weight, colour) as float32 device tensors of num_voxels elements."""
        import torch
        self._ctx.follow_torch_stream()
        outs = [torch.empty(self.num_voxels, dtype=torch.float32, device=f"cuda:{self._ctx.device}") for _ in range(3)]
        self._ctx.check(self._ctx.lib.hive_tsdf_get_volume(self._handle, ptr(outs[0]), ptr(outs[2]), ptr(outs[1])))
        return tuple(outs)

    def device_ptrs(self):
        a, b, c = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        self._ctx.check(self._ctx.lib.hive_tsdf_device_ptrs(self._handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    def _extract(self):
        self._ctx.follow_torch_stream()
        nv, nf = ctypes.c_int64(0), ctypes.c_int64(0)
        rc = self._ctx.lib.hive_tsdf_extract_mesh(self._handle, ctypes.byref(nv), ctypes.byref(nf))
        if rc == _lib.ERR_EMPTY:
            # same exception type and message as scikit-image's marching cubes, which HIVE catches
            # (/root/reference/scripts/experiments.py:165-168)
            raise ValueError("Surface level must be within volume data range.")
        self._ctx.check(rc)
        return nv.value, nf.value

    def get_mesh(self, return_voxel_coords=False):
        """Compute a mesh from the voxel volume using marching cubes (call site hive/fusion.py:127).

        :return: verts (V,3) float32 world coordinates, faces (F,3) int32, norms (V,3) float32, colors (V,3) uint8.
        """
        nv, nf = self._extract()
        verts = np.empty((nv, 3), np.float32)
        faces = np.empty((nf, 3), np.int32)
        norms = np.empty((nv, 3), np.float32)
        colors = np.empty((nv, 3), np.uint8)
        self._ctx.check(self._ctx.lib.hive_tsdf_copy_mesh(self._handle, ptr(verts), ptr(faces), ptr(norms), ptr(colors)))
        if return_voxel_coords:
            vvox = np.empty((nv, 3), np.float32)
            self._ctx.check(self._ctx.lib.hive_tsdf_copy_mesh_voxel_coords(self._handle, ptr(vvox)))
            return verts, faces, norms, colors, vvox
        return verts, faces, norms, colors

    def get_point_cloud(self):
        """Extract a point cloud from the voxel volume: (V, 6) [x, y, z, r, g, b]."""
        verts, _, _, colors = self.get_mesh()
        return np.hstack([verts, colors.astype(np.float32)])

    # -- frame-sharded fusion: accumulate -> all-reduce -> finalize -------------
    def accum_reset(self, accum):
        self._ctx.check(self._ctx.lib.hive_tsdf_accum_reset(self._handle, ptr(accum)))

    def accum_integrate(self, accum, color_im, depth_im, cam_intr, cam_pose, obs_weight=1.):
        color, depth, mem = self._frame_args(color_im, depth_im)
        K = np.ascontiguousarray(cam_intr, dtype=np.float32).reshape(3, 3)
        pose = np.ascontiguousarray(cam_pose, dtype=np.float64).reshape(4, 4)
        self._ctx.check(self._ctx.lib.hive_tsdf_accum_integrate(self._handle, ptr(accum), ptr(color), ptr(depth), depth.shape[0],
                                                                depth.shape[1], ptr(K), ptr(pose), float(obs_weight), mem))

    def accum_from_volume(self, accum):
        """planes = [tsdf * w, w, r * w, g * w, b * w] of this volume (a rank's contribution to the all-reduce)."""
        self._ctx.check(self._ctx.lib.hive_tsdf_accum_from_volume(self._handle, ptr(accum)))

    def accum_from_volume_sharded(self, out, world, chunk):
        """The same sums as float32 [world][5][chunk] (voxel i in piece i // chunk): the input of ONE reduce-scatter."""
        self._ctx.follow_torch_stream()
        self._ctx.check(self._ctx.lib.hive_tsdf_accum_from_volume_sharded(self._handle, ptr(out), int(world), int(chunk)))

    def set_volume_range(self, first, count, tsdf=None, weight=None, color=None):
        """Overwrite voxels [first, first + count) from device tensors of ``count`` float32 each (stream-ordered copies)."""
        self._ctx.follow_torch_stream()
        for a in (tsdf, weight, color):
            assert a is None or (a.numel() >= count and str(a.dtype) == "torch.float32" and a.is_contiguous())
        self._ctx.check(self._ctx.lib.hive_tsdf_set_volume_range(self._handle, int(first), int(count), ptr(tsdf), ptr(weight), ptr(color)))

    def accum_finalize(self, accum):
        self._ctx.check(self._ctx.lib.hive_tsdf_accum_finalize(self._handle, ptr(accum)))

    def accum_finalize_range(self, planes, plane_stride, count, outputs):
        """Fold ``count`` voxels of 5 summed planes (``planes[p * plane_stride + i]``) into ``outputs = (tsdf, weight, colour)``
        device tensors at [0, count): a rank's share after the reduce-scatter (``hive_amd.distributed.fuse_sharded``)."""
        self._ctx.follow_torch_stream()
        self._ctx.check(self._ctx.lib.hive_tsdf_accum_finalize_to(self._handle, ptr(planes), int(plane_stride), int(count),
                                                                  ptr(outputs[0]), ptr(outputs[1]), ptr(outputs[2])))

    def close(self):
        if getattr(self, "_handle", None) and _lib.alive():
            self._ctx.lib.hive_tsdf_destroy(self._handle)
        self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------
# Driver: the roles of `adjust_voxel_size` / `tsdf_fusion` (/root/reference/hive/fusion.py:37-134), organised around a frame
# set that is decoded ONCE and then lives in HBM.  The reference walks the dataset twice on the host -- every depth map is
# decoded for the bounds pass (:53-61) and again, with colour and mask, for the integrate loop (:113-124), with one
# `cv2.dilate` x N and one upload per frame.  Here: one decode per frame -> `DeviceFrames`; the bounds of the whole set come
# from one reduction launch (`hive_view_frustum_batch`); masks are dilated and applied to all depth maps in two launches
# (`hive_depth_apply_mask`); `integrate_batch` sweeps the frames in order.  Results are identical to the reference order of
# operations (min / max are exact, masking is elementwise, integration order is kept).
MASK_BACKGROUND, MASK_FOREGROUND = 0, 1


class DeviceFrames:
    """Colour u8 [n, H, W, 3], depth f32 [n, H, W] (metres, 0 = invalid) and optionally masks u8 [n, H, W] of a frame
    set as device tensors, plus the camera-to-world poses f64 [n, 4, 4] (host)."""

    def __init__(self, color, depth, poses_c2w, masks=None):
        assert color.shape[:3] == depth.shape and tuple(color.shape[3:]) == (3,)
        assert masks is None or masks.shape == depth.shape
        self.color, self.depth, self.masks = color, depth, masks
        self.poses = np.ascontiguousarray(poses_c2w, dtype=np.float64).reshape(-1, 4, 4)
        assert len(self.poses) == depth.shape[0]

    def __len__(self):
        return int(self.depth.shape[0])

    @classmethod
    def from_dataset(cls, dataset, frame_set, with_masks, device="cuda", with_color=True, staging=None):
        """Reads frame ``i`` of ``bg_rgb_dataset`` / ``bg_depth_dataset`` (/ ``mask_dataset``) once for every ``i`` in
        ``frame_set``, through pinned host buffers.  Poses: the dataset stores world-to-camera; the volume wants
        camera-to-world (hive/fusion.py:50-51).  ``with_color=False`` loads the depth maps only (the bounds pass of a chunked
        run); ``staging``: a ``_Staging`` to reuse pinned / device buffers between chunks."""
        import torch
        frame_set = list(frame_set)
        assert len(frame_set) > 0, "empty frame set"
        poses = dataset.camera_trajectory.inverse().to_homogenous_transforms()[frame_set]
        first = np.asarray(dataset.bg_depth_dataset[frame_set[0]])
        h, w = first.shape
        n = len(frame_set)
        staging = staging or _Staging(n, h, w, device)
        color_h, depth_h, masks_h = staging.host(n, with_color, with_masks)
        for j, i in enumerate(frame_set):
            depth_h.numpy()[j] = first if j == 0 else dataset.bg_depth_dataset[i]
            if with_color:
                color_h.numpy()[j] = dataset.bg_rgb_dataset[i]
            if with_masks:
                masks_h.numpy()[j] = dataset.mask_dataset[i]
        color, depth, masks = staging.upload(n, with_color, with_masks)
        if not with_color:
            color = torch.empty((n, h, w, 3), dtype=torch.uint8, device="meta")  # shape only: the bounds pass never reads colour
        return cls(color, depth, poses, masks)

    def masked_depth(self, iterations, mode=MASK_BACKGROUND, instance_id=0, ctx=None, dilation_filter=None):
        """Depth maps with the mask applied on the device: MASK_BACKGROUND = `depth[dilate(mask) > 0] = 0`
        (hive/fusion.py:118-121), MASK_FOREGROUND = the complement (only pixels of the undilated mask keep their depth).
        ``dilation_filter``: a structuring element other than the reference's 3x3 box (``MaskDilationOptions.filter``)."""
        import torch
        assert self.masks is not None, "this frame set was loaded without masks"
        ctx = ctx or _lib.default_context(self.depth.device.index or 0)
        out = torch.empty_like(self.depth)
        n, h, w = self.depth.shape
        se = None if dilation_filter is None else MaskDilationOptions(int(iterations), dilation_filter).structuring_element()
        for a in range(0, n, MAX_BATCH_FRAMES):
            b = min(n, a + MAX_BATCH_FRAMES)
            if se is None:
                ctx.check(ctx.lib.hive_depth_apply_mask(ctx.handle, self.depth[a:b].data_ptr(), self.masks[a:b].data_ptr(), b - a, h, w, int(iterations),
                                                        int(mode), int(instance_id), out[a:b].data_ptr()))
            else:
                ctx.check(ctx.lib.hive_depth_apply_mask_se(ctx.handle, self.depth[a:b].data_ptr(), self.masks[a:b].data_ptr(), b - a, h, w, ptr(se),
                                                           se.shape[0], se.shape[1], int(iterations), int(mode), int(instance_id), out[a:b].data_ptr()))
        return out


class _Staging:
    """Pinned host buffers + device buffers for up to ``capacity`` frames, reused from chunk to chunk (a chunked run keeps
    ``capacity`` frames resident at most: 9 bytes per pixel pinned and on the device, whatever the length of the sequence).
    The upload of a chunk is stream-ordered behind the kernels that read the previous chunk from the same device buffers; the
    host buffers are refilled only after that upload has completed (``upload`` records an event, ``host`` waits for it)."""

    def __init__(self, capacity, h, w, device="cuda"):
        import torch
        self.capacity, self.device = int(capacity), device
        self.color_h = None  # pinned lazily, like the masks: the bounds pass of a chunked run (frame_chunks(with_color=False)) never touches colours
        self.depth_h = torch.empty((capacity, h, w), dtype=torch.float32).pin_memory()
        self.masks_h = None
        self.color_d = self.depth_d = self.masks_d = None
        self.copied = None

    def host(self, n, with_color, with_masks):
        import torch
        assert n <= self.capacity
        if self.copied is not None:
            self.copied.synchronize()  # the previous chunk's upload has left the pinned buffers
        if with_masks and self.masks_h is None:
            self.masks_h = torch.empty(tuple(self.depth_h.shape), dtype=torch.uint8).pin_memory()
        if with_color and self.color_h is None:
            self.color_h = torch.empty(tuple(self.depth_h.shape) + (3,), dtype=torch.uint8).pin_memory()
        return (self.color_h[:n] if with_color else None), self.depth_h[:n], (self.masks_h[:n] if with_masks else None)

    def upload(self, n, with_color, with_masks):
        import torch

        def put(host, attr):
            dev = getattr(self, attr)
            if dev is None:
                dev = torch.empty(tuple(host.shape), dtype=host.dtype, device=self.device)
                setattr(self, attr, dev)
            dev[:n].copy_(host[:n], non_blocking=True)
            return dev[:n]
        color = put(self.color_h, "color_d") if with_color else None
        depth = put(self.depth_h, "depth_d")
        masks = put(self.masks_h, "masks_d") if with_masks else None
        self.copied = torch.cuda.Event()
        self.copied.record()
        return color, depth, masks


def chunk_staging(dataset, frame_set, chunk_frames):
    """The staging set of a chunked run over ``frame_set``: shared by its bounds pass and its integration pass (one set of pinned and
    device buffers for the whole run: 9 bytes per pixel and resident frame once the colours and masks are in use)."""
    frame_set = list(frame_set)
    h, w = np.asarray(dataset.bg_depth_dataset[frame_set[0]]).shape
    return _Staging(min(chunk_frames, len(frame_set)), h, w)


def frame_chunks(dataset, frame_set, with_masks, chunk_frames, with_color=True, staging=None):
    """The frame set as ``DeviceFrames`` of at most ``chunk_frames`` frames each, in order, through ONE reused staging set
    (``staging``: the caller's, shared between passes; default: a set of this generator's own)."""
    frame_set = list(frame_set)
    for a in range(0, len(frame_set), chunk_frames):
        ids = frame_set[a:a + chunk_frames]
        if staging is None:
            staging = chunk_staging(dataset, frame_set, chunk_frames)
        yield DeviceFrames.from_dataset(dataset, ids, with_masks, with_color=with_color, staging=staging)


MAX_BATCH_FRAMES = 32768  # hive_view_frustum_batch / hive_depth_apply_mask take the frame index from a 16-bit grid dimension


def view_frusta(depth_ims, cam_intr, cam_poses, ctx=None):
    """``get_view_frustum`` for n frames at once -> (n, 3, 5) float64: one reduction launch, one read-back."""
    ctx = ctx or _lib.default_context()
    if _is_torch(depth_ims):
        depth, mem = depth_ims.contiguous(), MEM_DEVICE
        assert str(depth.dtype) == "torch.float32"
    else:
        depth, mem = np.ascontiguousarray(depth_ims, dtype=np.float32), MEM_HOST
    assert depth.ndim == 3, "depth_ims must be (n, H, W)"
    n, h, w = (int(v) for v in depth.shape)
    K = np.ascontiguousarray(cam_intr, dtype=np.float32).reshape(3, 3)
    poses = np.ascontiguousarray(cam_poses, dtype=np.float64).reshape(n, 4, 4)
    out = np.empty((n, 3, 5), np.float64)
    for a in range(0, n, MAX_BATCH_FRAMES):
        b = min(n, a + MAX_BATCH_FRAMES)
        ctx.check(ctx.lib.hive_view_frustum_batch(ctx.handle, ptr(depth[a:b]), b - a, h, w, ptr(K), ptr(poses[a:b]), mem, ptr(out[a:b])))
    return out


def scene_bounds(frames, cam_intr, ctx=None) -> np.ndarray:
    """Axis-aligned bounds of the union of the view frusta of a frame set and the world origin -- the reference starts
    from `zeros((3, 2))`, so the origin is always inside (hive/fusion.py:48, 60-61).  ``frames``: a ``DeviceFrames`` or an
    iterable of them (the chunks of a long sequence: a running min / max, exact in any order)."""
    lo, hi = np.zeros(3), np.zeros(3)
    for chunk in ([frames] if isinstance(frames, DeviceFrames) else frames):
        corners = view_frusta(chunk.depth, cam_intr, chunk.poses, ctx)  # (n, 3, 5)
        lo = np.minimum(lo, corners.min(axis=(0, 2)))
        hi = np.maximum(hi, corners.max(axis=(0, 2)))
    return np.stack([lo, hi], axis=1)


def voxel_size_for_budget(vol_bnds, options) -> float:
    """`sdf_voxel_size`, or the size at which the volume has exactly `sdf_max_voxels` voxels when that would be exceeded
    (hive/fusion.py:66-74)."""
    extent = vol_bnds[:, 1] - vol_bnds[:, 0]
    wanted = np.ceil(np.prod(extent / options.sdf_voxel_size))
    if options.sdf_max_voxels and wanted > options.sdf_max_voxels:
        voxel_size = (np.prod(extent) / options.sdf_max_voxels) ** (1 / 3)
        logging.info("voxel size %.4f m instead of %s m: %.0f voxels would exceed the budget of %d", voxel_size, options.sdf_voxel_size,
                     wanted, options.sdf_max_voxels)
        return voxel_size
    return options.sdf_voxel_size


def adjust_voxel_size(dataset, options, frame_set: List[int], frames: Optional[DeviceFrames] = None) -> Tuple[float, np.ndarray]:
    """(voxel size, scene bounds (3, 2)) for a frame set -- signature of /root/reference/hive/fusion.py:37.
    ``frames``: the already loaded frame set (``tsdf_fusion`` passes it so that nothing is decoded twice)."""
    if frames is None:
        frames = DeviceFrames.from_dataset(dataset, frame_set, with_masks=False)
    vol_bnds = scene_bounds(frames, dataset.camera_matrix)
    return voxel_size_for_budget(vol_bnds, options), vol_bnds


def _resolve_frames(dataset, num_frames, frame_set):
    if num_frames == -1:
        num_frames = dataset.num_frames
    return list(range(num_frames)) if frame_set is None else list(frame_set)


# Frames kept resident at once by the drivers below.  Up to this many, the whole set is decoded ONCE and lives in HBM (bounds,
# masking and integration all read it there).  Longer sequences stream through one reused staging set of this size in two passes,
# as the reference walks its dataset twice (hive/fusion.py:53-61 and :113-124): 9 bytes per pixel pinned + 13 on the device per
# resident frame, whatever the sequence length (2000 frames of 1080p no longer need 35 GB pinned + 50 GB of HBM).
CHUNK_FRAMES = 256


def tsdf_fusion(dataset, options=None, num_frames=-1, frame_set: Optional[List[int]] = None, return_volume=False, chunk_frames=None):
    """Static-scene reconstruction of a dataset (/root/reference/hive/fusion.py:79-134): bounds -> voxel size -> volume ->
    every frame of the set, dynamic objects masked out of the depth unless the dataset carries inpainted frames -> mesh.

    ``dataset`` needs what the reference reads: ``num_frames``, ``camera_trajectory``, ``camera_matrix``, ``bg_rgb_dataset``,
    ``bg_depth_dataset``, ``mask_dataset``, ``has_inpainted_frame_data``.  Returns a ``trimesh.Trimesh`` when trimesh is
    installed, otherwise a ``hive_amd.mesh.Mesh`` with the same ``vertices / faces / vertex_normals / visual.vertex_colors``.
    """
    from hive_amd.mesh import make_mesh
    from hive_amd.options import BackgroundMeshOptions

    options = options or BackgroundMeshOptions()
    frame_set = _resolve_frames(dataset, num_frames, frame_set)
    needs_masks = not dataset.has_inpainted_frame_data
    chunk_frames = int(chunk_frames or CHUNK_FRAMES)
    if len(frame_set) <= chunk_frames:  # the whole set resident: one decode per frame
        frames = DeviceFrames.from_dataset(dataset, frame_set, with_masks=needs_masks)
        voxel_size, volume_bounds = adjust_voxel_size(dataset, options, frame_set, frames=frames)
        chunks = [frames]
    else:  # pass 1: depth maps only, running bounds; pass 2: everything, chunk by chunk, in sequence order
        staging = chunk_staging(dataset, frame_set, chunk_frames)  # one staging set for both passes
        volume_bounds = scene_bounds(frame_chunks(dataset, frame_set, False, chunk_frames, with_color=False, staging=staging), dataset.camera_matrix)
        voxel_size = voxel_size_for_budget(volume_bounds, options)
        chunks = frame_chunks(dataset, frame_set, needs_masks, chunk_frames, staging=staging)
    tsdf_vol = TSDFVolume(volume_bounds, voxel_size=voxel_size)
    for frames in chunks:
        depth = frames.masked_depth(options.depth_mask_dilation_iterations, MASK_BACKGROUND) if needs_masks else frames.depth
        tsdf_vol.integrate_batch(frames.color, depth, dataset.camera_matrix, frames.poses, obs_weight=1.)
    verts, faces, norms, colors = tsdf_vol.get_mesh()
    mesh = make_mesh(vertices=verts, faces=faces, vertex_colors=colors, vertex_normals=norms)
    return (mesh, tsdf_vol) if return_volume else mesh


def tsdf_fusion_fg_bg(dataset, options=None, num_frames=-1, frame_set: Optional[List[int]] = None, instance_id=0, chunk_frames=None, vol_bnds=None):
    """Dynamic-object variant (BASELINE config 5): two volumes over the same bounds and voxel size from one resident frame
    set -- background = depth with the dilated instance masks zeroed (exactly ``tsdf_fusion``'s volume, from
    ``dataset.depth_dataset`` / ``rgb_dataset``), foreground = the complement (depth kept only on the undilated masks, or on
    one ``instance_id``).  Returns ``{"bg": TSDFVolume, "fg": TSDFVolume}``; call ``get_mesh()`` on either (the foreground
    raises ``ValueError`` if no object pixel ever produced a surface)."""
    from hive_amd.options import BackgroundMeshOptions

    options = options or BackgroundMeshOptions()
    frame_set = _resolve_frames(dataset, num_frames, frame_set)
    raw = _RawFrames(dataset)
    chunk_frames = int(chunk_frames or CHUNK_FRAMES)
    given_bounds = vol_bnds is not None  # (frame-sharded runs: the bounds of the WHOLE frame set, hive_amd.distributed.tsdf_fusion_fg_bg_sharded)
    if len(frame_set) == 0:
        assert given_bounds, "an empty frame set needs the scene's bounds"
        chunks = []
    elif len(frame_set) <= chunk_frames:
        chunks = [DeviceFrames.from_dataset(raw, frame_set, with_masks=True)]
        if not given_bounds:
            vol_bnds = scene_bounds(chunks[0], dataset.camera_matrix)
    else:
        staging = chunk_staging(raw, frame_set, chunk_frames)
        if not given_bounds:
            vol_bnds = scene_bounds(frame_chunks(raw, frame_set, False, chunk_frames, with_color=False, staging=staging), dataset.camera_matrix)
        chunks = frame_chunks(raw, frame_set, True, chunk_frames, staging=staging)
    vol_bnds = np.asarray(vol_bnds, np.float64)
    voxel_size = voxel_size_for_budget(vol_bnds, options)
    modes = (("bg", MASK_BACKGROUND, options.depth_mask_dilation_iterations), ("fg", MASK_FOREGROUND, 0))
    volumes = {name: TSDFVolume(vol_bnds, voxel_size=voxel_size) for name, _, _ in modes}
    for frames in chunks:  # both volumes from each resident chunk, in sequence order
        for name, mode, iterations in modes:
            depth = frames.masked_depth(iterations, mode, instance_id if mode == MASK_FOREGROUND else 0)
            volumes[name].integrate_batch(frames.color, depth, dataset.camera_matrix, frames.poses, obs_weight=1.)
    return volumes


class _RawFrames:
    """View of a dataset whose bg_* accessors are the captured (not inpainted) frames: the dynamic path masks them itself."""

    def __init__(self, dataset):
        self.camera_trajectory = dataset.camera_trajectory
        self.bg_rgb_dataset = dataset.rgb_dataset
        self.bg_depth_dataset = dataset.depth_dataset
        self.mask_dataset = dataset.mask_dataset
