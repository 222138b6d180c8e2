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


class TSDFVolume:
    """Volumetric TSDF Fusion of RGB-D Images, resident in MI355X HBM.

    Same constructor and methods as the reference library's class (``use_gpu`` is accepted and
    ignored: there is only the GPU path, and it fails loudly without a device).
    """

    def __init__(self, vol_bnds, voxel_size, use_gpu=True, ctx=None, round_mode=None, storage=None):
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
        self._ctx.check(lib.hive_tsdf_create(self._ctx.handle, ptr(bnds), self._voxel_size, ptr(s[0]), ptr(s[1]), ptr(s[2]),
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

    def device_ptrs(self):
        a, b, c = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        self._ctx.check(self._ctx.lib.hive_tsdf_device_ptrs(self._handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    def _extract(self):
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

    def accum_finalize(self, accum):
        self._ctx.check(self._ctx.lib.hive_tsdf_accum_finalize(self._handle, ptr(accum)))

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
# Driver functions of /root/reference/hive/fusion.py:37-134
def adjust_voxel_size(dataset, options, frame_set: List[int]) -> Tuple[float, np.ndarray]:
    """Calculate the scene bounds and adjust voxel size to keep within the specified budget
    (/root/reference/hive/fusion.py:37-76).  Bounds always contain the world origin (:48)."""
    logging.info("Estimating voxel volume bounds...")
    vol_bnds = np.zeros((3, 2))
    # poses are world-to-cam on disk; the TSDF volume expects cam-to-world (fusion.py:50-51)
    camera_trajectory = dataset.camera_trajectory.inverse().to_homogenous_transforms()

    for i in frame_set:
        depth_im = dataset.bg_depth_dataset[i]
        cam_pose = camera_trajectory[i]
        view_frust_pts = get_view_frustum(depth_im, dataset.camera_matrix, cam_pose)
        vol_bnds[:, 0] = np.minimum(vol_bnds[:, 0], np.amin(view_frust_pts, axis=1))
        vol_bnds[:, 1] = np.maximum(vol_bnds[:, 1], np.amax(view_frust_pts, axis=1))

    voxel_count = np.ceil(np.prod((vol_bnds[:, 1] - vol_bnds[:, 0]) / options.sdf_voxel_size))

    if options.sdf_max_voxels and voxel_count > options.sdf_max_voxels:
        voxel_size = (np.prod(vol_bnds[:, 1] - vol_bnds[:, 0]) / options.sdf_max_voxels) ** (1 / 3)
        logging.info(f"Increasing voxel size to {voxel_size:.3f}: Using a voxel size of {options.sdf_voxel_size} would "
                     f"result in {voxel_count:,.0f} voxels, which is above the specified limit of "
                     f"{options.sdf_max_voxels:,d}.")
    else:
        voxel_size = options.sdf_voxel_size

    return voxel_size, vol_bnds


def tsdf_fusion(dataset, options=None, num_frames=-1, frame_set: Optional[List[int]] = None, return_volume=False):
    """Run TSDF fusion on a dataset (/root/reference/hive/fusion.py:79-134).

    ``dataset`` needs the attributes the reference reads: ``num_frames``, ``camera_trajectory``,
    ``camera_matrix``, ``bg_rgb_dataset``, ``bg_depth_dataset``, ``mask_dataset``,
    ``has_inpainted_frame_data``.  Returns a ``trimesh.Trimesh`` when trimesh is installed, otherwise
    a ``hive_amd.mesh.Mesh`` with the same ``vertices / faces / vertex_normals / visual.vertex_colors``.
    """
    from hive_amd.image_processing import dilate_mask
    from hive_amd.mesh import make_mesh
    from hive_amd.options import BackgroundMeshOptions, MaskDilationOptions

    if options is None:
        options = BackgroundMeshOptions()

    if num_frames == -1:
        num_frames = dataset.num_frames

    if frame_set is None:
        frame_set = range(num_frames)

    mask_dilation_options = MaskDilationOptions(num_iterations=options.depth_mask_dilation_iterations)

    voxel_size, volume_bounds = adjust_voxel_size(dataset=dataset, options=options, frame_set=frame_set)
    logging.info("Initializing voxel volume...")
    tsdf_vol = TSDFVolume(volume_bounds, voxel_size=voxel_size)

    logging.info("Fusing frames...")
    has_inpainted_frame_data = dataset.has_inpainted_frame_data
    camera_trajectory = dataset.camera_trajectory.inverse().to_homogenous_transforms()

    for i in frame_set:
        color_image = dataset.bg_rgb_dataset[i]
        depth_im = dataset.bg_depth_dataset[i]
        cam_pose = camera_trajectory[i]

        if not has_inpainted_frame_data:
            mask = dataset.mask_dataset[i]
            mask = dilate_mask(mask, mask_dilation_options)
            depth_im[mask > 0] = 0.0

        tsdf_vol.integrate(color_image, depth_im, dataset.camera_matrix, cam_pose, obs_weight=1.)

    verts, faces, norms, colors = tsdf_vol.get_mesh()
    mesh = make_mesh(vertices=verts, faces=faces, vertex_colors=colors, vertex_normals=norms)

    if return_volume:
        return mesh, tsdf_vol
    return mesh
