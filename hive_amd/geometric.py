"""Depth <-> point-cloud geometry of HIVE's hot path on the MI355X, plus the pose containers that
feed it.

Kernel-backed (float64 HIP kernels behind the C ABI, results checked against golden vectors from
the real module):
  ``point_cloud_from_depth``, ``point_cloud_from_rgbd``, ``image2world``, ``world2image``
  -- /root/reference/hive/geometric.py:107-206.
Host-side (tiny N x 7 / N x 4 x 4 arrays; SURVEY.md §8 a-12 keeps them in Python/scipy):
  ``pose_vec2mat``, ``pose_mat2vec``, ``get_pose_components``, ``Trajectory``, ``CameraMatrix``
  -- /root/reference/hive/geometric.py:34-78, 302-648, 651-737.
"""
import ctypes
import dataclasses
from typing import Dict, Tuple

import numpy as np
from scipy.interpolate import interp1d
from scipy.spatial.transform import Rotation, Slerp

from hive_amd import _lib
from hive_amd._lib import MEM_HOST, ptr
from hive_amd.utils import validate_camera_parameter_shapes, validate_shape


# ------------------------------------------------------------------------------------------------
# kernel-backed functions
def _kinv(K):
    # the reference inverts K in K's own dtype (float32 when loaded from disk) -- geometric.py:203
    return np.ascontiguousarray(np.linalg.inv(K), dtype=np.float64)


def _unproject(depth, mask, K, R, t, rgb=None):
    ctx = _lib.default_context()
    depth_f = np.ascontiguousarray(depth, dtype=np.float32)
    validate_shape(depth_f, 'depth', (None, None))
    validate_camera_parameter_shapes(K, R, t)
    H, W = depth_f.shape
    mask_u8 = None if mask is None else np.ascontiguousarray(np.asarray(mask).astype(bool), dtype=np.uint8)
    if mask_u8 is not None:
        assert mask_u8.shape == depth_f.shape, "mask and depth must have the same shape"
    rgb_u8 = None if rgb is None else np.ascontiguousarray(rgb, dtype=np.uint8)
    Kinv = _kinv(K)
    R64 = np.ascontiguousarray(R, dtype=np.float64)
    t64 = np.ascontiguousarray(t, dtype=np.float64).reshape(3)
    capacity = H * W
    xyz = np.empty((capacity, 3), np.float64)
    rgba = None if rgb is None else np.empty((capacity, 4), np.uint8)
    n = ctypes.c_int64(0)
    ctx.check(ctx.lib.hive_unproject(ctx.handle, ptr(depth_f), ptr(mask_u8), ptr(rgb_u8), H, W, ptr(Kinv), ptr(R64), ptr(t64),
                                     MEM_HOST, ptr(xyz), ptr(rgba), capacity, ctypes.byref(n)))
    return xyz[:n.value], (None if rgba is None else rgba[:n.value])


def point_cloud_from_depth(depth, mask, K, R=np.eye(3), t=np.zeros((3, 1))):
    """Create a point cloud from a depth map (/root/reference/hive/geometric.py:107-126).

    :param depth: A depth map (H, W).
    :param mask: A binary mask of the same shape; truthy values are kept.
    :param K, R, t: intrinsics (3, 3), world-to-camera rotation (3, 3) and translation (3, 1).
    :return: the (N, 3) float64 point cloud, ordered row-major (v, u) like ``np.nonzero``.
    """
    points, _ = _unproject(depth, mask, K, R, t)
    return points


def point_cloud_from_rgbd(rgb, depth, mask, K, R=np.eye(3), t=np.zeros((3, 1))):
    """Point cloud with per-vertex RGBA (alpha 255) from an RGB-D frame (geometric.py:129-152)."""
    rgb = np.asarray(rgb)
    points, rgba = _unproject(depth, mask, K, R, t, rgb=rgb)
    return points, rgba.astype(rgb.dtype, copy=False)


def image2world(points, depth, K, R=np.eye(3), t=np.zeros((3, 1)), scale_factor=1.0):
    """2D image coordinates + depth -> 3D world coordinates (geometric.py:183-206):
    ``X = R^T (d * K^-1 [u*s, v*s, 1]^T - t)``, float64."""
    points = np.asarray(points)
    depth = np.asarray(depth)
    validate_shape(points, 'points', expected_shape=(None, 2))
    validate_shape(depth, 'depth', expected_shape=(points.shape[0],))
    validate_camera_parameter_shapes(K, R, t)
    ctx = _lib.default_context()
    n = points.shape[0]
    uv = np.ascontiguousarray(points, dtype=np.float64)
    d64 = np.ascontiguousarray(depth, dtype=np.float64)
    out = np.empty((n, 3), np.float64)
    ctx.check(ctx.lib.hive_image2world(ctx.handle, ptr(uv), ptr(d64), n, ptr(_kinv(K)), ptr(np.ascontiguousarray(R, dtype=np.float64)),
                                       ptr(np.ascontiguousarray(t, dtype=np.float64).reshape(3)), float(scale_factor), MEM_HOST,
                                       ptr(out)))
    return out


def world2image(points, K, R=np.eye(3), t=np.zeros((3, 1)), scale_factor=1.0, dtype=np.int32):
    """3D world coordinates -> 2D image coordinates and depth (geometric.py:155-180).
    Integer dtypes are rounded half-to-even (np.round) before the cast."""
    points = np.ascontiguousarray(points, dtype=np.float64)
    validate_shape(points, 'points', expected_shape=(None, 3))
    validate_camera_parameter_shapes(K, R, t)
    ctx = _lib.default_context()
    n = points.shape[0]
    integer = issubclass(dtype, np.integer)
    uv_i = np.empty((n, 2), np.int32) if integer else None
    uv_f = None if integer else np.empty((n, 2), np.float64)
    depth = np.empty(n, np.float64)
    ctx.check(ctx.lib.hive_project(ctx.handle, ptr(points), n, ptr(np.ascontiguousarray(K, dtype=np.float64)),
                                   ptr(np.ascontiguousarray(R, dtype=np.float64)),
                                   ptr(np.ascontiguousarray(t, dtype=np.float64).reshape(3)), float(scale_factor), MEM_HOST,
                                   ptr(uv_i), ptr(uv_f), ptr(depth)))
    pixel_coords = uv_i if integer else uv_f
    return np.array(pixel_coords, dtype=dtype), depth


# ------------------------------------------------------------------------------------------------
# host-side pose helpers
def pose_vec2mat(pose: np.ndarray) -> np.ndarray:
    """7-vector [quaternion xyzw, t] -> (4, 4) homogeneous transform (geometric.py:34-49)."""
    validate_shape(pose, 'pose', expected_shape=(7,))
    rotation = Rotation.from_quat(pose[:4]).as_matrix()
    M = np.eye(4, dtype=rotation.dtype)
    M[:3, :3] = rotation
    M[:3, 3] = pose[4:]
    return M


def pose_mat2vec(pose: np.ndarray) -> np.ndarray:
    """(4, 4) homogeneous transform -> 7-vector [quaternion xyzw, t] (geometric.py:52-63)."""
    validate_shape(pose, 'pose', expected_shape=(4, 4))
    return np.hstack((Rotation.from_matrix(pose[:3, :3]).as_quat(), pose[:3, 3]))


def get_pose_components(pose):
    """(4, 4) pose -> (R (3, 3), t (3, 1)) (geometric.py:66-78)."""
    validate_shape(pose, 'pose', (4, 4))
    return pose[:3, :3], pose[:3, 3:]


def add_pose(pose_a, pose_b) -> np.ndarray:
    return pose_mat2vec(pose_vec2mat(pose_b) @ pose_vec2mat(pose_a))


def subtract_pose(pose_a, pose_b) -> np.ndarray:
    return pose_mat2vec(np.linalg.inv(pose_vec2mat(pose_b)) @ pose_vec2mat(pose_a))


def get_identity_pose():
    return np.asarray([0., 0., 0., 1., 0., 0., 0.])


class Quaternion:
    """Batched scalar-last quaternions as a (4, N) torch tensor with rows x, y, z, w -- the small algebra the
    reference's pose optimiser uses and its only unit tests cover (/root/reference/hive/geometric.py:209-299,
    tests/quaternion.py).  Host-side helper, not part of the dense-compute path."""

    def __init__(self, values):
        if len(values.shape) != 2 or values.shape[0] != 4:
            raise ValueError(f"Invalid shape. Expected shape (4, N) but got {values.shape}.")
        self.values = values

    x = property(lambda self: self.values[0])
    y = property(lambda self: self.values[1])
    z = property(lambda self: self.values[2])
    w = property(lambda self: self.values[3])

    def conjugate(self) -> 'Quaternion':
        import torch
        return Quaternion(torch.vstack((-self.x, -self.y, -self.z, self.w)))

    inverse = conjugate  # for unit quaternions

    def normalise(self) -> 'Quaternion':
        import torch
        return Quaternion(self.values / torch.linalg.norm(self.values, ord=2, dim=0))

    @staticmethod
    def multiply(q1: 'Quaternion', q2: 'Quaternion') -> 'Quaternion':
        """Hamilton product q1 q2, component-wise over the batch."""
        import torch
        x1, y1, z1, w1 = q1.values
        x2, y2, z2, w2 = q2.values
        return Quaternion(torch.vstack((w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                                        w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2,
                                        w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2,
                                        w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2)))

    def __mul__(self, other):
        if not isinstance(other, Quaternion):
            raise TypeError(f"Cannot multiply a {self.__class__.__name__} with a {type(other)}")
        return Quaternion.multiply(self, other)

    __rmul__ = __mul__

    def apply(self, v):
        """Rotate the (3, N) vectors: q (v, 0) q*."""
        import torch
        assert len(v.shape) == 2 and v.shape[0] == 3
        pure = Quaternion(torch.vstack((v, torch.zeros(v.shape[1], dtype=v.dtype, device=v.device))))
        return (self * pure * self.conjugate()).values[:3, :]

    def __repr__(self):
        return f"{self.__class__.__name__}({repr(self.values)})"


class Trajectory:
    """A sequence of camera poses, N x 7 rows of [scalar-last quaternion, xyz position]
    (/root/reference/hive/geometric.py:302-648)."""

    # The container's public names are the reference's (its callers index, slice, iterate and copy trajectories): each is a view of `values`.
    _QUAT, _XYZ = slice(0, 4), slice(4, 7)

    def __init__(self, values=None):
        if values is not None:
            validate_shape(values, 'values', (None, 7))
        self.values = values

    def __len__(self):
        return self.values.shape[0]

    def __iter__(self):
        yield from self.values

    def __getitem__(self, index):
        return self.values[index]

    def __setitem__(self, index, value):
        self.values[index] = value

    rotations = property(lambda self: self.values[:, Trajectory._QUAT], doc="(N, 4) scalar-last quaternions (a view)")
    positions = property(lambda self: self.values[:, Trajectory._XYZ], doc="(N, 3) camera positions (a view)")
    shape = property(lambda self: tuple(self.values.shape))

    def copy(self) -> 'Trajectory':
        return type(self)(np.array(self.values, copy=True))

    def save(self, f):
        np.savetxt(f, self.values)

    @classmethod
    def load(cls, f) -> 'Trajectory':
        rows = np.atleast_2d(np.loadtxt(f, dtype=np.float32))  # (a one-pose file loads as a vector)
        return cls(rows)

    def to_homogenous_transforms(self) -> np.ndarray:
        """(N, 7) -> (N, 4, 4) float64 camera matrices [R | t; 0 0 0 1]."""
        out = np.zeros((len(self), 4, 4), dtype=np.float64)
        out[:, 3, 3] = 1.0
        out[:, :3, :3] = Rotation.from_quat(self.rotations).as_matrix()
        out[:, :3, 3] = self.positions
        return out

    @staticmethod
    def from_homogenous_transforms(camera_trajectory: np.ndarray) -> 'Trajectory':
        validate_shape(camera_trajectory, 'camera_trajectory', (None, 4, 4))
        quaternions = Rotation.from_matrix(camera_trajectory[:, :3, :3]).as_quat()
        return Trajectory(np.hstack((quaternions, camera_trajectory[:, :3, 3])))

    def inverse(self) -> 'Trajectory':
        return self.from_homogenous_transforms(np.linalg.inv(self.to_homogenous_transforms()))

    def normalise(self) -> 'Trajectory':
        """Re-express the trajectory relative to its first pose (which becomes the identity)."""
        M = self.to_homogenous_transforms()
        M = np.linalg.inv(M[0]) @ M
        M[0] = np.eye(4, dtype=M.dtype)
        return self.from_homogenous_transforms(M)

    def normalise_position(self) -> 'Trajectory':
        """Shift the trajectory so that the first pose sits at the origin (rotation untouched)."""
        M = self.to_homogenous_transforms()
        first = M[0].copy()
        first[:3, :3] = np.eye(3)
        return self.from_homogenous_transforms(np.linalg.inv(first) @ M)

    def apply(self, transform: np.ndarray) -> 'Trajectory':
        return self.from_homogenous_transforms(self.to_homogenous_transforms() @ transform)

    def scale_trajectory(self, scale_factor: float) -> 'Trajectory':
        scaled = self.values.copy()
        scaled[:, -3:] *= scale_factor
        return Trajectory(scaled)

    def tensor(self):
        import torch
        return torch.from_numpy(self.values).to(torch.float32)

    def calculate_ate(self, other: 'Trajectory') -> np.ndarray:
        if len(self) != len(other):
            raise RuntimeError(f"Got trajectories of unequal length ({len(self)} and {len(other)})")
        a = self.normalise().positions
        b = other.normalise().positions
        scale = np.sum(a * b) / np.sum(np.square(b))
        return b * scale - a

    def calculate_rpe(self, other: 'Trajectory') -> Tuple[np.ndarray, np.ndarray]:
        if len(self) != len(other):
            raise RuntimeError(f"Got trajectories of unequal length ({len(self)} and {len(other)})")
        gt = self.normalise().to_homogenous_transforms()
        pred = other.normalise().to_homogenous_transforms()
        rotational, translational = [], []
        for i in range(len(self) - 1):
            rel_est = np.linalg.inv(pred[i]) @ pred[i + 1]
            rel_gt = np.linalg.inv(gt[i]) @ gt[i + 1]
            err = np.linalg.inv(rel_gt) @ rel_est
            translational.append(np.linalg.norm(err[:3, 3]))
            rotational.append(np.arccos(min(1, max(-1, (np.trace(err[:3, :3]) - 1) / 2))))
        return np.asarray(rotational), np.asarray(translational)

    @staticmethod
    def create_by_interpolating(poses: Dict[int, np.ndarray], frame_count: int) -> 'Trajectory':
        """Fill in the frames between known poses: slerp for rotations, lerp for positions."""
        if 0 not in poses:
            raise RuntimeError("Cannot interpolate trajectory where the pose for the first frame is missing.")
        if frame_count - 1 not in poses:
            raise RuntimeError("Cannot interpolate trajectory where the pose for the last frame is missing.")
        known = sorted(poses.keys())
        out = np.zeros((frame_count, 7))
        for start, end in zip(known[:-1], known[1:]):
            times = np.linspace(0, 1, num=end + 1 - start)
            slerp = Slerp(times=[0, 1], rotations=Rotation.from_quat([poses[start][:4], poses[end][:4]]))
            lerp = interp1d([0, 1], [poses[start][4:], poses[end][4:]], axis=0)
            out[start:end + 1, 4:] = lerp(times)
            out[start:end + 1, :4] = slerp(times).as_quat()
        return Trajectory(out)


@dataclasses.dataclass(frozen=True)
class CameraMatrix:
    """A 3x3 pinhole camera matrix with its sensor resolution (geometric.py:651-737)."""
    fx: float
    fy: float
    cx: float
    cy: float
    width: int
    height: int

    @property
    def fov_y(self) -> float:
        return 2.0 * np.arctan(self.height / (2.0 * self.fy))

    @property
    def aspect_ratio(self) -> float:
        return self.width / self.height

    @property
    def matrix(self) -> np.ndarray:
        return np.array([[self.fx, 0., self.cx], [0., self.fy, self.cy], [0., 0., 1.]])

    def transpose(self) -> 'CameraMatrix':
        return CameraMatrix(fx=self.fy, fy=self.fx, cx=self.cy, cy=self.cx, width=self.height, height=self.width)

    def scale(self, target_size) -> 'CameraMatrix':
        target_height, target_width = target_size
        sx, sy = target_width / self.width, target_height / self.height
        return CameraMatrix(fx=self.fx * sx, fy=self.fy * sy, cx=self.cx * sx, cy=self.cy * sy, width=target_width,
                            height=target_height)

    @classmethod
    def from_matrix(cls, matrix: np.ndarray, size) -> 'CameraMatrix':
        validate_shape(matrix, 'matrix', (3, 3))
        height, width = size
        return CameraMatrix(fx=matrix[0, 0], fy=matrix[1, 1], cx=matrix[0, 2], cy=matrix[1, 2], width=width, height=height)
