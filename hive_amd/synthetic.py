"""Seeded synthetic RGB-D sequences for the benchmark, the smoke test and the parity tests
(SURVEY.md §8d): a box room ray-cast analytically from a camera on a circular trajectory.

Nothing here is taken from a dataset; Kinect intrinsics are the reference's
(/root/reference/hive/sensor.py:27, dataset_adaptors.py:579-582).
"""
import numpy as np

KINECT_K = np.array([[580.0, 0.0, 319.5], [0.0, 580.0, 239.5], [0.0, 0.0, 1.0]], dtype=np.float32)


def scaled_intrinsics(height, width):
    """Kinect intrinsics rescaled to (height, width) like CameraMatrix.scale (geometric.py:699-717)."""
    K = KINECT_K.astype(np.float64).copy()
    K[0, :] *= width / 640.0
    K[1, :] *= height / 480.0
    return K.astype(np.float32)


def circular_trajectory(num_frames, volume_size=5.12, radius=1.0, yaw_step_deg=2.4):
    """Camera-to-world 4x4 float64 poses: camera on a circle of `radius` about the volume centre,
    looking inwards through the centre at the far wall (x right, y down, z forward)."""
    c = volume_size / 2.0
    poses = np.tile(np.eye(4), (num_frames, 1, 1))
    for i in range(num_frames):
        th = np.deg2rad(yaw_step_deg * i)
        fwd = np.array([np.sin(th), 0.0, np.cos(th)])
        right = np.array([np.cos(th), 0.0, -np.sin(th)])
        down = np.array([0.0, 1.0, 0.0])
        poses[i, :3, 0] = right
        poses[i, :3, 1] = down
        poses[i, :3, 2] = fwd
        poses[i, :3, 3] = np.array([c, c, c]) - radius * fwd
    return poses


def raycast_room_depth(pose_c2w, K, height, width, room_lo=0.32, room_hi=4.80):
    """z-depth (float32 metres) of the inside of the axis-aligned box [room_lo, room_hi]^3, and hit points."""
    K = np.asarray(K, np.float64)
    u, v = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
    d_c = np.stack([(u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], np.ones_like(u)], axis=-1)
    d_w = d_c @ pose_c2w[:3, :3].T
    o = pose_c2w[:3, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        t_hi = np.where(d_w > 0, (room_hi - o) / d_w, np.inf)
        t_lo = np.where(d_w < 0, (room_lo - o) / d_w, np.inf)
    t = np.minimum(t_hi, t_lo).min(axis=-1)
    return t.astype(np.float32), (o + t[..., None] * d_w)


def room_colour(points_w, rng, room_size=5.12):
    """Smooth low-frequency pattern of the hit point + noise -> uint8 RGB."""
    p = points_w / room_size * 2.0 * np.pi
    r = 127.5 + 100.0 * np.sin(p[..., 0] * 2.0) * np.cos(p[..., 1])
    g = 127.5 + 100.0 * np.sin(p[..., 1] * 3.0 + 1.0) * np.cos(p[..., 2])
    b = 127.5 + 100.0 * np.sin(p[..., 2] * 2.0 + 2.0) * np.cos(p[..., 0] * 3.0)
    rgb = np.stack([r, g, b], axis=-1) + rng.normal(0.0, 4.0, size=p.shape[:-1] + (3,))
    return np.clip(np.rint(rgb), 0, 255).astype(np.uint8)


def make_sequence(num_frames=150, height=480, width=640, volume_size=5.12, room_margin=0.32, seed=1234,
                  invalid_fraction=0.02, yaw_step_deg=2.4):
    """The room is the box [room_margin, volume_size - room_margin]^3, so its walls lie strictly inside the
    TSDF volume [0, volume_size]^3 (`room_bounds`) and marching cubes finds them.
    Returns dict(color u8 [T,H,W,3], depth f32 [T,H,W] metres, K f32 [3,3], poses f64 [T,4,4] cam-to-world)."""
    rng = np.random.default_rng(seed)
    K = scaled_intrinsics(height, width)
    poses = circular_trajectory(num_frames, volume_size, yaw_step_deg=yaw_step_deg)
    color = np.empty((num_frames, height, width, 3), np.uint8)
    depth = np.empty((num_frames, height, width), np.float32)
    for i in range(num_frames):
        d, pts = raycast_room_depth(poses[i], K, height, width, room_margin, volume_size - room_margin)
        color[i] = room_colour(pts, rng, volume_size)
        if invalid_fraction > 0:
            d = d.copy()
            d[rng.random(d.shape) < invalid_fraction] = 0.0
        depth[i] = d
    return {"color": color, "depth": depth, "K": K, "poses": poses}


def ellipse_masks(num_frames, height, width, num_objects=3, seed=1234):
    """Instance masks of BASELINE config 5: 1-3 ellipses drifting across the image, uint8 ids 1..k on 0
    (the format `create_masks` writes, /root/reference/hive/io.py:214-218).  Later objects overwrite earlier ones."""
    rng = np.random.default_rng(seed)
    v, u = np.mgrid[0:height, 0:width]
    centre = rng.uniform([0.2 * height, 0.2 * width], [0.8 * height, 0.8 * width], size=(num_objects, 2))
    drift = rng.uniform(-0.004, 0.004, size=(num_objects, 2)) * [height, width]
    axes = rng.uniform([0.08 * height, 0.04 * width], [0.25 * height, 0.12 * width], size=(num_objects, 2))
    masks = np.zeros((num_frames, height, width), np.uint8)
    for t in range(num_frames):
        for k in range(num_objects):
            cv, cu = centre[k] + drift[k] * t
            inside = ((v - cv) / axes[k, 0]) ** 2 + ((u - cu) / axes[k, 1]) ** 2 <= 1.0
            masks[t][inside] = k + 1
    return masks


def room_bounds(volume_size=5.12):
    """vol_bnds of the TSDF volume enclosing the synthetic room: [0, volume_size]^3 (512^3 at 1 cm voxels)."""
    return np.array([[0.0, volume_size]] * 3, dtype=np.float64)


def trajectory_rows_world_to_cam(poses_c2w):
    """HIVE's on-disk convention: N x 7 rows (xyzw quaternion + t), world-to-camera
    (/root/reference/README.md:274-282); hive/fusion.py:111 inverts them again."""
    from scipy.spatial.transform import Rotation
    w2c = np.linalg.inv(poses_c2w)
    q = Rotation.from_matrix(w2c[:, :3, :3]).as_quat()
    return np.hstack([q, w2c[:, :3, 3]]).astype(np.float32)
