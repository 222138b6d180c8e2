#!/usr/bin/env python3
"""Generate golden vectors from the REAL reference module ``hive.geometric`` (run in the build
container only; /root/reference never travels to the GPU box -- the .npz fixtures do).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python /root/repo/tests/golden/make_golden.py

Only inputs and outputs are stored (data, no reference source).  Harness-side shims: the reference
uses two NumPy-1 names (np.alltrue, np.product) that NumPy 2 removed.
"""
import os
import sys

import numpy as np

np.alltrue = np.all
np.product = np.prod
sys.path.insert(0, "/root/reference")
os.environ.setdefault("MPLBACKEND", "Agg")

import hive.geometric as G  # noqa: E402
from hive.sensor import KinectSensor  # noqa: E402
from scipy.spatial.transform import Rotation  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def random_pose(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    R = Rotation.from_quat(q).as_matrix()
    t = rng.uniform(-1.0, 1.0, size=(3, 1))
    return q, R, t


def main():
    rng = np.random.default_rng(0)
    H, W = 48, 64
    cam = KinectSensor.get_camera_matrix().scale((H, W))
    K64 = cam.matrix
    K32 = K64.astype(np.float32)  # what HiveDataset hands out (io.py:1100)
    out = {"K64": K64, "K32": K32, "H": H, "W": W}

    # ---- point_cloud_from_depth / point_cloud_from_rgbd / image2world -------------------------------
    depth = rng.uniform(0.5, 5.0, size=(H, W)).astype(np.float32)
    depth[rng.random((H, W)) < 0.10] = 0.0
    mask = rng.random((H, W)) < 0.8
    rgb = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    q, R, t = random_pose(rng)
    out.update(depth=depth, mask=mask, rgb=rgb, R=R, t=t)
    for name, K in (("k32", K32), ("k64", K64)):
        out[f"pc_{name}"] = G.point_cloud_from_depth(depth, mask, K, R, t)
        pts, col = G.point_cloud_from_rgbd(rgb, depth, mask, K, R, t)
        out[f"pcrgbd_pts_{name}"] = pts
        out[f"pcrgbd_col_{name}"] = col
    out["pc_identity"] = G.point_cloud_from_depth(depth, mask, K32)
    out["pc_allmask"] = G.point_cloud_from_depth(depth, np.ones_like(mask), K32, R, t)
    uv = np.stack([rng.uniform(0, W, 200), rng.uniform(0, H, 200)], axis=1)
    d = rng.uniform(0.3, 6.0, 200)
    out.update(i2w_uv=uv, i2w_d=d, i2w=G.image2world(uv, d, K32, R, t), i2w_scale=G.image2world(uv, d, K32, R, t, scale_factor=2.0))

    # ---- world2image: int32 (np.round half-even, incl. exact .5 ties) and float --------------------------
    pts = out["pc_k32"]
    q2, R2, t2 = random_pose(rng)
    out.update(R2=R2, t2=t2)
    uv_i, dep = G.world2image(pts, K32, R2, t2)
    uv_f, dep_f = G.world2image(pts, K32, R2, t2, dtype=np.float64)
    uv_s, _ = G.world2image(pts, K32, R2, t2, scale_factor=2.0)
    out.update(w2i_uv_i32=uv_i, w2i_depth=dep, w2i_uv_f64=uv_f, w2i_uv_scaled=uv_s)
    # ties: identity pose, K with integer focal length and half-pixel principal point
    Kt = np.array([[16.0, 0.0, 15.5], [0.0, 16.0, 15.5], [0.0, 0.0, 1.0]])
    gx, gy = np.meshgrid(np.arange(-8, 9), np.arange(-8, 9))
    tie_pts = np.stack([gx.ravel() / 16.0, gy.ravel() / 16.0, np.ones(gx.size)], axis=1)  # u = gx + 15.5 exactly
    tie_uv, _ = G.world2image(tie_pts, Kt)
    out.update(tie_K=Kt, tie_pts=tie_pts, tie_uv=tie_uv)

    # ---- pose helpers / Trajectory / CameraMatrix -----------------------------------------------------------
    N = 6
    quats = rng.normal(size=(N, 4))
    quats /= np.linalg.norm(quats, axis=1, keepdims=True)
    traj_values = np.hstack([quats, rng.uniform(-2, 2, size=(N, 3))])
    traj = G.Trajectory(traj_values.copy())
    M = traj.to_homogenous_transforms()
    T = np.eye(4)
    T[:3, :3] = Rotation.from_euler("xyz", [0.1, -0.2, 0.3]).as_matrix()
    T[:3, 3] = [0.5, -0.25, 1.0]
    out.update(traj_values=traj_values, traj_mats=M, traj_inverse=traj.inverse().values, traj_normalise=traj.normalise().values,
               traj_normalise_position=traj.normalise_position().values, traj_apply=traj.apply(T).values, traj_apply_T=T,
               traj_from_mats=G.Trajectory.from_homogenous_transforms(M).values,
               pose_vec2mat=G.pose_vec2mat(traj_values[1]), pose_mat2vec=G.pose_mat2vec(M[2]),
               traj_scaled=traj.scale_trajectory(2.5).values)
    poses = {0: traj_values[0], 4: traj_values[1], 9: traj_values[2]}
    out["traj_interp"] = G.Trajectory.create_by_interpolating(poses, 10).values
    other = G.Trajectory(np.hstack([quats[::-1], rng.uniform(-2, 2, size=(N, 3))]))
    out.update(traj_other=other.values, traj_ate=traj.calculate_ate(other))
    rpe_r, rpe_t = traj.calculate_rpe(other)
    out.update(traj_rpe_r=rpe_r, traj_rpe_t=rpe_t)
    kin = KinectSensor.get_camera_matrix()
    out.update(cam_matrix=kin.matrix, cam_fov_y=kin.fov_y, cam_scaled=kin.scale((240, 320)).matrix, cam_transposed=kin.transpose().matrix)

    np.savez_compressed(os.path.join(OUT, "geometric_48x64.npz"), **out)

    # ---- one full-resolution checksum case (640 x 480): inputs are re-generated from the seed by the tests ----
    rng = np.random.default_rng(1)
    H, W = 480, 640
    Kfull = KinectSensor.get_camera_matrix().matrix.astype(np.float32)
    depth = rng.uniform(0.5, 5.0, size=(H, W)).astype(np.float32)
    depth[rng.random((H, W)) < 0.10] = 0.0
    mask = rng.random((H, W)) < 0.9
    q, R, t = random_pose(rng)
    pc = G.point_cloud_from_depth(depth, mask, Kfull, R, t)
    uv, dep = G.world2image(pc, Kfull, R, t)
    idx = np.linspace(0, len(pc) - 1, 257).astype(np.int64)
    np.savez_compressed(os.path.join(OUT, "geometric_full_checksums.npz"), R=R, t=t, n=len(pc), idx=idx, pc_rows=pc[idx],
                        pc_sum=pc.sum(axis=0), pc_abs_sum=np.abs(pc).sum(axis=0), uv_rows=uv[idx],
                        uv_sum=uv.astype(np.int64).sum(axis=0), dep_sum=dep.sum())
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
