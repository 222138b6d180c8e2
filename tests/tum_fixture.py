"""Writes a synthetic TUM-layout sequence (rgb.txt, depth.txt, groundtruth.txt, rgb/, depth/) from the seeded
room generator -- the stand-in for TUM fr3/walking_xyz, which is not in the container."""
import os

import numpy as np
from PIL import Image
from scipy.spatial.transform import Rotation


def write_tum_sequence(root, num_frames=6, yaw_step_deg=20.0, seed=7):
    from hive_amd import synthetic
    seq = synthetic.make_sequence(num_frames=num_frames, height=480, width=640, yaw_step_deg=yaw_step_deg, seed=seed)
    os.makedirs(os.path.join(root, "rgb"), exist_ok=True)
    os.makedirs(os.path.join(root, "depth"), exist_ok=True)
    t0 = 1341846313.0
    rgb_lines, depth_lines, gt_lines = ["# color images"], ["# depth maps"], ["# ground truth trajectory", "# timestamp tx ty tz qx qy qz qw"]
    for i in range(num_frames):
        t_rgb, t_depth = t0 + i / 30.0, t0 + i / 30.0 + 0.004  # unsynchronised sensors
        Image.fromarray(seq["color"][i]).save(os.path.join(root, "rgb", f"{t_rgb:.6f}.png"))
        depth_5000 = np.round(seq["depth"][i].astype(np.float64) * 5000.0).astype(np.uint16)
        Image.fromarray(depth_5000).save(os.path.join(root, "depth", f"{t_depth:.6f}.png"))
        rgb_lines.append(f"{t_rgb:.6f} rgb/{t_rgb:.6f}.png")
        depth_lines.append(f"{t_depth:.6f} depth/{t_depth:.6f}.png")
    # ground truth at 100 Hz, cam-to-world, interpolated between the frame poses by nearest frame
    for k in range(num_frames * 4):
        t = t0 + k / 120.0
        i = min(int(round((t - t0) * 30.0)), num_frames - 1)
        pose = seq["poses"][i]
        q = Rotation.from_matrix(pose[:3, :3]).as_quat()
        tr = pose[:3, 3]
        gt_lines.append(f"{t:.4f} {tr[0]:.6f} {tr[1]:.6f} {tr[2]:.6f} {q[0]:.6f} {q[1]:.6f} {q[2]:.6f} {q[3]:.6f}")
    for name, lines in (("rgb.txt", rgb_lines), ("depth.txt", depth_lines), ("groundtruth.txt", gt_lines)):
        with open(os.path.join(root, name), "w") as f:
            f.write("\n".join(lines) + "\n")
    return seq
