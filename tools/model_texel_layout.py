#!/usr/bin/env python3
"""CPU model of the fused sweep's texel gathers (csrc/tsdf.hip integrate_multi_kernel): distinct 64-byte lines one gather INSTRUCTION touches -- its 64 lanes are
four neighbouring voxel rows x 16 consecutive z voxels -- under the row-major texel layout the kernel uses (8 texels of 8 bytes per line) and under the
4 x 2-pixel tiled layout DESIGN section 9 proposed, over the benchmark's own trajectory (150 poses, 2.4 degrees apart, 640 x 480 into 512^3 at
1 cm) and BASELINE config 4's geometry (1920 x 1080 into 1024^3 at 5 mm).  Every (row quad, 16-voxel z group) whose voxels project into the image in front of the
far wall is an instruction; the lane axis (x or y) is the kernel's own per-sweep choice.  Output: lines per instruction for both layouts per yaw, their means, and the
mean under a per-sweep choice of the better layout (the best the proposal could do).  No GPU.   python tools/model_texel_layout.py > profiles/r05_texel_layout_model.json"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_amd import synthetic  # noqa: E402


def lines_per_instruction(pose, K, H, W, dim, voxel, rng, samples=6000, max_depth=4.6):
    R, T = pose[:3, :3], pose[:3, 3]
    lanes_along_x = abs(pose[0, 1]) <= abs(pose[1, 1])  # the kernel's rule (launch_integrate_multi): lanes along the axis the camera's "down" has less of
    out = {"row_major": [], "tiled_4x2": []}
    n = 0
    while n < samples:
        a = rng.integers(0, dim - 4, 4096)
        b = rng.integers(0, dim, 4096)
        z0 = rng.integers(0, dim // 16, 4096) * 16
        for aa, bb, zz in zip(a, b, z0):
            quad = np.arange(4)
            xs = (aa + quad) if lanes_along_x else np.full(4, bb)
            ys = np.full(4, bb) if lanes_along_x else (aa + quad)
            zs = zz + np.arange(16)
            pts = np.stack(np.broadcast_arrays(xs[:, None] * voxel, ys[:, None] * voxel, zs[None, :] * voxel), axis=-1).reshape(-1, 3)
            cam = (pts - T) @ R
            if cam[:, 2].min() <= 0.05 or cam[:, 2].max() > max_depth:
                continue
            px = np.rint(K[0, 0] * cam[:, 0] / cam[:, 2] + K[0, 2]).astype(np.int64)
            py = np.rint(K[1, 1] * cam[:, 1] / cam[:, 2] + K[1, 2]).astype(np.int64)
            ok = (px >= 0) & (px < W) & (py >= 0) & (py < H)
            if ok.sum() < 48:  # (mostly outside the image: such segments are clipped off the work list)
                continue
            px, py = px[ok], py[ok]
            out["row_major"].append(len(np.unique((py * W + px) // 8)))
            out["tiled_4x2"].append(len(np.unique((py // 2) * (W // 4) + px // 4)))
            n += 1
            if n >= samples:
                break
    return {k: float(np.mean(v)) for k, v in out.items()}, bool(lanes_along_x)


def run(name, H, W, dim, voxel, frames):
    rng = np.random.default_rng(0)
    K = synthetic.scaled_intrinsics(H, W).astype(np.float64)
    poses = synthetic.circular_trajectory(150, 5.12, yaw_step_deg=2.4)
    rows = []
    for f in frames:
        m, along_x = lines_per_instruction(poses[f], K, H, W, dim, voxel, rng)
        rows.append({"frame": int(f), "yaw_deg": round(2.4 * f, 1), "lanes_along_x": along_x, **{k: round(v, 2) for k, v in m.items()}})
    rm = np.array([r["row_major"] for r in rows])
    ti = np.array([r["tiled_4x2"] for r in rows])
    return {"geometry": name, "per_yaw": rows, "mean_row_major": float(rm.mean()), "mean_tiled_4x2": float(ti.mean()),
            "mean_best_per_sweep": float(np.minimum(rm, ti).mean()), "sweeps_where_tiled_wins": float((ti < rm).mean()),
            "gain_of_a_per_sweep_choice": float(1.0 - np.minimum(rm, ti).mean() / rm.mean())}


if __name__ == "__main__":
    frames = list(range(0, 150, 6))
    res = [run("bench: 640x480 into 512^3 (1 cm)", 480, 640, 512, 0.01, frames), run("config 4: 1920x1080 into 1024^3 (5 mm)", 1080, 1920, 1024, 0.005, frames)]
    print(json.dumps({"note": "distinct 64-byte lines per 64-lane gather instruction (4 neighbouring rows x 16 consecutive z voxels), CPU model of csrc/tsdf.hip's gather role",
                      "results": res}, indent=1))
