#!/usr/bin/env python3
"""Calibration probe for the integrate kernel: a frame in which EVERY voxel is updated (camera 1 km behind
the volume, telephoto intrinsics so the whole volume projects inside the image, constant depth far beyond it:
sdf clamps to 1 everywhere).  N_upd = N exactly, so the algorithmic bytes are 24 N + 8 H W with every cache
line full -- the kernel's dense-sweep bandwidth and the FETCH_SIZE / WRITE_SIZE calibration point."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_amd import _lib, fusion  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dim", type=int, default=512)
ap.add_argument("--frames", type=int, default=10)
ap.add_argument("--fraction", type=float, default=1.0, help="updated fraction of every z row (depth plane cuts the volume)")
args = ap.parse_args()

D = args.dim
vs = 0.01
ext = D * vs
bounds = np.array([[0, ext], [0, ext], [0, ext]], dtype=np.float64)
ctx = _lib.default_context(0)
vol = fusion.TSDFVolume(bounds, vs, ctx=ctx)
H, W = 480, 640
f = 0.45 * H / (ext / 2) * 1000.0
K = np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], dtype=np.float32)
pose = np.eye(4)
pose[:3, 3] = [ext / 2, ext / 2, -1000.0]
# depth plane: beyond the volume (fraction 1) or cutting it at fraction * ext
d = 1000.0 + (ext * args.fraction if args.fraction < 1.0 else 1000.0)
depth = torch.full((H, W), d, dtype=torch.float32, device="cuda")
color = torch.randint(0, 255, (H, W, 3), dtype=torch.uint8, device="cuda")
n = vol.integrate(color, depth, K, pose, return_n_updated=True)
print("dims", vol.vol_dim, "N_upd / N =", n / vol.num_voxels)
ctx.set_timing(True)
for _ in range(args.frames):
    vol.integrate(color, depth, K, pose)
torch.cuda.synchronize()
cnt, ms = ctx.kernel_time_total()
alg = 24.0 * n + 8.0 * H * W
print(f"{cnt} launches, kernel avg {ms / cnt * 1e3:.1f} us, algorithmic {alg / 1e6:.1f} MB -> {alg / (ms / cnt * 1e-3) / 1e9:.1f} GB/s")
