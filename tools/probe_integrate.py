#!/usr/bin/env python3
"""Quick probe: integrate-kernel time on the room scene (HIP events), N_upd, GB/s: 512^3 / VGA by default."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_amd import _lib, fusion, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=30)
ap.add_argument("--voxel", type=float, default=0.01)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--height", type=int, default=480)
ap.add_argument("--width", type=int, default=640, help="--height 1080 --width 1920 --voxel 0.005: BASELINE config 4 (1024^3)")
ap.add_argument("--no-mesh", action="store_true")
ap.add_argument("--yaw-step", type=float, default=None, help="degrees between frames (default: 360 / frames; 2.4 = the bench trajectory's consecutive frames)")
args = ap.parse_args()

seq = synthetic.make_sequence(num_frames=args.frames, height=args.height, width=args.width, yaw_step_deg=args.yaw_step if args.yaw_step is not None else 360.0 / args.frames)
ctx = _lib.default_context(0)
vol = fusion.TSDFVolume(synthetic.room_bounds(), args.voxel, ctx=ctx)
color = torch.from_numpy(seq["color"]).cuda()
depth = torch.from_numpy(seq["depth"]).cuda()
H, W = depth.shape[1:]
n_upd = [vol.integrate(color[i], depth[i], seq["K"], seq["poses"][i], return_n_updated=True) for i in range(args.frames)]
print("dims", vol.vol_dim, "N_upd/N mean", np.mean(n_upd) / vol.num_voxels, "min/max", min(n_upd) / vol.num_voxels, max(n_upd) / vol.num_voxels)
for rep in range(args.reps):
    ctx.set_timing(True)
    torch.cuda.synchronize()
    t0 = time.time()
    vol.integrate_batch(color, depth, seq["K"], seq["poses"])
    torch.cuda.synchronize()
    wall = time.time() - t0
    n, ms = ctx.kernel_time_total()
    ctx.set_timing(False)
    alg = (24.0 * np.mean(n_upd) + 8.0 * H * W)
    fpl = args.frames / n  # frames per launch (integrate_batch sweeps up to 4 consecutive device frames at once)
    print(f"rep {rep}: {n} launches ({fpl:.2f} frames each), kernel avg {ms / n * 1e3:.1f} us = {ms / args.frames * 1e3:.1f} us/frame, wall/frame {wall / args.frames * 1e3:.3f} ms, "
          f"algorithmic {alg / 1e6:.1f} MB/frame -> {alg * fpl / (ms / n * 1e-3) / 1e9:.1f} GB/s")
if args.no_mesh:
    sys.exit(0)
t0 = time.time()
verts, faces, norms, colors = vol.get_mesh()
print(f"mesh: {len(verts)} verts, {len(faces)} faces in {time.time() - t0:.3f} s (incl. D2H)")
