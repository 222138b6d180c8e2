#!/usr/bin/env python3
"""The small kernels of the path at the benchmark's sizes, device-resident inputs, HIP events around each call (median of --reps):
marching cubes at 512^3 on the room volume, unproject / project_bbox / grid_mesh / depth_apply_mask / view_frustum_batch at
640 x 480.  Prints one JSON object with SURVEY.md 8(d)'s algorithmic bytes and the achieved GB/s per entry point; run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel table (tools/profile_r03.sh -> profiles/r03_small_kernels.*).

Reference call sites: hive/fusion.py:127 (get_mesh), hive/geometric.py:107-180 (point_cloud_from_depth, world2image),
hive/io.py:1117-1189 (select_key_frames), hive/pipeline.py:651-694 (foreground triangulation + filter), hive/fusion.py:53-61,
118-121 (bounds pass, mask dilation).  SURVEY.md section 6 quotes numpy at 411 ms (point_cloud_from_depth) / 119 ms (world2image)
per VGA frame for the reference."""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_amd import _lib, foreground, fusion, synthetic  # noqa: E402
from hive_amd._lib import MEM_DEVICE, ptr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=9)
ap.add_argument("--frames", type=int, default=32)
ap.add_argument("--voxel", type=float, default=0.01)
args = ap.parse_args()

H, W = 480, 640
seq = synthetic.make_sequence(num_frames=args.frames, height=H, width=W, yaw_step_deg=2.4)
ctx = _lib.default_context(0)
color = torch.from_numpy(seq["color"]).cuda()
depth = torch.from_numpy(seq["depth"]).cuda()
K = seq["K"]


def timed(fn, reps=args.reps):
    fn()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    return float(np.median(ms)), out


res = {}

# marching cubes, 512^3, the room after `frames` consecutive frames (hive_tsdf_extract_mesh; device side incl. its count read-backs)
vol = fusion.TSDFVolume(synthetic.room_bounds(), args.voxel, ctx=ctx)
vol.integrate_batch(color, depth, K, seq["poses"])
N = vol.num_voxels


ms, (nv, nf) = timed(vol._extract)  # every call re-extracts (frees the previous mesh, allocates the new one)
b = 4.0 * N + 36.0 * nv + 12.0 * nf + 4.0 * nv
res["marching_cubes_512"] = {"ms": ms, "vertices": nv, "faces": nf, "algorithmic_bytes": b, "gbs": b / ms / 1e6,
                             "bytes_formula": "4 N (tsdf) + 36 V (verts, norms f32 + colours) + 12 F (faces i32) + 4 V (colour gathers)"}

# unproject (point_cloud_from_depth): 5 H W in (depth f32 + mask u8), 24 N out (float64 xyz, as the reference returns)
d0 = depth[0].contiguous()
mask = (d0 > 0).to(torch.uint8)
Kinv = np.ascontiguousarray(np.linalg.inv(K), dtype=np.float64)
pose_w2c = np.linalg.inv(seq["poses"][0])
R, t = np.ascontiguousarray(pose_w2c[:3, :3]), np.ascontiguousarray(pose_w2c[:3, 3])
pts = torch.empty((H * W, 3), dtype=torch.float64, device="cuda")
n_pts = ctypes.c_int64(0)


def unproject():
    ctx.check(ctx.lib.hive_unproject(ctx.handle, d0.data_ptr(), mask.data_ptr(), None, H, W, ptr(Kinv), ptr(R), ptr(t), MEM_DEVICE, pts.data_ptr(), None,
                                     H * W, ctypes.byref(n_pts)))
    return n_pts.value


ms, n = timed(unproject)
b = 5.0 * H * W + 24.0 * n
res["unproject_vga"] = {"ms": ms, "points": n, "algorithmic_bytes": b, "gbs": b / ms / 1e6, "reference_numpy_ms": 411, "bytes_formula": "5 H W + 24 N"}

# project_bbox (select_key_frames' inner step): 24 N in, 20 bytes out
pose2 = np.linalg.inv(seq["poses"][4])
R2, t2 = np.ascontiguousarray(pose2[:3, :3]), np.ascontiguousarray(pose2[:3, 3])
K64 = np.ascontiguousarray(K, dtype=np.float64)
box = np.zeros(5, np.int32)


def bbox():
    ctx.check(ctx.lib.hive_project_bbox(ctx.handle, pts.data_ptr(), n, ptr(K64), ptr(R2), ptr(t2), W, H, MEM_DEVICE, ptr(box)))
    return box.copy()


ms, _ = timed(bbox)
res["project_bbox_vga"] = {"ms": ms, "points": n, "algorithmic_bytes": 24.0 * n, "gbs": 24.0 * n / ms / 1e6, "reference_numpy_ms": 119, "bytes_formula": "24 N",
                           "note": "incl. the 20-byte read-back (one stream sync per call)"}

# grid_mesh (foreground triangulation + face filter): 5 H W in, 12 F out; two calls (size, then fill) as the wrapper does
obj = synthetic.ellipse_masks(1, H, W, num_objects=3)[0]
obj_d = torch.from_numpy((obj > 0).astype(np.uint8)).cuda()
ms, (faces, nvert) = timed(lambda: foreground.grid_faces(d0, obj_d, ctx=ctx, return_vertex_count=True))
b = 5.0 * H * W + 12.0 * faces.shape[0]
res["grid_mesh_vga"] = {"ms": ms, "faces": int(faces.shape[0]), "vertices": int(nvert), "algorithmic_bytes": b, "gbs": b / ms / 1e6, "bytes_formula": "5 H W + 12 F",
                        "note": "sizing call + fill call, 5 launches each, 2 read-backs: latency-bound"}

# depth_apply_mask over the frame set: (4 + 1) H W in, 4 H W out per frame
masks = torch.from_numpy(synthetic.ellipse_masks(args.frames, H, W)).cuda()
out = torch.empty_like(depth)
ms, _ = timed(lambda: ctx.check(ctx.lib.hive_depth_apply_mask(ctx.handle, depth.data_ptr(), masks.data_ptr(), args.frames, H, W, 9, 0, 0, out.data_ptr())))
b = 9.0 * H * W * args.frames
res["depth_apply_mask_set"] = {"ms": ms, "frames": args.frames, "iterations": 9, "algorithmic_bytes": b, "gbs": b / ms / 1e6, "bytes_formula": "9 H W per frame"}

# view_frustum_batch (bounds pass): 4 H W per frame in
ms, _ = timed(lambda: fusion.view_frusta(depth, K, seq["poses"], ctx))
b = 4.0 * H * W * args.frames
res["view_frustum_batch_set"] = {"ms": ms, "frames": args.frames, "algorithmic_bytes": b, "gbs": b / ms / 1e6, "bytes_formula": "4 H W per frame", "note": "incl. the read-back"}

print(json.dumps(res, indent=1))
