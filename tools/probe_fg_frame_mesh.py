#!/usr/bin/env python3
"""BASELINE configs[4]'s foreground side: 300 synthetic 640 x 480 frames with 1-3 moving ellipse masks -- one object mesh per (frame, object) -- through
hive_fg_frame_mesh (device-resident frames, one call and one read-back per object) against the separate entry points (unproject + grid_mesh x 2 + texture_window, host
arrays in and out) and, on a sample, a numpy restatement of the reference's loop body with scipy's Delaunay (pipeline.py:383-461 without decimation / components).
Usage (GPU box): python tools/probe_fg_frame_mesh.py > profiles/r05_fg_frame_mesh.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from hive_amd import _lib, foreground, geometric, synthetic  # noqa: E402
from hive_amd.options import MeshFilteringOptions  # noqa: E402

T, H, W = 300, 480, 640
seq = synthetic.make_sequence(num_frames=30, height=H, width=W, yaw_step_deg=2.4)  # (30 distinct frames, wrapped: the masks move over all 300)
masks = synthetic.ellipse_masks(T, H, W, num_objects=3, seed=1234)
K = seq["K"]
opts = MeshFilteringOptions()
ctx = _lib.default_context(0)
depth_d = torch.from_numpy(seq["depth"]).cuda()
rgb_d = torch.from_numpy(seq["color"]).cuda()
masks_d = torch.from_numpy(masks).cuda()
buffers = foreground.FrameMeshBuffers(H, W)
poses = [np.linalg.inv(seq["poses"][f % 30]) for f in range(T)]


def one_call():
    n_obj = n_v = n_f = 0
    for f in range(T):
        R, t = poses[f][:3, :3], poses[f][:3, 3:4]
        for obj in (1, 2, 3):
            m = masks_d[f] == obj
            r = foreground.frame_mesh(depth_d[f % 30], m, rgb_d[f % 30], K, R, t, opts, ctx=ctx, buffers=buffers)
            if r["vertices"].shape[0]:
                n_obj, n_v, n_f = n_obj + 1, n_v + r["vertices"].shape[0], n_f + r["faces"].shape[0]
    torch.cuda.synchronize()
    return n_obj, n_v, n_f


def separate(frames):
    n = 0
    for f in frames:
        R, t = poses[f][:3, :3], poses[f][:3, 3:4]
        for obj in (1, 2, 3):
            m = masks[f] == obj
            v = geometric.point_cloud_from_depth(seq["depth"][f % 30], m, K, R, t)
            if len(v) == 0:
                continue
            foreground.grid_faces(seq["depth"][f % 30], m, opts, ctx=ctx)
            foreground.get_mesh_texture_and_uv(v, seq["color"][f % 30], K, R, t, ctx=ctx)
            n += 1
    return n


def numpy_reference(frames):
    from scipy.spatial import Delaunay
    n = 0
    Kinv = np.linalg.inv(K.astype(np.float64))
    for f in frames:
        R, t = poses[f][:3, :3], poses[f][:3, 3:4]
        depth, rgb = seq["depth"][f % 30], seq["color"][f % 30]
        for obj in (1, 2, 3):
            m = masks[f] == obj
            valid = m & (depth > 0)
            vv, uu = valid.nonzero()
            if len(vv) < 9:
                continue
            pts = np.vstack((uu, vv, np.ones_like(uu))).astype(np.float64)
            verts = (R.T @ (depth[valid] * (Kinv @ pts) - t)).T
            p2 = np.vstack((uu, vv)).T
            faces = np.asarray(Delaunay(p2).simplices)[:, ::-1]
            pd = np.linalg.norm(p2[faces[:, [0, 2, 0]]] - p2[faces[:, [1, 1, 2]]], axis=-1)
            dd = np.abs(depth[valid][faces[:, [0, 2, 0]]] - depth[valid][faces[:, [1, 1, 2]]])
            faces = faces[np.all((pd <= opts.max_pixel_distance) & (dd <= opts.max_depth_distance), axis=1)]
            cam = R @ verts.T + t
            uv = np.round((K.astype(np.float64) @ cam)[:2] / cam[2]).T.astype(np.int32)
            lo, hi = uv.min(0), uv.max(0) + 1
            rgb[lo[1]:hi[1], lo[0]:hi[0]].copy()
            n += 1
    return n


one_call()
t0 = time.perf_counter()
n_obj, n_v, n_f = one_call()
t_one = time.perf_counter() - t0
sample = list(range(0, T, 10))
separate(sample[:3])
t0 = time.perf_counter()
n_sep = separate(sample)
t_sep = (time.perf_counter() - t0) / max(n_sep, 1)
t0 = time.perf_counter()
n_np = numpy_reference(sample[:6])
t_np = (time.perf_counter() - t0) / max(n_np, 1)
print(json.dumps({"workload": f"{T} frames {W}x{H}, three moving ellipse masks: {n_obj} object meshes, {n_v} vertices, {n_f} faces in all",
                  "hive_fg_frame_mesh": {"ms_per_object": t_one / n_obj * 1e3, "ms_per_frame": t_one / T * 1e3, "seconds_300_frames": t_one,
                                         "note": "device-resident depth / rgb / masks, one C-ABI call + one pinned read-back per object (the mask comparison is a torch op)"},
                  "separate_entry_points": {"ms_per_object": t_sep * 1e3, "objects_timed": n_sep,
                                            "note": "hive_unproject + hive_grid_mesh x 2 + hive_texture_window, numpy arrays in and out (round 4's path)"},
                  "numpy_scipy_restatement": {"ms_per_object": t_np * 1e3, "objects_timed": n_np, "cores": os.cpu_count(),
                                              "note": "pipeline.py:383-461's arithmetic with scipy.spatial.Delaunay, single-threaded as the reference's loop body"}}, indent=1))
