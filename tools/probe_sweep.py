#!/usr/bin/env python3
"""Time the fused multi-frame sweep (hive_tsdf_integrate_batch on device frames) on the room scene: 32 consecutive frames 2.4 degrees apart into
512^3, HIP events around each batch call; prints microseconds per frame (pack + work list + sweep) -- for A/B runs of kernel variants
(HIVE_AMD_LIB=...)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_amd import _lib, fusion, synthetic  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 32
seq = synthetic.make_sequence(num_frames=frames, yaw_step_deg=2.4)
ctx = _lib.default_context(0)
vol = fusion.TSDFVolume(synthetic.room_bounds(), 0.01, ctx=ctx)
color, depth = torch.from_numpy(seq["color"]).cuda(), torch.from_numpy(seq["depth"]).cuda()
best = []
for rep in range(6):
    vol.reset()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    vol.integrate_batch(color, depth, seq["K"], seq["poses"])
    b.record()
    torch.cuda.synchronize()
    best.append(a.elapsed_time(b) * 1e3 / frames)
print(f"groups {vol.last_batch_groups()[:4]}..., us per frame (6 reps): " + " ".join(f"{v:.1f}" for v in best) + f"   min {min(best):.1f}  lib {os.path.basename(_lib.LIB_PATH)}", flush=True)
