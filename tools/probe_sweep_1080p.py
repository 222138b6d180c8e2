#!/usr/bin/env python3
"""BASELINE config 4's TSDF side alone: 8 consecutive 1920 x 1080 frames (2.4 degrees apart, analytic room depth) into a 1024^3 volume (5 mm voxels): HIP-event time of
the integrate launches (hive_ctx_set_timing) and of the whole hive_tsdf_integrate_batch leg.  The tuning switches come from the environment (one process per variant:
HIVE_TSDF_FRAMES_PER_LAUNCH is read once).  Usage (GPU box): [HIVE_TSDF_...=..] python tools/probe_sweep_1080p.py [label]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from hive_amd import _lib, fusion, synthetic  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else "default"
H, W, n = 1080, 1920, 8
seq = synthetic.make_sequence(num_frames=n, height=H, width=W, yaw_step_deg=2.4)
ctx = _lib.default_context(0)
vol = fusion.TSDFVolume(synthetic.room_bounds(), 0.005, ctx=ctx)
color = torch.from_numpy(seq["color"]).cuda()
depth = torch.from_numpy(seq["depth"]).cuda()
vol.integrate_batch(color, depth, seq["K"], seq["poses"])
torch.cuda.synchronize()
vol.reset()
best = None
for rep in range(3):
    vol.reset()
    torch.cuda.synchronize()
    ctx.set_timing(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    vol.integrate_batch(color, depth, seq["K"], seq["poses"])
    e1.record()
    e1.synchronize()
    k_n, k_ms = ctx.kernel_time_total()
    ctx.set_timing(False)
    cur = {"label": label, "groups": vol.last_batch_groups(), "launches": k_n, "sweep_us_per_frame": k_ms * 1e3 / n, "leg_us_per_frame": e0.elapsed_time(e1) * 1e3 / n,
           "worklist_voxels_last_sweep": vol.last_sweep_voxels()}
    if best is None or cur["sweep_us_per_frame"] < best["sweep_us_per_frame"]:
        best = cur
w = vol.device_tensors()[1]
best["weight_sum"] = int(w.double().sum().item())
best["env"] = {k: v for k, v in os.environ.items() if k.startswith("HIVE_TSDF")}
print(json.dumps(best), flush=True)
