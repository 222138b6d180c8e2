#!/usr/bin/env python3
"""Steady-state kernel table of the bench step (torch.profiler around 5 warm steps)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_amd import _lib, depth as depth_mod, fusion, synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.backends.cudnn.benchmark = True
seq = synthetic.make_sequence(num_frames=B, yaw_step_deg=2.4)
ctx = _lib.default_context(0)
model = depth_mod.build_model(None, dtype=torch.bfloat16)
vol = fusion.TSDFVolume(synthetic.room_bounds(), 0.01, ctx=ctx)
stream = depth_mod.DepthFusionStream(model, vol, seq["K"])
frames = torch.from_numpy(seq["color"]).cuda()
for _ in range(4):
    stream.step(frames, seq["poses"])
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(5):
        stream.step(frames, seq["poses"])
    torch.cuda.synchronize()
ev = [e for e in prof.key_averages() if e.device_time_total > 0 and getattr(e, "device_type", None) is not None]
rows = sorted(((e.device_time_total / 5.0, e.count / 5.0, e.key) for e in prof.key_averages() if e.self_device_time_total > 0 and e.is_user_annotation is False), reverse=True)
seen = 0.0
print(f"B={B}")
for t, c, k in rows[:60]:
    print(f"{t:9.1f} us/step {c:7.1f} calls/step  {k[:120]}")
