#!/usr/bin/env python3
"""BASELINE config 4, informational: 1920 x 1080 frames -> the reference's resize (640 x 480 target, keep aspect ratio, lower bound,
multiple of 32: 864 x 480 for 16:9) -> DPT-Large depth (hive_dpt_forward, backbone 1) -> nearest back to 1080p (estimate_depth_dpt's
rule) -> uint16-mm hand-off -> integrate into a 1024^3 volume (5 mm voxels).  Prints frames/s and the per-leg times."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_amd import _lib, fusion, synthetic, depth as depth_mod
from hive_amd.dpt.models import DPTDepthModel
from hive_amd.dpt.transforms import Resize
B, T = 8, 16
net_w, net_h = Resize(640, 480, resize_target=None, keep_aspect_ratio=True, ensure_multiple_of=32, resize_method="lower_bound").get_size(1920, 1080)
seq = synthetic.make_sequence(num_frames=T, height=1080, width=1920, yaw_step_deg=360.0 / T)
model = DPTDepthModel(path=None, scale=depth_mod.DPT_SCALE, shift=depth_mod.DPT_SHIFT, invert=True, backbone="vitl16_384", engine="hip").eval()
model = model.to(memory_format=torch.channels_last).to(torch.bfloat16).cuda()
ctx = _lib.default_context(0)
vol = fusion.TSDFVolume(synthetic.room_bounds(), 0.005, ctx=ctx)
frames = torch.from_numpy(seq["color"]).cuda()
def dpt(fr):
    small = torch.nn.functional.interpolate(fr.permute(0, 3, 1, 2).float(), size=(net_h, net_w), mode="area").round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    d, _, _ = model.forward_frames(small, max_depth=None)
    full = torch.nn.functional.interpolate(d[:, None], size=(1080, 1920), mode="nearest")[:, 0]
    mm = (full * 1000.0).to(torch.int32).clamp(0, 65535)
    m = mm.float() * (1.0 / 1000.0)
    return torch.where(m > 10.0, torch.zeros_like(m), m).contiguous()
def run():
    t_d = t_i = 0.0
    for i in range(0, T, B):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        dm = dpt(frames[i:i + B])
        e[1].record()
        vol.integrate_batch(frames[i:i + B], dm, seq["K"], seq["poses"][i:i + B])
        e[2].record()
        torch.cuda.synchronize()
        t_d += e[0].elapsed_time(e[1]); t_i += e[1].elapsed_time(e[2])
    return t_d, t_i
run()
vol.reset()
t0 = time.time(); t_d, t_i = run(); wall = time.time() - t0
print(f"network input {net_w} x {net_h}; {T} frames in {wall*1e3:.1f} ms = {T/wall:.1f} frames/s; DPT-Large {t_d/T:.2f} ms/frame (batch {B}), integrate 1080p -> {tuple(int(x) for x in vol.vol_dim)} {t_i/T:.2f} ms/frame")
