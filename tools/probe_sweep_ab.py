#!/usr/bin/env python3
"""A/B of the fused TSDF sweep's switches in ONE process, on both scenes of bench.py (room: analytic depth; dpt: DPT-Hybrid depth of the
seeded weights), 32 consecutive frames 2.4 degrees apart into 512^3.  Switches (HIVE_TSDF_<name>, all default 1): ROW_FAR (per-row far cut from the
depth tiles), FRAME_SKIP (work-item frame masks), FAST_COLOUR (division-free colour update), SORT (work list sorted by image band, eighths to the XCDs), QUAD (segments of four neighbouring rows interleaved); LANES=x|y forces the lane axis of the work-list kernel.
PROBE_CONFIGS = ';'-separated configurations, each a ','-separated list of NAME=0/1 (the first is the reference the others' volumes are compared
with); default: everything off, then everything on.
Per configuration: us per frame of the whole leg (prep + work list + sort + sweep, HIP events), us per sweep launch (the library's own events), work-list
voxels of the last sweep, and a check that the volume is bit-identical to the first configuration's.  Usage: probe_sweep_ab.py [frames] [scene ...]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_amd import _lib, depth as depth_mod, fusion, synthetic  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 32
scenes = sys.argv[2:] or ["room", "dpt"]
# PROBE_SIZE=HxW and PROBE_VOL=n (voxels per side of the 5.12 m room volume) select another shape (BASELINE config 4: 1080x1920, 1024; room scene only)
H, W = (int(v) for v in os.environ.get("PROBE_SIZE", "480x640").split("x"))
side = int(os.environ.get("PROBE_VOL", "512"))
seq = synthetic.make_sequence(num_frames=frames, height=H, width=W, yaw_step_deg=2.4)
ctx = _lib.default_context(0)
n_vox = side ** 3
storage = tuple(torch.empty(n_vox, dtype=torch.float32, device="cuda") for _ in range(3))
vol = fusion.TSDFVolume(synthetic.room_bounds(), 5.12 / side, ctx=ctx, storage=storage)
assert tuple(int(d) for d in vol.vol_dim) == (side, side, side), vol.vol_dim
color = torch.from_numpy(seq["color"]).cuda()
depths = {}
if "room" in scenes:
    depths["room"] = torch.from_numpy(seq["depth"]).cuda()
if "dpt" in scenes:
    model = depth_mod.build_model(None, device=torch.device("cuda", 0), dtype=torch.bfloat16, engine="hip", init_seed=1234)
    stream = depth_mod.DepthFusionStream(model, vol, seq["K"])
    depths["dpt"] = stream.depth(color)[0].clone()
    del model, stream
out = {}
NAMES = ("ROW_FAR", "FRAME_SKIP", "FAST_COLOUR", "SORT", "QUAD")  # on / off switches (default on); any other HIVE_TSDF_<NAME>=value may be given too (LANES=x|y)
spec = os.environ.get("PROBE_CONFIGS") or (",".join(n + "=0" for n in NAMES) + ";" + ",".join(n + "=1" for n in NAMES))
configs = [dict(kv.split("=") for kv in c.split(",") if kv) for c in spec.split(";")]
for scene, depth in depths.items():
    # N_upd per frame (counting single-frame kernel) and N_union per sweep of four
    vol.reset()
    n_upd = [vol.integrate(color[i], depth[i], seq["K"], seq["poses"][i], return_n_updated=True) for i in range(min(frames, 8))]
    ref = None
    for cfg in configs:
        for key in [k for k in os.environ if k.startswith("HIVE_TSDF_") and k != "HIVE_TSDF_TIMING_SAME_TEXELS"]:
            del os.environ[key]
        for name, value in cfg.items():
            os.environ["HIVE_TSDF_" + name] = str(value)
        leg, k_us = [], []
        for rep in range(5):
            vol.reset()
            torch.cuda.synchronize()
            ctx.set_timing(True)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            vol.integrate_batch(color, depth, seq["K"], seq["poses"])
            b.record()
            torch.cuda.synchronize()
            n, ms = ctx.kernel_time_total()
            ctx.set_timing(False)
            leg.append(a.elapsed_time(b) * 1e3 / frames)
            k_us.append(ms * 1e3 / max(n, 1))
        wl = vol.last_sweep_voxels()
        state = torch.stack([s.clone() for s in storage])
        if ref is None:
            ref = state
            same = True
        else:
            same = bool(torch.equal(ref, state))
        # N_union of the last sweep: weights that move across it
        vol.reset()
        groups = vol.last_batch_groups()
        before = storage[1].clone()
        nf = groups[-1]
        vol.integrate_batch(color[frames - nf:], depth[frames - nf:], seq["K"], seq["poses"][frames - nf:])
        n_union = int((storage[1] != before).sum().item())
        wl_last = vol.last_sweep_voxels()
        rec = {"config": dict(cfg), "leg_us_per_frame_min": min(leg), "launch_us_min": min(k_us), "launch_us_all": [round(v, 1) for v in k_us],
               "worklist_voxels_last_sweep": wl_last, "n_union_last_sweep": n_union, "frames_last_sweep": nf, "n_upd_mean": float(np.mean(n_upd)),
               "bit_identical_to_first": same, "groups": groups[:3]}
        out.setdefault(scene, []).append(rec)
        print(scene, json.dumps(rec), flush=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "sweep_ab.json"), "w"), indent=1)
