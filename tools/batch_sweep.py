#!/usr/bin/env python3
"""hive_dpt_forward at small batches: frames/s and ms/frame of the drop-in network object for B in {1, 2, 4, 8, 16, 32, 64, 107}, float16 (the
reference's type; its literal loop is batch 1, /root/reference/hive/dataset_adaptors.py:1406-1419) and bfloat16, 480 x 640 frames resident on
the device, HIP events around `reps` forwards after a warm-up.  Writes profiles/r04_batch_sweep.json (or argv[1])."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hive_amd import depth as depth_mod, synthetic  # noqa: E402

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04_batch_sweep.json")
batches = [int(b) for b in os.environ.get("SWEEP_BATCHES", "1,2,4,8,16,32,64,107").split(",")]
seq = synthetic.make_sequence(num_frames=max(batches), yaw_step_deg=2.4)
frames = torch.from_numpy(seq["color"]).cuda()
result = {"image": [480, 640], "network": "DPT-Hybrid (vitb_rn50_384), seeded weights, hive_dpt_forward (one C-ABI call per batch)", "lib": os.path.basename(os.environ.get("HIVE_AMD_LIB", "libhive_mi355x.so")), "rows": []}
for name, dtype in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
    model = depth_mod.build_model(None, device=torch.device("cuda", 0), dtype=dtype, engine="hip", init_seed=1234)
    with torch.no_grad():
        for b in batches:
            fr = frames[:b].contiguous()
            for _ in range(2):
                model.forward_frames(fr, max_depth=10.0)
            reps = max(3, min(50, 400 // b))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                model.forward_frames(fr, max_depth=10.0)
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1) / reps
            row = {"dtype": name, "batch": b, "ms_per_forward": ms, "ms_per_frame": ms / b, "frames_per_s": b / ms * 1e3}
            result["rows"].append(row)
            print(json.dumps(row), flush=True)
    del model
    torch.cuda.empty_cache()
for name in ("fp16", "bf16"):
    rows = {r["batch"]: r for r in result["rows"] if r["dtype"] == name}
    top = rows[max(rows)]["frames_per_s"]
    for r in rows.values():
        r["share_of_largest_batch_rate"] = r["frames_per_s"] / top
json.dump(result, open(out_path, "w"), indent=1)
