import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hive_amd import _lib
torch.cuda.set_device(0)
device = torch.device("cuda", 0)
ctx = _lib.default_context(0)
for batch, steps in ((16, 2), (24, 2), (32, 2), (48, 2)):
    r = bench.config4_leg(device, ctx, steps=steps, batch=batch, overlap=True, prefetch=True, unique_frames=24)
    print(json.dumps({"batch": batch, "frames_per_s": round(r["value"], 1), "ms_per_step": round(r["ms_per_step"], 2), "dpt_ms_per_frame": round(r["dpt_ms_per_frame"], 3), "sweep_us_per_frame": round(r["roofline"]["us_per_frame"], 1)}), flush=True)
    torch.cuda.empty_cache()
