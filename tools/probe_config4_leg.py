#!/usr/bin/env python3
"""A/B of bench.py's config4 leg (1080p, DPT-Large, 1024^3): sweeps overlapped or not x uploads prefetched or in line, and the batch size.
Usage (GPU box): python tools/probe_config4_leg.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from hive_amd import _lib  # noqa: E402

torch.cuda.set_device(0)
device = torch.device("cuda", 0)
ctx = _lib.default_context(0)
for batch, steps in ((8, 3), (16, 2)):
    for overlap in (False, True):
        for prefetch in (False, True):
            r = bench.config4_leg(device, ctx, steps=steps, batch=batch, overlap=overlap, prefetch=prefetch)
            print(json.dumps({"batch": batch, "overlap": overlap, "prefetch": prefetch, "frames_per_s": round(r["value"], 1), "ms_per_step": round(r["ms_per_step"], 2),
                              "dpt_ms_per_frame": round(r["dpt_ms_per_frame"], 3), "sweep_us_per_frame": round(r["roofline"]["us_per_frame"], 1)}), flush=True)
