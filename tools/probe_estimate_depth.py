#!/usr/bin/env python3
"""End-to-end throughput of `estimate_depth_dpt` (/root/reference/hive/dataset_adaptors.py:1346-1435: checkpoint on disk -> one 16-bit PNG per frame): 150 synthetic
640 x 480 frames and 24 frames of 1920 x 1080 held in memory, seeded weights, float16 (optimize=True), batch sizes 8 and 32.  What bounds it is the PNG encoder (zlib, one
core per file), so the files are written by a thread pool behind the GPU batches (round 5); `serial` times the same encodes one after the other for comparison.
Usage (GPU box): python tools/probe_estimate_depth.py > profiles/r05_estimate_depth.json"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from hive_amd import depth as depth_mod, synthetic  # noqa: E402
from hive_amd.dpt.init import seeded_init  # noqa: E402
from hive_amd.dpt.models import DPTDepthModel  # noqa: E402

out = {"cores": len(os.sched_getaffinity(0)), "rows": []}
with tempfile.TemporaryDirectory() as tmp:
    src = seeded_init(DPTDepthModel(path=None, engine="torch"), seed=1234).eval()
    torch.save(src.state_dict(), os.path.join(tmp, "dpt_hybrid_nyu.pt"))
    os.environ["WEIGHTS_PATH"] = tmp
    for (h, w, n) in ((480, 640, 150), (1080, 1920, 24)):
        seq = synthetic.make_sequence(num_frames=min(n, 24), height=h, width=w, yaw_step_deg=2.4)
        frames = [seq["color"][i % len(seq["color"])] for i in range(n)]
        for bs in (8, 32):
            dst = os.path.join(tmp, f"depth_{h}_{bs}")
            depth_mod.estimate_depth_dpt(frames[:bs], dst, batch_size=bs)  # warm-up (weights, arena, imports)
            t0 = time.perf_counter()
            depth_mod.estimate_depth_dpt(frames, dst, batch_size=bs)
            dt = time.perf_counter() - t0
            out["rows"].append({"frame": [h, w], "frames": n, "batch_size": bs, "seconds": dt, "frames_per_s": n / dt})
        # the PNG encoder alone, serially: what one thread writes per second
        from PIL import Image
        files = sorted(os.listdir(dst))[:16]
        imgs = [np.asarray(Image.open(os.path.join(dst, f))) for f in files]
        t0 = time.perf_counter()
        for i, im in enumerate(imgs):
            depth_mod._write_png16(os.path.join(tmp, f"serial_{i}.png"), im)
        out["rows"].append({"frame": [h, w], "png_encode_ms_per_frame_one_thread": (time.perf_counter() - t0) / len(imgs) * 1e3})
print(json.dumps(out, indent=1))
