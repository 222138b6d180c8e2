#!/usr/bin/env python3
"""Soak: the same forward many times, every result compared bit for bit with the first -- the small-batch paths (split-K's last arriver, the attention's key-split exchange through LDS,
the merged q | k | v^T launch) and the bench batch.  Usage (GPU box): python tools/soak_determinism.py [reps_small=300] [reps_large=20]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hive_amd import depth as depth_mod  # noqa: E402

reps_small = int(sys.argv[1]) if len(sys.argv) > 1 else 300
reps_large = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for dtype in (torch.float16, torch.bfloat16):
    model = depth_mod.build_model(None, dtype=dtype, init_seed=1234)
    for batch, reps in ((1, reps_small), (2, reps_small // 2), (3, reps_small // 3), (8, reps_small // 6), (107, reps_large)):
        frames = torch.randint(0, 256, (batch, 480, 640, 3), dtype=torch.uint8, device="cuda")
        with torch.no_grad():
            first = [t.clone() for t in model.forward_frames(frames, max_depth=10.0) if t is not None]
            bad = 0
            for i in range(reps):
                out = [t for t in model.forward_frames(frames, max_depth=10.0) if t is not None]
                if not all(torch.equal(a, b) for a, b in zip(first, out)):
                    bad += 1
        torch.cuda.synchronize()
        print(f"{str(dtype):16s} batch {batch:4d}: {reps} repeats, {bad} differ from the first", flush=True)
        assert bad == 0
print("soak ok")
