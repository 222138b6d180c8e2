#!/usr/bin/env python3
"""One number: milliseconds per hive_dpt_forward at a batch (default 107 frames of 480 x 640, bf16), median of `reps` timed calls after 3.  The library comes from
HIVE_AMD_LIB (A/B of builds: tools/ab_forward.sh alternates processes on one box).  Usage: python tools/forward_ms.py [batch] [dtype] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hive_amd import depth as depth_mod  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 107
dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
model = depth_mod.build_model(None, dtype=dtype, init_seed=1234)
frames = torch.randint(0, 256, (batch, 480, 640, 3), dtype=torch.uint8, device="cuda")
ms = []
with torch.no_grad():
    for i in range(3 + reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        model.forward_frames(frames, max_depth=10.0)
        e1.record()
        e1.synchronize()
        if i >= 3:
            ms.append(e0.elapsed_time(e1))
ms.sort()
print(f"{os.path.basename(os.environ.get('HIVE_AMD_LIB', 'libhive_mi355x.so'))} batch {batch}: median {ms[len(ms) // 2]:.3f} ms, min {ms[0]:.3f}", flush=True)
