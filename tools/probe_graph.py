#!/usr/bin/env python3
"""Does a captured HIP graph buy anything for small batches?  hive_dpt_forward at B = 1 / 2 / 4 / 8 (DPT-Hybrid, 480 x 640, float16, frames resident):
eager C-ABI call vs replay of the same call captured with torch.cuda.CUDAGraph (the library launches on torch's current stream, so torch's capture sees
every launch).  Usage (GPU box): python tools/probe_graph.py [batches...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from hive_amd import depth as depth_mod, synthetic  # noqa: E402

batches = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
dtype = {"fp16": torch.float16, "bf16": torch.bfloat16}[os.environ.get("HIVE_PROBE_DTYPE", "fp16")]
model = depth_mod.build_model(None, device="cuda", dtype=dtype, engine="hip", init_seed=1234)
seq = synthetic.make_sequence(num_frames=8, height=480, width=640, seed=1234, yaw_step_deg=2.4)
frames = torch.from_numpy(seq["color"]).cuda()


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    t_issue = time.perf_counter() - t0
    e1.synchronize()
    return e0.elapsed_time(e1) / reps, t_issue / reps * 1e3


with torch.no_grad():
    for b in batches:
        fr = frames[:b].contiguous()
        native = model.native()
        out = {}

        def eager():
            out["r"] = native.forward(fr, max_depth=10.0)

        ms_eager, issue_eager = timed(eager)
        ref = [t.clone() for t in out["r"]]
        # capture: same call on a capture stream (buffers of the outputs are allocated inside the capture by torch's graph pool)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            eager()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            captured = native.forward(fr, max_depth=10.0)
        ms_graph, issue_graph = timed(g.replay)
        same = all(torch.equal(a, b_) for a, b_ in zip(ref, captured))
        print(f"B={b}: eager {ms_eager:.3f} ms ({ms_eager / b:.3f}/frame; CPU issue {issue_eager:.3f} ms)   graph {ms_graph:.3f} ms ({ms_graph / b:.3f}/frame; issue {issue_graph:.3f})   "
              f"identical={same}", flush=True)
