#!/usr/bin/env python3
"""Time the fused depth-head kernel (hive_dpt_head_fused) at the bench shape: [B][240][320][128] -> [B][480][640]."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H, W = 240, 320
ctx = _lib.default_context(0)
x = (torch.randn(B, H, W, 128, device="cuda") * 0.5).bfloat16()
w3 = (torch.randn(3, 3, 32, 128, device="cuda") * 0.05).bfloat16()
b0 = torch.randn(128, device="cuda") * 0.1 if os.environ.get("HIVE_PROBE_HEAD_B0") else None  # the network passes NULL (output_conv[0] adds its bias itself)
b3 = np.random.randn(32).astype(np.float32) * 0.1
w1 = np.random.randn(32).astype(np.float32) * 0.3
depth = torch.empty((B, 2 * H, 2 * W), dtype=torch.float32, device="cuda")
mm = torch.empty((B, 2 * H, 2 * W), dtype=torch.int16, device="cuda")
m = torch.empty((B, 2 * H, 2 * W), dtype=torch.float32, device="cuda")


def run():
    ctx.check(ctx.lib.hive_dpt_head_fused(ctx.handle, x.data_ptr(), _lib.ptr(b0), _lib.BF16, B, H, W, 128, 32, w3.data_ptr(), b3.ctypes.data, w1.ctypes.data,
                                          0.05, 1, 1, 0.01, 0.1, depth.data_ptr(), 1e-3, 10.0, mm.data_ptr(), m.data_ptr()))


for _ in range(3):
    run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    run()
b.record()
torch.cuda.synchronize()
us = a.elapsed_time(b) * 100.0
flops = 2.0 * B * 4 * H * W * 128 * 32 * 9
print(f"B={B}: head_conv_kernel {us:.1f} us, {flops / us / 1e6:.1f} TFLOP/s (3x3 conv only)")
