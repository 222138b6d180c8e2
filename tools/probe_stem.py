#!/usr/bin/env python3
"""Time the ResNetV2 stem kernels at the bench batch (hive_resnet_stem_conv_gn, hive_nhwc_group_norm_relu_maxpool).  HIVE_AMD_LIB=... for tuning builds."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 107
ctx = _lib.default_context(0)
lib = ctx.lib
x = (torch.rand(B, 480, 640, 3, device="cuda") * 2 - 1).bfloat16()
w = torch.zeros(64, 7, 32, device="cuda", dtype=torch.bfloat16)
w[:, :, :21] = (torch.randn(64, 7, 21, device="cuda") * 0.1).bfloat16()
out = torch.empty(B, 240, 320, 64, device="cuda", dtype=torch.bfloat16)
pooled = torch.empty(B, 120, 160, 64, device="cuda", dtype=torch.bfloat16)
nfl = int(lib.hive_nhwc_conv_gn_partial_floats(B * 240 * 320, 64))
partial = torch.empty(nfl, dtype=torch.float32, device="cuda")
gamma, beta = torch.ones(64, device="cuda", dtype=torch.bfloat16), torch.zeros(64, device="cuda", dtype=torch.bfloat16)
rows = ctypes.c_int(0)


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


print("library:", _lib.LIB_PATH)
print(f"stem conv + sums : {timed(lambda: ctx.check(lib.hive_resnet_stem_conv_gn(ctx.handle, x.data_ptr(), _lib.BF16, B, 480, 640, w.data_ptr(), out.data_ptr(), partial.data_ptr(), nfl, ctypes.byref(rows)))):8.1f} us")
print(f"stem conv        : {timed(lambda: ctx.check(lib.hive_resnet_stem_conv(ctx.handle, x.data_ptr(), _lib.BF16, B, 480, 640, w.data_ptr(), out.data_ptr()))):8.1f} us")
print(f"gn + relu + pool : {timed(lambda: ctx.check(lib.hive_nhwc_group_norm_relu_maxpool(ctx.handle, out.data_ptr(), _lib.BF16, B, 240, 320, 64, 32, gamma.data_ptr(), beta.data_ptr(), 1e-5, pooled.data_ptr(), partial.data_ptr(), rows.value))):8.1f} us")
