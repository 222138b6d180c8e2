#!/usr/bin/env python3
"""Time hive_bneck_gn_conv3x3 (csrc/bneck.hip) at the bench shape: 107 x 120 x 160 x 64.  HIVE_AMD_LIB=... for tuning builds."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_amd import _lib  # noqa: E402

B, H, W = (int(sys.argv[1]) if len(sys.argv) > 1 else 107), 120, 160
ctx = _lib.default_context(0)
lib = ctx.lib
t = (torch.randn(B, H, W, 64, device="cuda") * 0.7).bfloat16()
w1 = (torch.randn(64, 1, 1, 64, device="cuda") * 0.1).bfloat16()
w2 = (torch.randn(64, 3, 3, 64, device="cuda") * 0.05).bfloat16()
gamma, beta = torch.ones(64, device="cuda", dtype=torch.bfloat16), torch.zeros(64, device="cuda", dtype=torch.bfloat16)
nfl = int(lib.hive_nhwc_conv_gn_partial_floats(B * H * W, 64))
p_in, p_out = torch.empty(nfl, dtype=torch.float32, device="cuda"), torch.empty(nfl, dtype=torch.float32, device="cuda")
t1, out = torch.empty_like(t), torch.empty_like(t)
rows_in, rows_out, fused = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
ctx.check(lib.hive_nhwc_conv_gn(ctx.handle, t.data_ptr(), _lib.BF16, B, H, W, 64, 64, 1, 1, 0, 0, H, W, w1.data_ptr(), None, 0, None, None, t1.data_ptr(), None, p_in.data_ptr(),
                                nfl, ctypes.byref(rows_in)))


def run():
    ctx.check(lib.hive_bneck_gn_conv3x3(ctx.handle, t1.data_ptr(), _lib.BF16, B, H, W, 64, p_in.data_ptr(), rows_in.value, gamma.data_ptr(), beta.data_ptr(), 1e-5,
                                        w2.data_ptr(), out.data_ptr(), p_out.data_ptr(), nfl, ctypes.byref(rows_out), ctypes.byref(fused)))


for _ in range(3):
    run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    run()
b.record()
torch.cuda.synchronize()
print(f"bneck gn + conv3x3 (finalize + kernel): {a.elapsed_time(b) * 100:.1f} us, fused {fused.value}, lib {os.path.basename(_lib.LIB_PATH)}")
