#!/usr/bin/env python3
"""Time the convolutions of one ResNetV2 bottleneck (stage 1: 120 x 160, 256 / 64 channels; stage 3: 30 x 40, 1024 / 256) at the bench batch,
straight through the C ABI.  HIVE_AMD_LIB=hive_amd/lib/libhive_conv_abN.so times a tuning build (make -C hive_amd/csrc ablate_conv).
Usage: python tools/probe_resnet.py [batch]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 107
DT = torch.bfloat16
ctx = _lib.default_context(0)
lib = ctx.lib


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


def case(name, h, w, cin, cout, k, mode):
    x = (torch.randn(B, h, w, cin, device="cuda") * 0.5).to(DT)
    wt = (torch.randn(cout, k, k, cin, device="cuda") * 0.05).to(DT)
    out = torch.empty(B, h, w, cout, device="cuda", dtype=DT)
    res = (torch.randn(B, h, w, cout, device="cuda") * 0.5).to(DT)
    gamma, beta = torch.ones(cout, device="cuda", dtype=DT), torch.zeros(cout, device="cuda", dtype=DT)
    nfl = int(lib.hive_nhwc_conv_gn_partial_floats(B * h * w, cout)) + 2 * B * 32
    scratch = torch.empty(nfl, dtype=torch.float32, device="cuda")
    rows, fused = ctypes.c_int(0), ctypes.c_int(0)
    pad = k // 2
    if mode == "stats":  # conv + statistics in the epilogue (conv1 / conv2 of a bottleneck)
        def fn():
            ctx.check(lib.hive_nhwc_conv_gn(ctx.handle, x.data_ptr(), _lib.BF16, B, h, w, cin, cout, k, 1, pad, pad, h, w, wt.data_ptr(), None, 0, None, None,
                                            out.data_ptr(), None, scratch.data_ptr(), nfl, ctypes.byref(rows)))
    elif mode == "two_pass":  # conv3: statistics pass + normalising pass (+ shortcut + ReLU)
        def fn():
            ctx.check(lib.hive_nhwc_conv_gn_apply(ctx.handle, x.data_ptr(), _lib.BF16, B, h, w, cin, cout, k, 1, pad, pad, h, w, wt.data_ptr(), 32, gamma.data_ptr(),
                                                  beta.data_ptr(), 1e-5, res.data_ptr(), 1, out.data_ptr(), scratch.data_ptr(), nfl, ctypes.byref(fused)))
    elif mode == "rcu":  # decoder: conv2 of a residual unit, out = conv + x + skip, and relu(out) beside it
        res2, out2 = res.clone(), torch.empty_like(out)

        def fn():
            ctx.check(lib.hive_nhwc_conv(ctx.handle, x.data_ptr(), _lib.BF16, B, h, w, cin, cout, k, 1, pad, pad, h, w, wt.data_ptr(), None, 0, res.data_ptr(),
                                         res2.data_ptr(), out.data_ptr(), out2.data_ptr()))
    else:  # plain convolution with bias-less epilogue
        def fn():
            ctx.check(lib.hive_nhwc_conv(ctx.handle, x.data_ptr(), _lib.BF16, B, h, w, cin, cout, k, 1, pad, pad, h, w, wt.data_ptr(), None, 0, None, None,
                                         out.data_ptr(), None))
    us = timed(fn)
    m = B * h * w
    flops = 2.0 * m * cin * k * k * cout * (2 if mode == "two_pass" else 1)
    byts = 2.0 * m * (cin * (2 if mode == "two_pass" else 1) + cout * (2 if mode == "two_pass" else 1))  # two-pass: input twice, shortcut + output
    print(f"{name:34s} {mode:9s} {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  {byts / us / 1e3:7.1f} GB/s  tiles/CU {m / 256 * max(cout // 256, 1) / 256:6.1f}", flush=True)


print("library:", _lib.LIB_PATH)
case("stage1 conv1 1x1 256->64", 120, 160, 256, 64, 1, "stats")
case("stage1 conv2 3x3 64->64", 120, 160, 64, 64, 3, "stats")
case("stage1 conv3 1x1 64->256", 120, 160, 64, 256, 1, "two_pass")
case("stage1 conv3 1x1 64->256", 120, 160, 64, 256, 1, "plain")
case("refinenet1 3x3 256->256", 120, 160, 256, 256, 3, "plain")
case("refinenet1 3x3 256->256", 120, 160, 256, 256, 3, "rcu")
case("stage2 conv1 1x1 512->128", 60, 80, 512, 128, 1, "stats")
case("stage2 conv2 3x3 128->128", 60, 80, 128, 128, 3, "stats")
case("stage2 conv3 1x1 128->512", 60, 80, 128, 512, 1, "two_pass")
case("stage3 conv1 1x1 1024->256", 30, 40, 1024, 256, 1, "stats")
case("stage3 conv2 3x3 256->256", 30, 40, 256, 256, 3, "stats")
case("stage3 conv3 1x1 256->1024", 30, 40, 256, 1024, 1, "two_pass")
case("stage3 conv3 1x1 256->1024", 30, 40, 256, 1024, 1, "plain")
