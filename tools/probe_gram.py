#!/usr/bin/env python3
"""conv3 / downsample + GroupNorm + shortcut + ReLU of the ResNetV2 stages at the bench batch: the two-pass form (statistics from a first pass of the convolution,
hive_nhwc_conv_gn_apply) against the Gram form (statistics from the input's Gram matrices, hive_nhwc_conv_gn_apply_gram), interleaved round-robin in one
process (medians of 7 rounds of 6 calls), and the statistics kernels alone."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 107
ctx = _lib.default_context(0)
lib, G, dt, code = ctx.lib, 32, torch.bfloat16, _lib.dtype_code(torch.bfloat16)


def timed(fn, reps=6):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for name, cin, cout, stride, h, w in (("stage 1 conv3", 64, 256, 1, 120, 160), ("stage 2 conv3", 128, 512, 1, 60, 80), ("stage 3 conv3", 256, 1024, 1, 30, 40),
                                      ("stage 2 downsample", 256, 512, 2, 120, 160)):
    oh, ow = (h + stride - 1) // stride, (w + stride - 1) // stride
    x = torch.relu(torch.randn(B, h, w, cin, device="cuda") + 0.3).to(dt)
    wt = (torch.randn(cout, cin, device="cuda") / cin ** 0.5).to(dt)
    gamma, beta = torch.ones(cout, device="cuda", dtype=dt), torch.zeros(cout, device="cuda", dtype=dt)
    res = torch.randn(B, oh, ow, cout, device="cuda").to(dt)
    out = torch.empty_like(res)
    tables = torch.empty(int(lib.hive_gn_gram_table_floats(cin, G)), dtype=torch.float32, device="cuda")
    ctx.check(lib.hive_gn_gram_prepare(ctx.handle, wt.data_ptr(), code, cin, cout, G, tables.data_ptr()))
    nfl = int(lib.hive_nhwc_conv_gn_partial_floats(B * oh * ow, cout)) + 2 * B * G
    scratch = torch.empty(nfl, dtype=torch.float32, device="cuda")
    stats = torch.empty(B * G * 2, dtype=torch.float32, device="cuda")
    fused = ctypes.c_int(0)
    two = lambda: ctx.check(lib.hive_nhwc_conv_gn_apply(ctx.handle, x.data_ptr(), code, B, h, w, cin, cout, 1, stride, 0, 0, oh, ow, wt.data_ptr(), G, gamma.data_ptr(),
                                                        beta.data_ptr(), 1e-5, res.data_ptr(), 1, out.data_ptr(), scratch.data_ptr(), nfl, ctypes.byref(fused)))
    gram = lambda: ctx.check(lib.hive_nhwc_conv_gn_apply_gram(ctx.handle, x.data_ptr(), code, B, h, w, cin, cout, stride, oh, ow, wt.data_ptr(), tables.data_ptr(), G,
                                                              gamma.data_ptr(), beta.data_ptr(), 1e-5, res.data_ptr(), 1, out.data_ptr(), scratch.data_ptr(), nfl,
                                                              ctypes.byref(fused)))
    only = lambda: ctx.check(lib.hive_gn_gram_stats(ctx.handle, x.data_ptr(), code, B, h, w, cin, cout, stride, oh, ow, G, tables.data_ptr(), 1e-5, stats.data_ptr(), None, None))
    for f in (two, gram, only):
        f()
    t = {"two-pass": [], "gram": [], "gram statistics alone": []}
    for _ in range(7):
        t["two-pass"].append(timed(two))
        t["gram"].append(timed(gram))
        t["gram statistics alone"].append(timed(only))
    print(f"{name:20s} {cin:4d} -> {cout:4d} at {oh} x {ow} x {B}: " + " | ".join(f"{k} {sorted(v)[3]:7.1f} us" for k, v in t.items()), flush=True)
