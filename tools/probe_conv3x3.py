#!/usr/bin/env python3
"""TFLOP/s of the hand-written implicit-GEMM 3 x 3 convolution (csrc/conv.hip) on the decoder shapes of DPT-Hybrid at 480 x 640,
next to MIOpen (torch F.conv2d, cudnn.benchmark) on the same tensors.  Usage: python tools/probe_conv3x3.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, torch.nn.functional as F
from hive_amd.dpt import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
torch.backends.cudnn.benchmark = True
shapes = [("rcu 120x160", 256, 256, 120, 160), ("rcu 60x80", 256, 256, 60, 80), ("rcu 30x40", 256, 256, 30, 40), ("rcu 15x20", 256, 256, 15, 20),
          ("layer2_rn", 512, 256, 60, 80), ("layer3_rn", 768, 256, 30, 40), ("layer4_rn", 768, 256, 15, 20), ("head 240x320", 256, 128, 240, 320)]
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters
tot_h = tot_m = 0.0
for name, cin, cout, h, w in shapes:
    x = torch.randn(B, cin, h, w, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    conv = nn.Conv2d(cin, cout, 3, 1, 1, bias=True).to(memory_format=torch.channels_last).to(torch.bfloat16).cuda()
    flops = 2.0 * B * h * w * cout * 9 * cin
    with torch.no_grad():
        t_h = bench(lambda: ops.conv3x3(x, conv, relu=True))
        t_m = bench(lambda: F.conv2d(x, conv.weight, None, 1, 1))
    tot_h += t_h; tot_m += t_m
    print(f"{name:14s} M={B*h*w:8d} tiles={(B*h*w+255)//256:5d}  hip {t_h*1e3:8.1f} us {flops/t_h/1e9:7.1f} TFLOP/s | MIOpen {t_m*1e3:8.1f} us {flops/t_m/1e9:7.1f} TFLOP/s")
print(f"sum: hip {tot_h:.3f} ms, MIOpen {tot_m:.3f} ms")
