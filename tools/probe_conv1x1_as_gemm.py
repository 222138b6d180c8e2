#!/usr/bin/env python3
"""The ViT GEMM shapes run as 1 x 1 convolutions through the persistent implicit-GEMM kernel (csrc/conv.hip), next to
hive_vit_linear's kernels: does the cross-tile prefetch of the conv kernel pay at K = 768?  Usage: python tools/probe_conv1x1_as_gemm.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from hive_amd.dpt import ops
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters
for B in (24, 16, 48):
    for cin, cout in ((768, 1536), (768, 768), (768, 3072), (3072, 768)):
        x = torch.randn(B, cin, 32, 38, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
        conv = nn.Conv2d(cin, cout, 1, 1, 0, bias=True).to(memory_format=torch.channels_last).to(torch.bfloat16).cuda()
        assert ops.conv_eligible(x, conv)
        with torch.no_grad():
            t = bench(lambda: ops.conv2d(x, conv))
        M = B * 32 * 38
        print(f"M={M} N={cout} K={cin}: conv1x1 {t*1e3:8.1f} us {2.0*M*cin*cout/t/1e9:7.1f} TF/s")
