#!/usr/bin/env python3
"""One DPT-Large (vitl16_384) forward of the hip engine at 480 x 864 (the reference's resize of a 1080p frame), for a kernel trace:
which kernels of BASELINE config 4's network are still PyTorch's / a vendor library's.  Usage: rocprofv3 --kernel-trace --stats -- python3 tools/list_kernels_large.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_amd.dpt.models import DPTDepthModel
m = DPTDepthModel(path=None, scale=1.0, shift=0.0, invert=False, backbone="vitl16_384", engine="hip").eval()
m = m.to(memory_format=torch.channels_last).to(torch.bfloat16).cuda()
x = torch.randn(2, 3, 480, 864, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(3):
        d = m(x)
torch.cuda.synchronize()
print(d.shape, float(d.mean()))
