#!/usr/bin/env python3
"""Bandwidth of the HBM-bound passes of the network at the bench's shapes (64 frames): GroupNorm (3 kernels: statistics read x,
apply reads x and writes y), x2 upsampling, LayerNorm -- next to a plain device copy of the same bytes on the same box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_amd.dpt import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
def bench(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for (c, h, w) in [(256, 120, 160), (64, 120, 160), (128, 120, 160), (512, 60, 80), (128, 60, 80), (1024, 30, 40), (256, 30, 40)]:
    x = torch.randn(B, c, h, w, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    g = torch.ones(c, device="cuda").bfloat16(); b = torch.zeros(c, device="cuda").bfloat16()
    y = torch.empty_like(x)
    nbytes = x.numel() * 2
    t = bench(lambda: ops.group_norm_act(x, 32, g, b, 1e-5, relu=True, engine="hip"))
    tc = bench(lambda: y.copy_(x))
    print(f"GN  C={c:5d} {h}x{w}: {nbytes/1e6:7.1f} MB  group_norm {t*1e6:8.1f} us ({3*nbytes/t/1e12:5.2f} TB/s over 3 passes) | copy {tc*1e6:8.1f} us ({2*nbytes/tc/1e12:5.2f} TB/s)")
for (c, h, w) in [(256, 120, 160), (256, 60, 80)]:
    x = torch.randn(B, c, h, w, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    t = bench(lambda: ops.upsample2x(x, engine="hip"))
    nbytes = x.numel() * 2
    print(f"UP  C={c:5d} {h}x{w}: read {nbytes/1e6:7.1f} MB write {4*nbytes/1e6:7.1f} MB  {t*1e6:8.1f} us ({5*nbytes/t/1e12:5.2f} TB/s)")
    z = torch.empty(B, c, 2 * h, 2 * w, device="cuda", dtype=torch.bfloat16)
    tf = bench(lambda: z.zero_())
    print(f"    fill of the output alone: {tf*1e6:8.1f} us ({4*nbytes/tf/1e12:5.2f} TB/s)")
