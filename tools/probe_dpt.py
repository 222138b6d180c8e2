#!/usr/bin/env python3
"""Probe: DPT-Hybrid forward time on the GPU by engine / dtype / batch, with a per-section breakdown."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_amd.dpt.models import DPTDepthModel, count_flops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--engine", default="torch")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--batches", default="1,4,8")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--sections", action="store_true")
args = ap.parse_args()
dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
torch.manual_seed(0)
model = DPTDepthModel(path=None, scale=0.000305, shift=0.1378, invert=True, engine=args.engine).eval()
model = model.to(memory_format=torch.channels_last).to(dtype).cuda()
flops = count_flops()["total"]
torch.backends.cudnn.benchmark = True


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / iters


with torch.no_grad():
    for b in [int(v) for v in args.batches.split(",")]:
        x = torch.randn(b, 3, 480, 640, device="cuda", dtype=dtype).contiguous(memory_format=torch.channels_last)
        dt = timed(lambda: model(x), args.iters)
        print(f"engine={args.engine} dtype={args.dtype} batch={b}: {dt * 1e3:.2f} ms/batch, {b / dt:.1f} frames/s, "
              f"{flops * b / dt / 1e12:.1f} TFLOP/s")
        if args.sections:
            p, vit, s = model.pretrained, model.pretrained.model, model.scratch
            bb = vit.patch_embed.backbone
            feat = bb.stem(x)
            l1 = bb.stages[0](feat)
            l2 = bb.stages[1](l1)
            f3 = bb.stages[2](l2)
            tokens = vit.patch_embed.proj(f3).flatten(2).transpose(1, 2)
            tokens = torch.cat((vit.cls_token.expand(b, -1, -1).to(tokens.dtype), tokens), dim=1).contiguous()
            print("  stem        %.2f ms" % (timed(lambda: bb.stem(x), args.iters) * 1e3))
            print("  stage0      %.2f ms" % (timed(lambda: bb.stages[0](feat), args.iters) * 1e3))
            print("  stage1      %.2f ms" % (timed(lambda: bb.stages[1](l1), args.iters) * 1e3))
            print("  stage2      %.2f ms" % (timed(lambda: bb.stages[2](l2), args.iters) * 1e3))
            print("  vit blocks  %.2f ms" % (timed(lambda: model._run_blocks(tokens), args.iters) * 1e3))
            l1, l2, l3, l4 = model.forward_backbone(x)
            print("  backbone    %.2f ms" % (timed(lambda: model.forward_backbone(x), args.iters) * 1e3))

            def decoder():
                p4 = s.refinenet4(s.layer4_rn(l4))
                p3 = s.refinenet3(p4, s.layer3_rn(l3))
                p2 = s.refinenet2(p3, s.layer2_rn(l2))
                p1 = s.refinenet1(p2, s.layer1_rn(l1))
                return p1
            p1 = decoder()
            print("  decoder     %.2f ms" % (timed(decoder, args.iters) * 1e3))
            print("  head        %.2f ms" % (timed(lambda: s.output_conv(p1), args.iters) * 1e3))
