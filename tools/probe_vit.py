#!/usr/bin/env python3
"""Probe: per-kernel time / TFLOP/s of the HIP ViT kernels at the DPT shapes (torch events, 20 reps)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_amd import _lib  # noqa: E402
DT = int(__import__("os").environ.get("HIVE_PROBE_DTYPE", "2"))  # hive_dtype of the operands: 2 = bf16 (default), 1 = f16

ap = argparse.ArgumentParser()
ap.add_argument("--batches", default="1,8")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--torch", action="store_true", help="also time torch (hipBLASLt) for the same GEMMs")
args = ap.parse_args()
ctx = _lib.default_context(0)
lib = ctx.lib
D, H, F = 768, 12, 3072


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.reps * 1e-3


for B in [int(b) for b in args.batches.split(",")]:
    N, Np = 1201, 1216
    M = B * Np
    bf = dict(device="cuda", dtype=torch.bfloat16 if DT == 2 else torch.float16)
    x = torch.randn(M, D, **bf)
    res = torch.randn(M, D, **bf)
    hid = torch.randn(M, F, **bf)
    w_qkv, w_proj, w_fc1, w_fc2 = (torch.randn(3 * D, D, **bf) * 0.03, torch.randn(D, D, **bf) * 0.03, torch.randn(F, D, **bf) * 0.03,
                                   torch.randn(D, F, **bf) * 0.02)
    b_qkv, b_d, b_f = torch.zeros(3 * D, device="cuda"), torch.zeros(D, device="cuda"), torch.zeros(F, device="cuda")
    g = torch.ones(D, device="cuda")
    qk = torch.empty(M, 2 * D, **bf)
    vT = torch.empty(B, H, 64, Np, **bf)
    out = torch.empty(M, D, **bf)
    outf = torch.empty(M, F, **bf)
    rows = [
        ("layernorm", lambda: lib.hive_vit_layernorm(ctx.handle, x.data_ptr(), DT, g.data_ptr(), b_d.data_ptr(), out.data_ptr(), M, D, 1e-6), 0),
        ("qkv gemm", lambda: lib.hive_vit_qkv(ctx.handle, x.data_ptr(), DT, w_qkv.data_ptr(), b_qkv.data_ptr(), qk.data_ptr(), vT.data_ptr(), B, Np, D, H),
         2 * M * D * 3 * D),
        ("attention", lambda: lib.hive_vit_attention(ctx.handle, qk.data_ptr(), DT, vT.data_ptr(), out.data_ptr(), B, N, Np, D, H), 4 * B * H * N * N * 64),
        ("proj gemm+res", lambda: lib.hive_vit_linear(ctx.handle, x.data_ptr(), DT, w_proj.data_ptr(), b_d.data_ptr(), res.data_ptr(), out.data_ptr(), M, D, D, 2),
         2 * M * D * D),
        ("fc1 gemm+gelu", lambda: lib.hive_vit_linear(ctx.handle, x.data_ptr(), DT, w_fc1.data_ptr(), b_f.data_ptr(), None, outf.data_ptr(), M, F, D, 1),
         2 * M * D * F),
        ("fc1 gemm", lambda: lib.hive_vit_linear(ctx.handle, x.data_ptr(), DT, w_fc1.data_ptr(), b_f.data_ptr(), None, outf.data_ptr(), M, F, D, 0),
         2 * M * D * F),
        ("fc2 gemm+res", lambda: lib.hive_vit_linear(ctx.handle, hid.data_ptr(), DT, w_fc2.data_ptr(), b_d.data_ptr(), res.data_ptr(), out.data_ptr(), M, D, F, 2),
         2 * M * D * F),
    ]
    if args.torch:
        rows += [("torch fc1", lambda: torch.nn.functional.linear(x, w_fc1, b_f.bfloat16()), 2 * M * D * F),
                 ("torch fc2", lambda: torch.nn.functional.linear(hid, w_fc2, b_d.bfloat16()), 2 * M * D * F),
                 ("torch qkv", lambda: torch.nn.functional.linear(x, w_qkv, b_qkv.bfloat16()), 2 * M * D * 3 * D)]
    print(f"B={B} M={M}")
    total = 0.0
    for name, fn, flops in rows:
        dt = timed(fn)
        if not name.startswith("torch") and name != "fc1 gemm":
            total += dt * (2 if name == "layernorm" else 1)
        print(f"  {name:16s} {dt * 1e6:9.1f} us" + (f"  {flops / dt / 1e12:7.1f} TFLOP/s" if flops else ""))
    print(f"  one block (sum)  {total * 1e6:9.1f} us -> 12 blocks {total * 12e3:.2f} ms")
