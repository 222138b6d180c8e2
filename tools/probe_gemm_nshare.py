#!/usr/bin/env python3
"""Does the long-K GEMM (fc2: K = 3072) lose its A-panel sharing between the N tiles of an M block?  Time per 64-deep K-step and round of 256 CUs for N = 256 (one N tile:
nothing to share), 512, 768, at K = 768 and 3072.  Usage (GPU box): [HIVE_AMD_LIB=.../libhive_kloop.so] python tools/probe_gemm_nshare.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hive_amd import _lib  # noqa: E402

ctx = _lib.default_context(0)
lib = ctx.lib
M = 130112


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for K in (768, 3072):
    A = torch.randn(M, K, device="cuda").bfloat16()
    for N in (256, 512, 768, 1536):
        W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
        b = torch.zeros(N, device="cuda")
        R = torch.zeros(M, N, device="cuda").bfloat16()
        C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        us = sorted(timed(lambda: ctx.check(lib.hive_vit_linear(ctx.handle, A.data_ptr(), 2, W.data_ptr(), b.data_ptr(), R.data_ptr(), C.data_ptr(), M, N, K, 2))) for _ in range(3))[1]
        tiles = 509 * (N // 256)
        rounds = tiles / 256.0
        print(f"K = {K:5d} N = {N:5d}: {us:7.1f} us, {tiles} tiles = {rounds:5.2f} rounds, {us / (K // 64) / rounds:6.3f} us per K-step and round, {2 * M * N * K / us / 1e6:6.0f} TFLOP/s", flush=True)
