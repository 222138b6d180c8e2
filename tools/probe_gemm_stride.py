#!/usr/bin/env python3
"""Is fc2's K loop (K = 3072: 1.76 us per 64-deep step against 1.47 for the K = 768 GEMMs) slowed by its operands' row pitch (6144 B = 3 x 2^11: every row of an
8-row LDS-DMA piece lands on the same few L2 channels)?  hive_vit_linear at N = 768 for K = 3072 and for neighbouring K whose pitch is not such a multiple (same
work per step; time per K-step is what is compared).  Usage (GPU box): [HIVE_AMD_LIB=.../libhive_kloop.so] python tools/probe_gemm_stride.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hive_amd import _lib  # noqa: E402

ctx = _lib.default_context(0)
lib = ctx.lib
M, N = 130112, 768


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


res = {}
for rnd in range(3):
    for K in (3072, 3136, 3008, 2944, 3200, 768, 832):
        A = torch.randn(M, K, device="cuda").bfloat16()
        W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
        b = torch.zeros(N, device="cuda")
        R = torch.zeros(M, N, device="cuda").bfloat16()
        C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        us = timed(lambda: ctx.check(lib.hive_vit_linear(ctx.handle, A.data_ptr(), 2, W.data_ptr(), b.data_ptr(), R.data_ptr(), C.data_ptr(), M, N, K, 2)))
        res.setdefault(K, []).append(us)
        del A, W, R, C
for K, v in res.items():
    us = sorted(v)[len(v) // 2]
    steps = K // 64
    print(f"K = {K:5d} (pitch {2 * K} B): {us:7.1f} us, {us / steps / 5.96:6.3f} us per K-step and round, {2 * M * N * K / us / 1e6:6.0f} TFLOP/s", flush=True)
