#!/usr/bin/env python3
"""Probe: hive_vit_linear time vs K / M / N (finds fixed overheads and scaling)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_amd import _lib
DT = int(__import__("os").environ.get("HIVE_PROBE_DTYPE", "2"))  # hive_dtype of the operands: 2 = bf16 (default), 1 = f16
ctx = _lib.default_context(0); lib = ctx.lib
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for (M, N, K) in [(9728, 3072, 64), (9728, 3072, 128), (9728, 3072, 256), (9728, 3072, 768), (9728, 3072, 3072), (9728, 768, 3072), (9728, 128, 768), (256, 3072, 768), (2048, 3072, 768), (4096, 4096, 4096)]:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = torch.randn(N, K, device="cuda").bfloat16(); b = torch.zeros(N, device="cuda")
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    dt = timed(lambda: lib.hive_vit_linear(ctx.handle, A.data_ptr(), DT, W.data_ptr(), b.data_ptr(), None, C.data_ptr(), M, N, K, 0))
    dtt = timed(lambda: torch.nn.functional.linear(A, W))
    print(f"M={M} N={N} K={K}: hive {dt*1e6:8.1f} us {2*M*N*K/dt/1e12:7.1f} TF/s | torch {dtt*1e6:8.1f} us {2*M*N*K/dtt/1e12:7.1f} TF/s")
