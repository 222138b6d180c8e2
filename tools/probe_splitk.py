#!/usr/bin/env python3
"""Split-K of the 128-row GEMM tile at small M: us per hive_vit_linear call for HIVE_SPLITK = 0 (off) / n ways, on the ViT's shapes at one and
two frames (M = 1216 / 2432).  The weights rotate over 8 copies so that a call does not find them in L2 (in a forward every weight is read once)."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_amd import _lib  # noqa: E402

ctx = _lib.default_context(0)
half = torch.float16
code = _lib.dtype_code(half)
out = []
for M in (1216, 2432, 4864):
    for name, N, K, epi in (("qk", 1536, 768, 0), ("proj", 768, 768, 2), ("fc1", 3072, 768, 1), ("fc2", 768, 3072, 2)):
        A = torch.randn(M, K, device="cuda").to(half)
        Ws = [(torch.randn(N, K, device="cuda") / K ** 0.5).to(half) for _ in range(8)]
        bias = torch.randn(N, device="cuda") * 0.1
        res = torch.randn(M, N, device="cuda").to(half)
        C = torch.empty(M, N, device="cuda", dtype=half)
        row = {"M": M, "gemm": name, "N": N, "K": K}
        for ring, ways in (("2", "0"), ("4", "0"), ("4", "2"), ("4", "3"), ("4", "4")):
            os.environ["HIVE_SPLITK"] = ways
            os.environ["HIVE_GEMM_RING"] = ring
            def call(i):
                ctx.check(ctx.lib.hive_vit_linear(ctx.handle, A.data_ptr(), code, Ws[i % 8].data_ptr(), bias.data_ptr(), res.data_ptr() if epi == 2 else None, C.data_ptr(), M, N, K, epi))
            for i in range(8):
                call(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(64):
                call(i)
            e1.record()
            e1.synchronize()
            row[f"us_ring{ring}_split{ways}"] = round(e0.elapsed_time(e1) * 1e3 / 64, 2)
        out.append(row)
        print(json.dumps(row), flush=True)
