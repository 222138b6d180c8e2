#!/usr/bin/env python3
"""fc2 (M = 130112, N = 768, K = 3072) and proj (K = 768) a few times each: for rocprofv3 --pmc FETCH_SIZE (how often is the A panel read from beyond L2?)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from hive_amd import _lib  # noqa: E402

ctx = _lib.default_context(0)
lib = ctx.lib
M, N = 130112, 768
for K in (3072, 768):
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    b = torch.zeros(N, device="cuda")
    R = torch.zeros(M, N, device="cuda").bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(4):
        ctx.check(lib.hive_vit_linear(ctx.handle, A.data_ptr(), 2, W.data_ptr(), b.data_ptr(), R.data_ptr(), C.data_ptr(), M, N, K, 2))
    torch.cuda.synchronize()
