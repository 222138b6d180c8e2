#!/usr/bin/env python3
"""Build note behind two bounds of tests/test_dpt_gpu.py::test_batch_independence_and_determinism, measured once over 8 seeds (not in CI):
  (a) the ViT engine on ONE image with split-K (fc2's K loop dealt to four workgroups) vs without: relative Frobenius difference of the last tap;
  (b) the whole model, a frame inside a batch of 6 vs the frame alone: median depth difference in millimetres (bf16 and fp16).
Usage (GPU box): python tools/diag_splitk_bound.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

from dpt_weights import seeded_init, seeded_input  # noqa: E402
from hive_amd import _lib  # noqa: E402
from hive_amd.dpt.models import DPTDepthModel  # noqa: E402
from hive_amd.dpt.vit_engine import VitEngine  # noqa: E402

rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
med = lambda a, b: float(((a - b).abs() * 1000.0).flatten().median())
for dtype in (torch.bfloat16, torch.float16):
    ref = DPTDepthModel(path=None, scale=0.000305, shift=0.1378, invert=True, engine="torch").eval()
    seeded_init(ref, seed=1234)
    hip = DPTDepthModel(path=None, scale=0.000305, shift=0.1378, invert=True, engine="hip").eval()
    hip.load_state_dict(ref.state_dict())
    hip = hip.to(memory_format=torch.channels_last).to(dtype).cuda()
    ctx = _lib.default_context(0)
    eng = VitEngine(hip.pretrained.model, ctx=ctx)
    a, b = [], []
    for seed in range(8):
        torch.manual_seed(seed)
        tokens = torch.randn(1, 1201, 768, device="cuda").to(dtype)
        t_split = eng.forward(tokens, taps=(8, 11))
        ctx.set_deterministic(True)
        t_one = eng.forward(tokens, taps=(8, 11))
        ctx.set_deterministic(False)
        a.append(rel(t_split[1], t_one[1]))
        x = seeded_input(6, 480, 640, seed=100 + seed).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            d_all = hip(x)
            d_one = hip(x[4:5].contiguous(memory_format=torch.channels_last))
        b.append(med(d_all[4], d_one[0]))
    print(dtype, "split vs unsplit ViT tap_4, relative:", [round(v, 5) for v in a], "max", round(max(a), 5))
    print(dtype, "frame in a batch of 6 vs alone, median mm:", [round(v, 3) for v in b], "max", round(max(b), 3), flush=True)
