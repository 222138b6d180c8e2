#!/usr/bin/env python3
"""Per-layer time of every convolution of DPT-Hybrid at batch B (forward hooks + CUDA events on a warm model):
shape, calls per forward, microseconds, TFLOP/s -- to see which MIOpen convolutions are far from the roofline."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_amd import depth as depth_mod  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.backends.cudnn.benchmark = True
model = depth_mod.build_model(None, dtype=torch.bfloat16)
x = torch.randn(B, 3, 480, 640, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
records = {}


def hook(name):
    def pre(mod, inp):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        mod._t0 = e

    def post(mod, inp, out):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        records.setdefault(name, []).append((mod._t0, e, tuple(inp[0].shape), tuple(out.shape), mod))
    return pre, post


for name, m in model.named_modules():
    if isinstance(m, torch.nn.Conv2d):
        pre, post = hook(name)
        m.register_forward_pre_hook(pre)
        m.register_forward_hook(post)
# convolutions issued through torch.nn.functional.conv2d (the fused bias/ReLU/skip units call it directly)
_orig_conv2d = torch.nn.functional.conv2d


class _Fn:
    def __init__(self, w, stride):
        self.kernel_size, self.stride, self.in_channels, self.groups = tuple(w.shape[2:]), stride, w.shape[1], 1


def conv2d_probe(inp, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    import traceback
    frames = [f for f in traceback.extract_stack(limit=6) if "hive_amd" in f.filename]
    if not frames or frames[-1].name == "_conv_forward":
        return _orig_conv2d(inp, weight, bias, stride, padding, dilation, groups)
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    out = _orig_conv2d(inp, weight, bias, stride, padding, dilation, groups)
    b.record()
    st = stride if isinstance(stride, tuple) else (stride, stride)
    key = f"F.conv2d@{frames[-1].name}:{tuple(inp.shape[1:])}->{tuple(out.shape[1:])}"
    records.setdefault(key, []).append((a, b, tuple(inp.shape), tuple(out.shape), _Fn(weight, st)))
    return out


torch.nn.functional.conv2d = conv2d_probe
with torch.no_grad():
    for _ in range(3):
        model(x)
    records.clear()
    for _ in range(3):
        model(x)
torch.cuda.synchronize()
rows = []
for name, recs in records.items():
    t = sum(a.elapsed_time(b) for a, b, *_ in recs) / 3.0 * 1e3
    _, _, ish, osh, m = recs[0]
    kh, kw = m.kernel_size
    flops = 2.0 * osh[0] * osh[1] * osh[2] * osh[3] * (m.in_channels // m.groups) * kh * kw * (len(recs) / 3.0)
    rows.append((t, name, ish, osh, (kh, kw), m.stride, flops / (t * 1e-6) / 1e12))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"B={B}: {len(rows)} conv layers, {tot / 1e3:.2f} ms per forward (event-bracketed, includes launch gaps)")
for t, name, ish, osh, k, s, tf in rows[:40]:
    print(f"{t:8.1f} us {tf:7.1f} TF/s  k{k[0]}x{k[1]} s{s[0]}  in {ish[1:]} out {osh[1:]}  {name}")
