"""Are two identical forwards bit-identical?  ViT engine alone, glue ops, whole model (HIP engine), whole model (PyTorch bf16 ops)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from dpt_weights import seeded_init, seeded_input
from hive_amd.dpt.models import DPTDepthModel, VisionTransformerHybrid
from hive_amd.dpt.vit_engine import VitEngine

def build(engine):
    m = DPTDepthModel(path=None, scale=0.000305, shift=0.1378, invert=True, engine=engine).eval()
    seeded_init(m, 1234)
    return m.to(memory_format=torch.channels_last).to(torch.bfloat16).cuda()

def maxdiff(a, b): return float((a.float() - b.float()).abs().max())

hip = build("hip")
eng = VitEngine(hip.pretrained.model)
tok = torch.randn(4, 1201, 768, device="cuda").bfloat16()
outs = [eng.forward(tok, taps=(8, 11)) for _ in range(4)]
print("vit engine: run-to-run max |diff| tap11:", [maxdiff(outs[0][1], o[1]) for o in outs[1:]])
x = seeded_input(4, 480, 640, seed=7).cuda().bfloat16().contiguous(memory_format=torch.channels_last)
for name, model in (("hip", hip), ("torch-bf16", build("torch"))):
    with torch.no_grad():
        runs = []
        for r in range(4):
            st = {}
            d = model(x, stages=st)
            runs.append((d, st))
    for k in ("layer_1", "layer_2", "tokens", "tap_4", "layer_4", "path_4", "path_1", "head_in"):
        print(f"{name:10s} {k:8s} run1..3 vs run0 max|diff|:", [round(maxdiff(runs[0][1][k], r[1][k]), 6) for r in runs[1:]],
              " run3 vs run2:", round(maxdiff(runs[2][1][k], runs[3][1][k]), 6))
    print(f"{name:10s} depth mm run1..3 vs run0:", [round(maxdiff(runs[0][0], r[0]) * 1000, 3) for r in runs[1:]])
