"""Where does the bf16 DPT deviate from float32?  Per-module relative error of (a) the bf16 PyTorch-op engine and
(b) the bf16 HIP engine against the float32 model, same seeded weights, 480 x 640."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from dpt_weights import seeded_init, seeded_input
from hive_amd.dpt.models import DPTDepthModel, Bottleneck, Block, ResNetV2Stem, FeatureFusionBlock, GroupNormAct, StdConv2dSame

def build(engine, dtype):
    m = DPTDepthModel(path=None, scale=0.000305, shift=0.1378, invert=True, engine=engine).eval()
    seeded_init(m, 1234)
    if dtype is not None:
        m = m.to(memory_format=torch.channels_last).to(dtype)
    return m.cuda()

def capture(model, x, kinds):
    outs, hooks = {}, []
    for name, mod in model.named_modules():
        if isinstance(mod, kinds):
            hooks.append(mod.register_forward_hook(lambda m, i, o, name=name: outs.__setitem__(name, (o[0] if isinstance(o, tuple) else o).detach().float().cpu())))
    st = {}
    with torch.no_grad():
        d = model(x, stages=st)
    for h in hooks: h.remove()
    for k, v in st.items(): outs["stage." + k] = v.detach().float().cpu()
    outs["depth"] = d.float().cpu()
    return outs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
x = seeded_input(B, 480, 640, seed=7).bfloat16().float()
kinds = (Bottleneck, ResNetV2Stem, FeatureFusionBlock) + ((GroupNormAct, StdConv2dSame) if "--fine" in sys.argv else ())
ref = capture(build("torch", None), x.cuda(), kinds)
xb = x.cuda().bfloat16().contiguous(memory_format=torch.channels_last)
tor = capture(build("torch", torch.bfloat16), xb, kinds)
hip = capture(build("hip", torch.bfloat16), xb, kinds)
rel = lambda a, b: float((a - b).norm() / b.norm())
print(f"{'module':70s} {'bf16 torch':>10s} {'bf16 hip':>10s} {'hip vs torch':>12s}  ref std")
for k in ref:
    if k in tor and k in hip and tor[k].shape == ref[k].shape:
        print(f"{k:70s} {rel(tor[k], ref[k]):10.4f} {rel(hip[k], ref[k]):10.4f} {rel(hip[k], tor[k]):12.4f}  {float(ref[k].std()):.3f}")
