import os, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from dpt_weights import seeded_init
from hive_amd.dpt.models import DPTDepthModel
half = torch.bfloat16
ref32 = DPTDepthModel(path=None, scale=0.000305, shift=0.1378, invert=True, engine="torch").eval()
seeded_init(ref32, seed=4)
def mk(engine):
    m = DPTDepthModel(path=None, scale=0.000305, shift=0.1378, invert=True, engine=engine).eval()
    m.load_state_dict(ref32.state_dict())
    return m.to(memory_format=torch.channels_last).to(half).cuda()
tor = mk("torch"); ref32 = ref32.cuda()
med = lambda a, b: float(((a - b).abs() * 1000.0).flatten().median())
for fold in ("1", "0"):
    os.environ["HIVE_LN_FOLD"] = fold
    hip = mk("hip")
    out = []
    for seed in range(8):
        torch.manual_seed(seed)
        x = (torch.rand(2, 3, 96, 128, device="cuda") * 2 - 1).to(half).float()
        xb = x.to(half).contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            d32 = ref32(x); d_hip = hip(xb); d_tor = tor(xb)
        out.append((round(med(d_hip, d32), 1), round(med(d_tor, d32), 1)))
    print("fold", fold, out)
