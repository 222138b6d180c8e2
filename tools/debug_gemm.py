import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_amd import _lib
DT = int(__import__("os").environ.get("HIVE_PROBE_DTYPE", "2"))  # hive_dtype of the operands: 2 = bf16 (default), 1 = f16
ctx = _lib.default_context(0)
M = N = K = 128
A = torch.eye(M, device="cuda").bfloat16()
W = (torch.arange(N * K, device="cuda").reshape(N, K) % 251).float().bfloat16()
bias = torch.zeros(N, device="cuda")
for it in range(6):
    C = torch.full((M, N), -7.0, device="cuda", dtype=torch.bfloat16)
    ctx.check(ctx.lib.hive_vit_linear(ctx.handle, A.data_ptr(), DT, W.data_ptr(), bias.data_ptr(), None, C.data_ptr(), M, N, K, 0))
    torch.cuda.synchronize()
    ref = W.float().t()
    bad = (C.float() != ref).nonzero()
    print("iter", it, "mismatches", len(bad), bad[:12].tolist())
    if len(bad):
        m, n = bad[0].tolist()
        print("  got", C[m, n].item(), "expected", ref[m, n].item(), "row got", C[m, :16].float().tolist())
# random check too
A2 = torch.randn(256, 128, device="cuda").bfloat16(); W2 = torch.randn(128, 128, device="cuda").bfloat16()
C2 = torch.empty(256, 128, device="cuda", dtype=torch.bfloat16)
ctx.check(ctx.lib.hive_vit_linear(ctx.handle, A2.data_ptr(), DT, W2.data_ptr(), bias.data_ptr(), None, C2.data_ptr(), 256, 128, 128, 0))
print("random max err", (C2.float() - A2.float() @ W2.float().t()).abs().max().item())
for (M2, N2, K2) in [(256, 128, 128), (128, 256, 128), (128, 128, 64), (128, 128, 192), (384, 384, 256)]:
    A2 = torch.randn(M2, K2, device="cuda").bfloat16(); W2 = torch.randn(N2, K2, device="cuda").bfloat16()
    b2 = torch.zeros(N2, device="cuda")
    for it in range(3):
        C2 = torch.full((M2, N2), -7.0, device="cuda", dtype=torch.bfloat16)
        ctx.check(ctx.lib.hive_vit_linear(ctx.handle, A2.data_ptr(), DT, W2.data_ptr(), b2.data_ptr(), None, C2.data_ptr(), M2, N2, K2, 0))
        torch.cuda.synchronize()
        ref = A2.float() @ W2.float().t()
        err = (C2.float() - ref).abs()
        bad = (~(err < 0.02 * ref.abs().max())).nonzero()
        print((M2, N2, K2), "it", it, "bad", len(bad), "nan", int(torch.isnan(C2.float()).sum()), "sentinel", int((C2.float() == -7).sum()),
              "first bad", bad[:6].tolist())
