"""Numerics of the hand-written HIP ViT kernels (through the C ABI) against a plain PyTorch fp32
reference of the same op, evaluated on the same 16-bit-rounded inputs, for BOTH element types of the kernels:
bfloat16 (north_star's contract) and float16 (what the reference runs: ``model.half()``,
/root/reference/hive/dataset_adaptors.py:1394-1401).

Tolerance (stated, floating point): accumulation is f32; outputs are rounded once.  bfloat16 (8 significant bits,
relative step 2^-8 = 3.9e-3): |err| <= 1.5e-2 * max|ref| + 1e-2 * |ref| elementwise and a relative Frobenius error
<= 6e-3.  float16 (11 significant bits, step 2^-11 = 4.9e-4): EIGHT times tighter on all three figures."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["bfloat16", "float16"])
def half(request):
    """The 16-bit element type of the kernels under test."""
    import torch
    return getattr(torch, request.param)


def _code(dtype):
    from hive_amd import _lib
    return _lib.dtype_code(dtype)


def _close(out, ref, name):
    import torch
    tight = 0.125 if out.dtype == torch.float16 else 1.0  # float16 carries 3 more significant bits than bfloat16
    out, ref = out.float(), ref.float()
    assert torch.isfinite(out).all(), f"{name}: non-finite output"
    scale = ref.abs().max().item()
    err = (out - ref).abs()
    bound = tight * (1.5e-2 * scale + 1e-2 * ref.abs())
    assert (err <= bound).all(), f"{name}: max err {err.max().item():.4g} vs scale {scale:.4g}"
    rel = (out - ref).norm().item() / max(ref.norm().item(), 1e-30)
    assert rel <= tight * 6e-3, f"{name}: relative Frobenius error {rel:.4g}"


@pytest.mark.parametrize("M,D", [(1216, 768), (37, 768), (130, 1024), (5, 256)])
def test_layernorm(gpu_ctx, half, M, D):
    import torch
    torch.manual_seed(0)
    x = (torch.randn(M, D, device="cuda") * 3 + 0.5).to(half)
    g = torch.randn(D, device="cuda") * 0.5 + 1
    b = torch.randn(D, device="cuda") * 0.1
    out = torch.empty_like(x)
    gpu_ctx.check(gpu_ctx.lib.hive_vit_layernorm(gpu_ctx.handle, x.data_ptr(), _code(half), g.data_ptr(), b.data_ptr(), out.data_ptr(), M, D, 1e-6))
    ref = torch.nn.functional.layer_norm(x.float(), (D,), g, b, 1e-6)
    _close(out, ref, "layernorm")


@pytest.mark.parametrize("M,N,K,epi", [(1216, 768, 768, 0), (1216, 3072, 768, 1), (1216, 768, 3072, 2), (200, 128, 64, 0),
                                        (129, 256, 128, 2), (2432, 2304, 768, 0),
                                        # wide N with >= 256 tiles of 256 x 256: the 8-wave 256-tile kernel, incl. a ragged last row tile
                                        (5632, 3072, 768, 1), (7300, 2304, 768, 2), (5600, 3072, 128, 0),
                                        # enough 256 x 256 tiles to fill >= 60 % of the CU slots: the persistent 256-tile kernel, several tiles per
                                        # workgroup, ragged last row tiles
                                        (14000, 768, 3072, 2), (30000, 1536, 768, 0), (20001, 3072, 768, 1)])
def test_linear_epilogues(gpu_ctx, half, M, N, K, epi):
    import torch
    torch.manual_seed(1)
    A = torch.randn(M, K, device="cuda").to(half)
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(half)
    bias = torch.randn(N, device="cuda") * 0.1
    res = torch.randn(M, N, device="cuda").to(half)
    C = torch.empty(M, N, device="cuda", dtype=half)
    gpu_ctx.check(gpu_ctx.lib.hive_vit_linear(gpu_ctx.handle, A.data_ptr(), _code(half), W.data_ptr(), bias.data_ptr(), res.data_ptr() if epi == 2 else None,
                                              C.data_ptr(), M, N, K, epi))
    ref = A.float() @ W.float().t() + bias
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    if epi == 2:
        ref = ref + res.float()
    _close(C, ref, f"linear epi={epi}")
    if epi == 2:  # the residual stream is updated in place in the model (x += proj(...)): same result
        inplace = res.clone()
        gpu_ctx.check(gpu_ctx.lib.hive_vit_linear(gpu_ctx.handle, A.data_ptr(), _code(half), W.data_ptr(), bias.data_ptr(), inplace.data_ptr(), inplace.data_ptr(), M, N, K, epi))
        assert torch.equal(inplace, C)


@pytest.mark.parametrize("M,N,K,epi", [(1216, 768, 3072, 2), (1216, 3072, 768, 1), (769, 1536, 768, 0), (300, 768, 704, 2)])
def test_linear_split_k(gpu_ctx, half, monkeypatch, M, N, K, epi):
    """Small M (the reference's literal loop is batch 1): the K-steps of a tile are dealt to several workgroups, the last one to arrive adds the f32
    partials in a fixed order and runs the epilogue (csrc/mfma_pipe.hpp splitk_combine).  HIVE_SPLITK = 0 (off), unset (the launch policy), 5 and 7
    (ways that do not divide the 11 / 12 / 48 K-steps): all within the float32 reference's tolerance, each bit-identical from run to run (the arrival
    order does not enter the sum; the counters are back at zero for the next launch), and within one 16-bit step of the unsplit result."""
    import torch
    torch.manual_seed(3)
    A = torch.randn(M, K, device="cuda").to(half)
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(half)
    bias = torch.randn(N, device="cuda") * 0.1
    res = torch.randn(M, N, device="cuda").to(half)
    ref = A.float() @ W.float().t() + bias
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    if epi == 2:
        ref = ref + res.float()
    out = {}
    for ways in ("0", None, "5", "7"):
        if ways is None:
            monkeypatch.delenv("HIVE_SPLITK", raising=False)
        else:
            monkeypatch.setenv("HIVE_SPLITK", ways)
        runs = []
        for _ in range(3):
            C = torch.empty(M, N, device="cuda", dtype=half)
            gpu_ctx.check(gpu_ctx.lib.hive_vit_linear(gpu_ctx.handle, A.data_ptr(), _code(half), W.data_ptr(), bias.data_ptr(), res.data_ptr() if epi == 2 else None,
                                                      C.data_ptr(), M, N, K, epi))
            runs.append(C)
        assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2]), f"split {ways}: not reproducible"
        # the same workspace addresses carry another input's partials in between: a stale partial (a line a CU or an XCD still holds from the launch before)
        # would show as the other input's numbers
        A2, C2 = (A.float() * 0.5 + 1.0).to(half), torch.empty(M, N, device="cuda", dtype=half)
        for _ in range(6):
            gpu_ctx.check(gpu_ctx.lib.hive_vit_linear(gpu_ctx.handle, A2.data_ptr(), _code(half), W.data_ptr(), bias.data_ptr(), res.data_ptr() if epi == 2 else None,
                                                      C2.data_ptr(), M, N, K, epi))
            C = torch.empty(M, N, device="cuda", dtype=half)
            gpu_ctx.check(gpu_ctx.lib.hive_vit_linear(gpu_ctx.handle, A.data_ptr(), _code(half), W.data_ptr(), bias.data_ptr(), res.data_ptr() if epi == 2 else None,
                                                      C.data_ptr(), M, N, K, epi))
            assert torch.equal(C, runs[0]), f"split {ways}: a launch saw another launch's partials"
        _close(runs[0], ref, f"linear epi={epi} split={ways}")
        out[ways] = runs[0].float()
    step = 2.0 ** (-7 if half == torch.bfloat16 else -10)
    for ways in (None, "5", "7"):
        d = (out[ways] - out["0"]).abs()
        assert bool((d <= step * out["0"].abs().clamp_min(2.0 ** -6) * 1.01).all()), f"split {ways}: more than one step from the unsplit result"


def test_linear_asymmetric_identity(gpu_ctx, half):
    """A = I with an asymmetric W catches a transposed C write (a symmetric operand would hide it)."""
    import torch
    M = N = K = 128
    A = torch.eye(M, device="cuda").to(half)
    W = (torch.arange(N * K, device="cuda").reshape(N, K) % 251).float().to(half)  # integers < 256: exact in both types
    bias = torch.zeros(N, device="cuda")
    C = torch.empty(M, N, device="cuda", dtype=half)
    gpu_ctx.check(gpu_ctx.lib.hive_vit_linear(gpu_ctx.handle, A.data_ptr(), _code(half), W.data_ptr(), bias.data_ptr(), None, C.data_ptr(), M, N, K, 0))
    assert torch.equal(C.float(), W.float().t())


def test_linear_rejects_bad_shapes(gpu_ctx, half):
    from hive_amd._lib import HiveError
    import torch
    t = torch.zeros(128 * 128, device="cuda", dtype=half)
    f = torch.zeros(128, device="cuda")
    with pytest.raises(HiveError):
        gpu_ctx.check(gpu_ctx.lib.hive_vit_linear(gpu_ctx.handle, t.data_ptr(), _code(half), t.data_ptr(), f.data_ptr(), None, t.data_ptr(), 128, 100, 64, 0))
    with pytest.raises(HiveError):
        gpu_ctx.check(gpu_ctx.lib.hive_vit_linear(gpu_ctx.handle, t.data_ptr(), _code(half), t.data_ptr(), f.data_ptr(), None, t.data_ptr(), 128, 128, 64, 2))


@pytest.mark.parametrize("B,N,D,H", [(1, 1201, 768, 12), (2, 77, 768, 12), (1, 64, 768, 12), (3, 130, 768, 12), (24, 1201, 768, 12), (2, 500, 768, 12), (1, 300, 768, 12),
                                     (1, 577, 1024, 16)])
# 24 x 1216 rows: q|k and v^T on the persistent 256-tile kernel; 1 x 1201, 2 x 500, 1 x 300: grids of at most one workgroup per CU -- the keys split two ways inside the
# workgroup (19 = 10 + 9, 8 = 4 + 4, 5 = 3 + 2 tiles); 1 x 577 at D = 1024, 16 heads: DPT-Large's one-frame shapes (384 x 384) on the merged q | k | v^T launch and the key split
def test_qkv_and_attention(gpu_ctx, half, B, N, D, H, monkeypatch):
    """qkv projection (q|k row-major, v transposed) + softmax(q k^T / 8) v, incl. ragged N (key masking),
    and a spiked key row that forces the online-softmax rescale branch.  At small batches q | k and v^T are ONE launch (vit.hip qkv_t, round 5): the
    two-launch orchestration (HIVE_QKV_MERGE=0) and the two-workgroups-per-CU merge ("2") must give the same bits."""
    import torch
    torch.manual_seed(2)
    Np = (N + 63) // 64 * 64
    x = torch.zeros(B, Np, D, device="cuda")
    x[:, :N] = torch.randn(B, N, D, device="cuda")
    x = x.to(half)
    W = (torch.randn(3 * D, D, device="cuda") / D ** 0.5).to(half)
    # spike: make one late key dominate for every query of head 0 (running max jumps mid-sequence)
    W[D:D + 64] *= 1.0
    x[:, N - 3] *= 6.0
    bias = torch.randn(3 * D, device="cuda") * 0.1
    qk = torch.empty(B * Np, 2 * D, device="cuda", dtype=half)
    vT = torch.empty(B, H, 64, Np, device="cuda", dtype=half)
    gpu_ctx.check(gpu_ctx.lib.hive_vit_qkv(gpu_ctx.handle, x.data_ptr(), _code(half), W.data_ptr(), bias.data_ptr(), qk.data_ptr(), vT.data_ptr(), B, Np, D, H))
    ref_qkv = x.float().reshape(B * Np, D) @ W.float().t() + bias
    c = 0.125 * 1.4426950408889634  # q is stored in the softmax's base-2 exponent units: q * head_dim^-0.5 * log2(e)
    ref_qk = ref_qkv[:, :2 * D].clone()
    ref_qk[:, :D] *= c
    _close(qk, ref_qk, "q|k")
    ref_v = ref_qkv[:, 2 * D:].reshape(B, Np, H, 64).permute(0, 2, 3, 1)
    # v^T is stored with token quads 4..7 and 8..11 of every 16 swapped (bits 2 and 3 of the token index exchanged): undo it
    tok = torch.arange(Np, device="cuda")
    slot = (tok & ~12) | ((tok & 4) << 1) | ((tok & 8) >> 1)
    vT_tokens = vT[..., slot]  # column t of the logical v^T lives in column slot[t]
    _close(vT_tokens, ref_v, "v^T")
    for mode in ("0", "2"):
        monkeypatch.setenv("HIVE_QKV_MERGE", mode)
        qk2, vT2 = torch.full_like(qk, 7.0), torch.full_like(vT, 7.0)
        gpu_ctx.check(gpu_ctx.lib.hive_vit_qkv(gpu_ctx.handle, x.data_ptr(), _code(half), W.data_ptr(), bias.data_ptr(), qk2.data_ptr(), vT2.data_ptr(), B, Np, D, H))
        assert torch.equal(qk2, qk) and torch.equal(vT2, vT), f"HIVE_QKV_MERGE={mode}"
    monkeypatch.delenv("HIVE_QKV_MERGE")
    out = torch.empty(B * Np, D, device="cuda", dtype=half)
    gpu_ctx.check(gpu_ctx.lib.hive_vit_attention(gpu_ctx.handle, qk.data_ptr(), _code(half), vT.data_ptr(), out.data_ptr(), B, N, Np, D, H))
    # reference on the kernel's own 16-bit q, k, v
    q = qk[:, :D].float().reshape(B, Np, H, 64).permute(0, 2, 1, 3)[:, :, :N]
    k = qk[:, D:].float().reshape(B, Np, H, 64).permute(0, 2, 1, 3)[:, :, :N]
    v = vT_tokens.float().permute(0, 1, 3, 2)[:, :, :N]
    attn = torch.softmax(q @ k.transpose(-1, -2) * 0.6931471805599453, dim=-1)  # exp2(q' k^T) with the stored q' = c q
    ref = (attn @ v).permute(0, 2, 1, 3).reshape(B, N, D)
    _close(out.reshape(B, Np, D)[:, :N], ref, "attention")
    # both forms of the kernel (keys in one chain / split two ways and merged: vit.hip attention_kernel KS) against the same reference, and against each other
    for ks in ("0", "1"):
        monkeypatch.setenv("HIVE_ATT_KSPLIT", ks)
        out2 = torch.full_like(out, 3.0)
        gpu_ctx.check(gpu_ctx.lib.hive_vit_attention(gpu_ctx.handle, qk.data_ptr(), _code(half), vT.data_ptr(), out2.data_ptr(), B, N, Np, D, H))
        _close(out2.reshape(B, Np, D)[:, :N], ref, f"attention, HIVE_ATT_KSPLIT={ks}")
        _close(out2.reshape(B, Np, D)[:, :N], out.reshape(B, Np, D)[:, :N].float(), f"attention forms, HIVE_ATT_KSPLIT={ks}")
    monkeypatch.delenv("HIVE_ATT_KSPLIT")


@pytest.mark.parametrize("N,C,H,W,relu,res", [(2, 64, 24, 32, True, False), (1, 256, 30, 40, False, True), (3, 1024, 6, 8, True, False),
                                               (2, 512, 15, 20, False, False), (1, 128, 33, 17, True, True)])
def test_group_norm_fused_matches_torch(gpu_ctx, half, N, C, H, W, relu, res):
    """GroupNorm(32) [+ residual] [+ ReLU] on channels-last bf16 vs F.group_norm in fp32 on the same values."""
    import torch
    from hive_amd.dpt import ops
    torch.manual_seed(5)
    cl = torch.channels_last
    x = (torch.randn(N, C, H, W, device="cuda") * 2 + 0.3).to(half).contiguous(memory_format=cl)
    g = (torch.randn(C, device="cuda") * 0.5 + 1).to(half)
    b = (torch.randn(C, device="cuda") * 0.2).to(half)
    r = torch.randn(N, C, H, W, device="cuda").to(half).contiguous(memory_format=cl) if res else None
    out = ops.group_norm_act(x, 32, g, b, 1e-5, relu=relu, residual=r, engine="hip")
    assert out.is_contiguous(memory_format=cl) and out.dtype == half
    ref = torch.nn.functional.group_norm(x.float(), 32, g.float(), b.float(), 1e-5)
    if res:
        ref = ref.to(half).float() + r.float()
    if relu:
        ref = torch.relu(ref)
    _close(out, ref, "group_norm")
    # and bit-for-bit deterministic
    assert torch.equal(out, ops.group_norm_act(x, 32, g, b, 1e-5, relu=relu, residual=r, engine="hip"))


@pytest.mark.parametrize("N,C,H,W", [(2, 256, 15, 20), (1, 128, 24, 32), (1, 8, 1, 1), (2, 32, 7, 5), (3, 256, 30, 40), (1, 64, 9, 13), (1, 1024, 6, 5)])
def test_upsample2x_matches_torch(gpu_ctx, half, N, C, H, W):
    import torch
    from hive_amd.dpt import ops
    torch.manual_seed(6)
    x = torch.randn(N, C, H, W, device="cuda").to(half).contiguous(memory_format=torch.channels_last)
    out = ops.upsample2x(x, engine="hip")
    ref = torch.nn.functional.interpolate(x.float(), scale_factor=2, mode="bilinear", align_corners=True)
    # bias folded into the load == bias added (and rounded to bf16) first, bit for bit
    b = torch.randn(C, device="cuda").to(half)
    xb = (x + b.view(1, -1, 1, 1)).contiguous(memory_format=torch.channels_last)
    assert torch.equal(ops.upsample2x(x, engine="hip", bias=b), ops.upsample2x(xb, engine="hip"))
    assert out.shape == ref.shape and out.is_contiguous(memory_format=torch.channels_last)
    # the LDS-tiled kernel (C <= 512) and the gather kernel evaluate the same formula on the same operands: bit-identical
    import os
    os.environ["HIVE_UPSAMPLE_GATHER"] = "1"
    try:
        assert torch.equal(ops.upsample2x(x, engine="hip"), out) and torch.equal(ops.upsample2x(x, engine="hip", bias=b), ops.upsample2x(xb, engine="hip"))
    finally:
        del os.environ["HIVE_UPSAMPLE_GATHER"]
    # same formula evaluated in float: only the final rounding (half an ulp: 2^-9 relative for bfloat16, 2^-12 for float16)
    # differs from the fp32 reference
    ulp = 2.0 ** -8 if half == torch.bfloat16 else 2.0 ** -11
    assert (out.float() - ref).abs().max().item() <= ulp * ref.abs().max().item() + 1e-6
    ref16 = torch.nn.functional.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    assert (out.float() - ref16.float()).abs().max().item() <= 2 * ulp * ref.abs().max().item() + 1e-6


def test_vit_forward_matches_fp32_blocks(gpu_ctx, half):
    """All 12 blocks through hive_vit_forward vs the fp32 torch blocks with the same (bf16-rounded) weights."""
    import torch
    from hive_amd.dpt.models import VisionTransformerHybrid
    from hive_amd.dpt.vit_engine import VitEngine
    torch.manual_seed(3)
    vit = VisionTransformerHybrid().eval()
    for p in vit.blocks.parameters():  # round the weights once so that both paths see identical values
        p.data = p.data.bfloat16().float()
    vit = vit.cuda()
    import copy
    from hive_amd._lib import HiveError
    with pytest.raises(HiveError):  # a float32 model is not silently down-cast
        VitEngine(vit, ctx=gpu_ctx)
    vit16 = copy.deepcopy(vit).to(half)
    eng = VitEngine(vit16, ctx=gpu_ctx)
    B, N = 2, 301
    tokens = torch.randn(B, N, 768, device="cuda").to(half)
    t8, t11 = eng.forward(tokens, taps=(8, 11))
    with torch.no_grad():
        x = tokens.float()
        refs = {}
        for i, blk in enumerate(vit.blocks):
            x = blk(x)
            refs[i] = x
    # 12 blocks of 16-bit activations: the error accumulates along the residual stream (float16: 8 x tighter; its weights were
    # rounded to bfloat16 values above, which float16 holds exactly)
    for out, ref, name in ((t8, refs[8], "block 8"), (t11, refs[11], "block 11")):
        rel = (out.float() - ref).norm().item() / ref.norm().item()
        assert torch.isfinite(out).all() and rel < (2e-2 if half == torch.bfloat16 else 2.5e-3), f"{name}: relative error {rel:.4g}"
    assert t8.shape == (B, N, 768) and t8.dtype == half


def test_vit_forward_layernorm_folded_into_the_gemms(gpu_ctx, half, monkeypatch):
    """hive_vit_forward folds each LayerNorm into the GEMM that consumes it (x (gamma o W)^T with the rows' (mean, rstd) applied in the epilogue,
    the statistics left by the proj / fc2 epilogues): against the same forward with a LayerNorm pass (HIVE_LN_FOLD=0) and against the fp32 blocks,
    with LayerNorm gains / biases away from 1 / 0 and tokens whose mean is several times their spread (what the fold's mean c1 term has to cancel)."""
    import copy
    import torch
    from hive_amd.dpt.models import VisionTransformerHybrid
    from hive_amd.dpt.vit_engine import VitEngine
    torch.manual_seed(11)
    vit = VisionTransformerHybrid().eval()
    with torch.no_grad():
        for blk in vit.blocks:
            for norm in (blk.norm1, blk.norm2):
                norm.weight.copy_(1.0 + 0.5 * torch.randn_like(norm.weight))
                norm.bias.copy_(0.3 * torch.randn_like(norm.bias))
    for p in vit.blocks.parameters():
        p.data = p.data.bfloat16().float()
    vit = vit.cuda()
    B, N = 2, 301
    tokens = (torch.randn(B, N, 768, device="cuda") + 3.0 * torch.randn(B, N, 1, device="cuda")).to(half)  # per-token mean offsets of ~3 sigma
    with torch.no_grad():
        x = tokens.float()
        refs = {}
        for i, blk in enumerate(vit.blocks):
            x = blk(x)
            refs[i] = x
    outs = {}
    for fold in ("1", "0"):
        monkeypatch.setenv("HIVE_LN_FOLD", fold)
        eng = VitEngine(copy.deepcopy(vit).to(half), ctx=gpu_ctx)
        outs[fold] = eng.forward(tokens, taps=(0, 8, 11))
        del eng
    tol = 2e-2 if half == torch.bfloat16 else 2.5e-3
    for k, blk in enumerate((0, 8, 11)):
        ref = refs[blk]
        rel = {f: (outs[f][k].float() - ref).norm().item() / ref.norm().item() for f in outs}
        assert torch.isfinite(outs["1"][k]).all() and rel["1"] < tol, f"block {blk}: folded {rel['1']:.4g}"
        assert rel["1"] <= 1.25 * rel["0"] + 1e-4, f"block {blk}: folded {rel['1']:.4g} vs LayerNorm pass {rel['0']:.4g}"
    # padded batch sizes / several batches reuse the statistics buffers: same result for the same tokens
    monkeypatch.setenv("HIVE_LN_FOLD", "1")
    eng = VitEngine(copy.deepcopy(vit).to(half), ctx=gpu_ctx)
    a = eng.forward(tokens, taps=(11,))[0]
    eng.forward(torch.cat([tokens, tokens]), taps=(11,))
    b = eng.forward(tokens, taps=(11,))[0]
    assert torch.equal(a, b)


def test_dpt_hip_engine_matches_torch_engine(gpu_ctx, half):
    """Whole DPT-Hybrid at a small size (96 x 128), seeded non-degenerate weights (tests/dpt_weights.py): engine='hip' against
    the float32 model and against PyTorch's own bf16 operators, in millimetres of depth; and the device hand-off arithmetic
    of the reference (uint16 mm truncation, metres, > max_depth -> 0) bit for bit on the depth the engine produced.
    (tests/test_dpt_gpu.py holds the per-stage comparison at the benchmark's 480 x 640.)"""
    import torch
    from dpt_weights import seeded_init
    from hive_amd.dpt.models import DPTDepthModel
    torch.manual_seed(2)  # (the input below is random: at this size -- 13 tokens -- the median error of ANY bf16 engine moves by +-4 mm with it; tools/diag_small_dpt_seeds.py)
    ref32 = DPTDepthModel(path=None, scale=0.000305, shift=0.1378, invert=True, engine="torch").eval()
    seeded_init(ref32, seed=4)
    hip = DPTDepthModel(path=None, scale=0.000305, shift=0.1378, invert=True, engine="hip").eval()
    hip.load_state_dict(ref32.state_dict())
    hip = hip.to(memory_format=torch.channels_last).to(half).cuda()
    tor = DPTDepthModel(path=None, scale=0.000305, shift=0.1378, invert=True, engine="torch").eval()
    tor.load_state_dict(ref32.state_dict())
    tor = tor.to(memory_format=torch.channels_last).to(half).cuda()
    ref32 = ref32.cuda()
    x = (torch.rand(2, 3, 96, 128, device="cuda") * 2 - 1).to(half).float()
    xb = x.to(half).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        d32 = ref32(x)
        d_hip, mm, m = hip(xb, handoff=(10.0,))
        d_tor, mm_t, m_t = tor(xb, handoff=(10.0,))
    assert d_hip.shape == (2, 96, 128) and d_hip.dtype == torch.float32 and torch.isfinite(d_hip).all()
    assert float(d32.max() - d32.min()) > 2.0, "seeded weights must give a depth range of metres"
    med = lambda a, b: float(((a - b).abs() * 1000.0).flatten().median())
    e_hip, e_tor = med(d_hip, d32), med(d_tor, d32)
    limit = 25.0 if half == torch.bfloat16 else 4.0  # float16: 3 more significant bits
    assert e_hip <= limit, f"HIP engine ({half}) vs float32: median {e_hip:.1f} mm"
    assert e_hip <= 1.3 * e_tor + limit / 12, f"HIP engine ({e_hip:.1f} mm) must not be less accurate than PyTorch's own {half} operators ({e_tor:.1f} mm)"
    # device hand-off == reference arithmetic on the same f32 depth: trunc(depth * 1000) -> uint16 -> / 1000 -> > 10 -> 0
    exp_mm = (d_hip * 1000.0).to(torch.int32)
    assert torch.equal(mm.to(torch.int32) & 0xFFFF, exp_mm)
    exp_m = exp_mm.float() * (1.0 / 1000.0)
    assert torch.equal(m, torch.where(exp_m > 10.0, torch.zeros_like(exp_m), exp_m))


@pytest.mark.parametrize("N,H,W", [(1, 8, 16), (2, 13, 21), (1, 48, 64)])
def test_fused_head_matches_torch(gpu_ctx, half, N, H, W):
    """hive_dpt_head_fused (Interpolate x2 -> conv3x3 128->32 -> ReLU -> conv1x1 -> ReLU -> inversion -> hand-off) vs the
    same operators evaluated in float32 by PyTorch on the bf16 upsampled map of the unfused path.  Tolerance: f32
    accumulation order only (relative 2e-4 on the pre-inversion value)."""
    import torch
    import torch.nn.functional as F
    from hive_amd import _lib
    torch.manual_seed(N * 100 + H)
    x = (torch.randn(N, 128, H, W, device="cuda") * 0.5).to(half).contiguous(memory_format=torch.channels_last)
    w3 = (torch.randn(32, 128, 3, 3, device="cuda") * 0.05).to(half)
    b3 = torch.randn(32) * 0.1
    w1 = torch.randn(32) * 0.3
    b1 = 0.05
    scale, shift = 0.01, 0.1
    # the stand-alone HIP upsampling kernel (itself tested against F.interpolate to one bf16 ulp) gives the bf16
    # map that the fused kernel builds tile by tile: same expression, same rounding
    from hive_amd.dpt import ops as dpt_ops
    b0 = torch.randn(128, device="cuda") * 0.2  # bias of the producing convolution, folded into the kernel's load
    xb = (x.float() + b0.reshape(1, 128, 1, 1)).to(half).contiguous(memory_format=torch.channels_last)
    up = dpt_ops.upsample2x(xb, engine="hip").float()
    feat = F.relu(F.conv2d(up, w3.float(), b3.cuda(), padding=1))
    pre = F.relu(F.conv2d(feat, w1.cuda().reshape(1, 32, 1, 1), torch.tensor([b1], device="cuda"))).squeeze(1)
    ref = 1.0 / torch.clamp(scale * pre + shift, min=1e-8)
    depth = torch.empty((N, 2 * H, 2 * W), dtype=torch.float32, device="cuda")
    mm = torch.empty((N, 2 * H, 2 * W), dtype=torch.int16, device="cuda")
    m = torch.empty((N, 2 * H, 2 * W), dtype=torch.float32, device="cuda")
    w3_dev = w3.permute(2, 3, 0, 1).contiguous()
    b3_np, w1_np = b3.numpy().astype("float32"), w1.numpy().astype("float32")
    ctx = gpu_ctx
    ctx.check(ctx.lib.hive_dpt_head_fused(ctx.handle, x.data_ptr(), b0.data_ptr(), _code(half), N, H, W, 128, 32, w3_dev.data_ptr(), b3_np.ctypes.data,
                                          w1_np.ctypes.data, b1, 1, 1, scale, shift, depth.data_ptr(), 1.0 / 1000.0, 10.0,
                                          mm.data_ptr(), m.data_ptr()))
    torch.cuda.synchronize()
    got_pre = (1.0 / depth - shift) / scale
    err = (got_pre - pre).abs().max().item() / (pre.abs().max().item() + 1e-6)
    assert err < 2e-4, f"pre-inversion relative error {err:.3g}"  # f32 accumulation order only, either element type
    assert torch.allclose(depth, ref, rtol=1e-4, atol=1e-6)
    exp_mm = (depth * 1000.0).clamp(0, 65535).to(torch.int32)
    assert torch.equal(mm.to(torch.int32) & 0xFFFF, exp_mm)
    exp_m = exp_mm.float() * (1.0 / 1000.0)
    assert torch.equal(m, torch.where(exp_m > 10.0, torch.zeros_like(exp_m), exp_m))
