"""The hand-written implicit-GEMM convolutions (csrc/conv.hip, csrc/stem.hip, through the C ABI) against torch's float32 conv2d on
the same 16-bit-exact operands, for both element types of the kernels: bfloat16 and float16 (the reference's ``model.half()``,
/root/reference/hive/dataset_adaptors.py:1394-1401).  Tolerance: float32 accumulation on both sides, one rounding of the result
(half an ulp: 2^-9 relative for bfloat16, 2^-12 for float16) plus accumulation-order noise over K = k k C_in products; every
bound below is stated in ulps of the element type, i.e. eight times tighter for float16."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["bfloat16", "float16"])
def half(request):
    """The 16-bit element type of the kernels under test."""
    return getattr(torch, request.param)


def _ulp(dtype):
    return 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11


def _mk(n, cin, cout, h, w, bias, seed, half):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(n, cin, h, w, generator=g).to(half)
    conv = nn.Conv2d(cin, cout, 3, 1, 1, bias=bias)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (9 * cin)) ** 0.5)
        if bias:
            conv.bias.copy_(torch.randn(cout, generator=g) * 0.3)
    conv = conv.to(memory_format=torch.channels_last).to(half).cuda()
    return x.cuda().contiguous(memory_format=torch.channels_last), conv


def _check(out, ref, what):
    assert out.shape == ref.shape and out.dtype in (torch.bfloat16, torch.float16) and out.is_contiguous(memory_format=torch.channels_last)
    u = _ulp(out.dtype)
    err = (out.float() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 2 * u * scale + 0.25 * u, f"{what}: max error {err:.4g} vs scale {scale:.4g}"
    rel = ((out.float() - ref).norm() / ref.norm()).item()
    assert rel < 0.77 * u, f"{what}: relative Frobenius error {rel:.4g}"


@pytest.mark.parametrize("n,cin,cout,h,w", [
    (2, 768, 256, 15, 20),    # layer4_rn at 480 x 640
    (1, 512, 256, 60, 80),    # layer2_rn
    (3, 256, 256, 7, 9),      # M = 189: one partial tile; every pixel near a border
    (1, 256, 256, 33, 41),    # M = 1353: tiles end mid-row
    (2, 256, 128, 24, 32),    # output_conv[0]: 128 output channels (the 64 x 64 per wave tiling)
    (1, 64, 128, 16, 16),     # one K-step per tap
])
def test_conv3x3_plain(gpu_ctx, half, n, cin, cout, h, w):
    from hive_amd.dpt import ops
    x, conv = _mk(n, cin, cout, h, w, bias=False, seed=cin + h, half=half)
    assert ops.conv3x3_eligible(x, conv)
    out = ops.conv3x3(x, conv)
    ref = F.conv2d(x.float(), conv.weight.float(), None, 1, 1)
    _check(out, ref, "conv")


def test_conv3x3_fused_epilogue(gpu_ctx, half):
    """bias + two skip connections + ReLU variants: out = relu?(conv + b + r1 + r2), out_relu = relu(out)."""
    from hive_amd.dpt import ops
    x, conv = _mk(2, 256, 256, 30, 40, bias=True, seed=5, half=half)
    g = torch.Generator(device="cpu").manual_seed(9)
    r1 = torch.randn(2, 256, 30, 40, generator=g).to(half).cuda().contiguous(memory_format=torch.channels_last)
    r2 = torch.randn(2, 256, 30, 40, generator=g).to(half).cuda().contiguous(memory_format=torch.channels_last)
    base = F.conv2d(x.float(), conv.weight.float(), conv.bias.float(), 1, 1)
    _check(ops.conv3x3(x, conv, relu=True), F.relu(base), "bias + relu")
    _check(ops.conv3x3(x, conv, residual=r1), base + r1.float(), "bias + residual")
    out, out_relu = ops.conv3x3(x, conv, residual=r1, residual2=r2, also_relu=True)
    ref = base + r1.float() + r2.float()
    _check(out, ref, "bias + two residuals")
    _check(out_relu, F.relu(ref), "relu copy")
    assert torch.equal(out_relu, F.relu(out)), "out_relu must be relu(out) of the SAME rounded values"
    _check(ops.conv3x3(x, conv, with_bias=False), F.conv2d(x.float(), conv.weight.float(), None, 1, 1), "bias skipped")
    # reproducible: no atomics, fixed accumulation order
    assert torch.equal(ops.conv3x3(x, conv, relu=True), ops.conv3x3(x, conv, relu=True))


def test_conv3x3_full_resolution_refinenet1(gpu_ctx, half):
    """The largest decoder shape of the benchmark: 240 x 320 x 256 -> 256, two images (M = 153,600 = 600 full tiles)."""
    from hive_amd.dpt import ops
    x, conv = _mk(2, 256, 256, 240, 320, bias=True, seed=11, half=half)
    out = ops.conv3x3(x, conv, relu=True)
    ref = F.relu(F.conv2d(x.float(), conv.weight.float(), conv.bias.float(), 1, 1))
    _check(out, ref, "240 x 320")


def test_conv3x3_weights_not_channels_last_and_rejections(gpu_ctx, half):
    from hive_amd import _lib
    from hive_amd.dpt import ops
    x, conv = _mk(1, 64, 128, 8, 8, bias=True, seed=1, half=half)
    conv_nchw = nn.Conv2d(64, 128, 3, 1, 1).to(half).cuda()  # default (contiguous) weight layout
    with torch.no_grad():
        conv_nchw.weight.copy_(conv.weight)
        conv_nchw.bias.copy_(conv.bias)
    assert not conv_nchw.weight.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(ops.conv3x3(x, conv_nchw), ops.conv3x3(x, conv))
    assert not ops.conv3x3_eligible(x, nn.Conv2d(64, 128, 3, 2, 1).to(half).cuda())   # stride 2
    assert not ops.conv3x3_eligible(x, nn.Conv2d(64, 96, 3, 1, 1).to(half).cuda())    # C_out % 128
    assert not ops.conv3x3_eligible(x.float(), conv)
    ctx = gpu_ctx
    out = torch.empty_like(x)
    rc = ctx.lib.hive_nhwc_conv3x3(ctx.handle, x.data_ptr(), _lib.dtype_code(half), 1, 8, 8, 64, 96, conv.weight.data_ptr(), None, 0, None, None, out.data_ptr(), None)
    assert rc == _lib.ERR_INVALID
    rc = ctx.lib.hive_nhwc_conv3x3(ctx.handle, x.data_ptr(), _lib.dtype_code(half), 1, 8, 8, 64, 128, conv.weight.data_ptr(), None, 0, None, None, x.data_ptr(), None)
    assert rc == _lib.ERR_INVALID, "in-place convolution must be refused"


@pytest.mark.parametrize("cin,cout,k,stride,h,w,same", [
    (64, 64, 1, 1, 24, 32, True),      # stage-0 conv1: 64-wide output tile
    (64, 64, 3, 1, 24, 32, True),      # stage-0 conv2
    (64, 256, 1, 1, 24, 32, True),     # stage-0 conv3 / downsample
    (256, 128, 1, 1, 24, 32, True),    # stage-1 block-0 conv1
    (128, 128, 3, 2, 24, 32, True),    # stage-1 block-0 conv2: stride 2, SAME padding = 0 top / left, 1 bottom / right
    (128, 128, 3, 2, 23, 31, True),    # odd input: SAME padding 1 / 1
    (256, 512, 1, 2, 24, 32, True),    # stage-1 downsample: 1 x 1 stride 2
    (1024, 256, 1, 1, 15, 20, True),   # stage-2 conv1
    (256, 1024, 1, 1, 15, 20, True),   # stage-2 conv3
    (1024, 768, 1, 1, 15, 20, False),  # patch_embed.proj (with bias)
    (768, 768, 3, 2, 15, 20, False),   # act_postprocess4[4]: 3 x 3 stride 2 padding 1
    (256, 256, 1, 1, 30, 40, False),   # FeatureFusionBlock.out_conv
])
def test_general_conv_matches_torch(gpu_ctx, half, cin, cout, k, stride, h, w, same):
    """hive_nhwc_conv on every convolution shape family of the hybrid backbone, against float32 torch with the same padding
    rule (timm StdConv2dSame's TensorFlow 'SAME' padding or nn.Conv2d's symmetric one)."""
    from hive_amd.dpt import ops
    from hive_amd.dpt.models import StdConv2dSame
    g = torch.Generator(device="cpu").manual_seed(cin * 7 + k * 3 + stride + h)
    x = torch.randn(2, cin, h, w, generator=g).to(half).cuda().contiguous(memory_format=torch.channels_last)
    if same:
        conv = StdConv2dSame(cin, cout, k, stride=stride)
    else:
        conv = nn.Conv2d(cin, cout, k, stride, 1 if k == 3 else 0, bias=True)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (k * k * cin)) ** 0.5)
        if conv.bias is not None:
            conv.bias.copy_(torch.randn(cout, generator=g) * 0.3)
    conv = conv.to(memory_format=torch.channels_last).to(half).cuda().eval()
    assert ops.conv_eligible(x, conv)
    if same:
        conv.engine = "torch"
        with torch.no_grad():
            wt = conv.standardized_weight()
            ih, iw = h, w
            oh, ow = -(-ih // stride), -(-iw // stride)
            ph, pw = max((oh - 1) * stride + k - ih, 0), max((ow - 1) * stride + k - iw, 0)
            xp = F.pad(x.float(), (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
            ref = F.conv2d(xp, wt.float(), None, stride, 0)
            conv.engine = "hip"
            out = conv(x)  # StdConv2dSame.forward -> the HIP kernel
    else:
        with torch.no_grad():
            ref = F.conv2d(x.float(), conv.weight.float(), conv.bias.float(), stride, conv.padding)
            out = ops.conv2d(x, conv)
    _check(out, ref, f"{cin}->{cout} k{k} s{stride}")


@pytest.mark.parametrize("n,h,w", [(2, 480, 640), (1, 96, 128), (1, 61, 77)])
def test_stem_conv_and_maxpool_match_torch(gpu_ctx, half, n, h, w):
    """ResNetV2 stem on the HIP engine (csrc/stem.hip): the 7 x 7 / 2 weight-standardised "SAME" convolution from the 3-channel
    channels-last frame and MaxPool2dSame(3, 2), against the PyTorch formulation of the same modules in float32."""
    from hive_amd.dpt.models import MaxPool2dSame, StdConv2dSame
    g = torch.Generator(device="cpu").manual_seed(h)
    x = (torch.rand(n, 3, h, w, generator=g) * 2 - 1).to(half).cuda().contiguous(memory_format=torch.channels_last)
    conv = StdConv2dSame(3, 64, 7, stride=2)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g))
    conv = conv.to(memory_format=torch.channels_last).to(half).cuda().eval()
    with torch.no_grad():
        conv.engine = "torch"
        wt = conv.standardized_weight().float()
        oh, ow = (h + 1) // 2, (w + 1) // 2
        ph, pw = max((oh - 1) * 2 + 7 - h, 0), max((ow - 1) * 2 + 7 - w, 0)
        ref = F.conv2d(F.pad(x.float(), (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2)), wt, None, 2, 0)
        conv.engine = "hip"
        out = conv(x)
    _check(out, ref, "stem 7x7/2")
    pool = MaxPool2dSame(3, 2)
    y = out  # bf16 channels-last [n, 64, oh, ow]
    pool.engine = "torch"
    ref_p = pool(y.float())
    pool.engine = "hip"
    got_p = pool(y)
    assert got_p.shape == ref_p.shape and got_p.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got_p.float(), ref_p), "max pooling is exact"
    # the stem as the network runs it: the GroupNorm's sums out of the convolution's epilogue (whole 8 x 32 tiles only), and
    # GroupNorm + ReLU + max pool as ONE pass -- bit-identical to the three kernels in sequence on the same statistics
    from hive_amd.dpt import ops
    from hive_amd.dpt.models import GroupNormAct
    norm = GroupNormAct(64)
    with torch.no_grad():
        norm.weight.copy_(torch.rand(64, generator=g) + 0.5)
        norm.bias.copy_(torch.randn(64, generator=g) * 0.2)
    norm = norm.to(half).cuda().eval()
    norm.engine = "hip"
    stats = getattr(out, "hive_gn_stats", None)
    assert (stats is not None) == (oh % 8 == 0 and ow % 32 == 0), "sums from the epilogue exactly when the map is whole tiles"
    if stats is not None:
        partial, tile_rows = stats
        run = tile_rows // 256  # 8 x 32 tiles summed per row of the partials (consecutive tiles of one image)
        assert tile_rows == 256 * run and run in (1, 2, 4) and ((oh // 8) * (ow // 32)) % run == 0
        rows = out.permute(0, 2, 3, 1).float()  # [n][oh][ow][64]
        n_runs = (oh // 8) * (ow // 32) // run
        got = partial[: n * n_runs * 4 * 64].view(n, n_runs, 2, 2, 64).double()
        tiles = rows.double().view(n, oh // 8, 8, ow // 32, 32, 64).permute(0, 1, 3, 2, 4, 5).reshape(n, n_runs, run * 256, 64)  # tiles in row-major order
        assert torch.allclose(got[:, :, 0, 0], tiles.sum(2), rtol=1e-5, atol=1e-2) and torch.allclose(got[:, :, 0, 1], (tiles ** 2).sum(2), rtol=1e-5, atol=1e-2)
        assert got[:, :, 1].abs().max().item() == 0.0
    with torch.no_grad():
        for st in ({None, None} if stats is None else (stats, None)):
            seq = pool(ops.group_norm_act(out, 32, norm.weight, norm.bias, norm.eps, relu=True, engine="hip", stats=st))
            one = ops.group_norm_relu_maxpool(out, norm, stats=st)
            assert torch.equal(one, seq), "fused GroupNorm + ReLU + max pool vs the separate kernels"
        ref_n = pool.__class__(3, 2)(F.relu(F.group_norm(out.float(), 32, norm.weight.float(), norm.bias.float(), norm.eps)))
    assert (one.float() - ref_n).abs().max().item() <= 4 * _ulp(half) * max(ref_n.abs().max().item(), 1.0)


@pytest.mark.parametrize("n,cin,cout,h,w", [
    (1, 256, 256, 24, 32),    # 6 x 2 tiles of 128 x 128, 36 K-steps: split four ways by the launch policy
    (1, 256, 256, 15, 20),    # M = 300: a ragged last tile
    (2, 512, 128, 30, 40),    # 72 K-steps, one column of tiles
    (1, 64, 256, 60, 80),     # 9 K-steps: the ring alone
])
def test_conv_small_launches_deep_ring_and_split_k(gpu_ctx, half, monkeypatch, n, cin, cout, h, w):
    """Launches that do not fill the chip (the reference's literal loop is one frame per forward) run conv_deep_kernel: 128 x 128 tiles, a four-stage
    ring and, for long K loops, split-K.  Without the split (HIVE_SPLITK=0) its results are BIT-IDENTICAL to conv_kernel's (HIVE_CONV_DEEP=0): the same
    K order into the same accumulators; split 2 / 5 / 7 ways and by the launch policy they are within the float32 reference's bound, reproducible
    from run to run, and untouched by another input's partials left in the workspace.  All four epilogues: plain, bias + ReLU, two shortcuts, and
    the GroupNorm statistics."""
    from hive_amd.dpt import ops
    x, conv = _mk(n, cin, cout, h, w, bias=True, seed=cin + h, half=half)
    g = torch.Generator(device="cpu").manual_seed(3)
    r1 = torch.randn(n, cout, h, w, generator=g).to(half).cuda().contiguous(memory_format=torch.channels_last)
    r2 = torch.randn(n, cout, h, w, generator=g).to(half).cuda().contiguous(memory_format=torch.channels_last)
    x2 = (x.float() * 0.5 + 1.0).to(half).contiguous(memory_format=torch.channels_last)
    base = F.conv2d(x.float(), conv.weight.float(), conv.bias.float(), 1, 1)

    def run(inp=x):
        a = ops.conv3x3(inp, conv, relu=True)
        b, b_relu = ops.conv3x3(inp, conv, residual=r1, residual2=r2, also_relu=True)
        c = ops.conv2d(inp, conv, gn_stats=True)
        partial, tile_rows = c.hive_gn_stats
        used = ((c.numel() // cout + tile_rows - 1) // tile_rows) * 4 * cout if tile_rows else 0  # (the rest of the buffer is never written)
        return a, b, b_relu, c, partial[:used].clone(), tile_rows

    monkeypatch.setenv("HIVE_CONV_DEEP", "0")
    plain = run()
    monkeypatch.setenv("HIVE_CONV_DEEP", "1")
    monkeypatch.setenv("HIVE_SPLITK", "0")
    ring = run()
    for got, want in zip(ring[:4], plain[:4]):
        assert torch.equal(got, want), "the ring alone must not change a bit"
    for ways in (None, "2", "5", "7"):
        if ways is None:
            monkeypatch.delenv("HIVE_SPLITK")
        else:
            monkeypatch.setenv("HIVE_SPLITK", ways)
        first = run()
        _check(first[0], F.relu(base), f"split {ways}: bias + relu")
        _check(first[1], base + r1.float() + r2.float(), f"split {ways}: two shortcuts")
        assert torch.equal(first[2], F.relu(first[1]))
        _check(first[3], base, f"split {ways}: plain with statistics")
        tile_rows = first[5]
        if tile_rows:  # the sums are those of the stored values
            rows = first[3].permute(0, 2, 3, 1).reshape(-1, cout).double()
            n_tiles = (rows.shape[0] + tile_rows - 1) // tile_rows
            got = first[4].view(n_tiles, 2, 2, cout).double().sum((0, 1))
            assert torch.allclose(got[0], rows.sum(0), rtol=1e-5, atol=1e-2) and torch.allclose(got[1], (rows ** 2).sum(0), rtol=1e-5, atol=1e-2)
        for _ in range(3):
            run(x2)  # another input's partials at the same workspace addresses
            again = run()
            for got, want in zip(again[:5], first[:5]):
                assert torch.equal(got, want), f"split {ways}: not reproducible"


@pytest.mark.parametrize("n,cin,cout,k,stride,h,w", [
    (3, 64, 256, 1, 1, 20, 24),     # HW = 480: the 256-row tiles straddle the samples; 256 output channels (one N tile)
    (2, 256, 64, 1, 1, 40, 56),     # 64 output channels: 8 waves of 32 rows
    (5, 128, 128, 3, 2, 47, 61),    # stride-2 3 x 3 "SAME" (conv2 of a stage's first block), odd sizes, ragged last tile
    (2, 256, 512, 1, 1, 30, 40),    # two N tiles write disjoint channel ranges of the tile rows
    (40, 64, 64, 1, 1, 48, 64),     # M = 122 880: more tiles than CUs (persistent workgroups run several epilogues)
    (4, 64, 128, 3, 1, 9, 13),      # HW = 117 < a tile: no statistics from the epilogue, the GroupNorm makes its own pass
])
def test_conv_epilogue_group_norm_statistics(gpu_ctx, half, n, cin, cout, k, stride, h, w):
    """hive_nhwc_conv_gn: the per-tile channel sums the epilogue leaves equal the sums over the stored output, and the GroupNorm that
    takes its statistics from them equals the GroupNorm that makes its own pass (statistics: f32 sums in another order; the
    normalised bf16 outputs may differ by one rounding in a few places)."""
    from hive_amd.dpt import ops
    from hive_amd.dpt.models import StdConv2dSame
    g = torch.Generator(device="cpu").manual_seed(n * 1000 + cout)
    conv = StdConv2dSame(cin, cout, k, stride=stride)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g))
    conv = conv.to(memory_format=torch.channels_last).to(half).cuda().eval()
    x = (torch.randn(n, cin, h, w, generator=g) + 0.3).to(half).cuda().contiguous(memory_format=torch.channels_last)
    assert ops.conv_eligible(x, conv)
    wstd = conv.standardized_weight()
    plain = ops.conv2d(x, conv, weight=wstd, same_pad=True)
    out = ops.conv2d(x, conv, weight=wstd, same_pad=True, gn_stats=True)
    assert torch.equal(out, plain), "the statistics must not change the convolution"
    partial, tile_rows = out.hive_gn_stats
    hw = out.shape[2] * out.shape[3]
    if hw < 128:
        assert tile_rows == 0
    else:
        assert tile_rows in (128, 256) and tile_rows <= hw
        rows = out.permute(0, 2, 3, 1).reshape(-1, cout).float()  # [M][C] in memory order
        m = rows.shape[0]
        n_tiles = (m + tile_rows - 1) // tile_rows
        got = partial[: n_tiles * 4 * cout].view(n_tiles, 2, 2, cout).double()
        # per sample: the tiles' first-image halves of the tiles starting in it + the second halves of the tile straddling into it
        ref_s = rows.double().view(n, hw, cout).sum(1)
        ref_q = (rows.double() ** 2).view(n, hw, cout).sum(1)
        acc_s = torch.zeros_like(ref_s)
        acc_q = torch.zeros_like(ref_q)
        for t in range(n_tiles):
            first = (t * tile_rows) // hw
            acc_s[first] += got[t, 0, 0]
            acc_q[first] += got[t, 0, 1]
            if first + 1 < n:
                acc_s[first + 1] += got[t, 1, 0]
                acc_q[first + 1] += got[t, 1, 1]
            else:
                assert got[t, 1].abs().max().item() == 0.0
        assert torch.allclose(acc_s, ref_s, rtol=1e-5, atol=1e-2), (acc_s - ref_s).abs().max().item()
        assert torch.allclose(acc_q, ref_q, rtol=1e-5, atol=1e-2), (acc_q - ref_q).abs().max().item()
    gamma = (torch.rand(cout, generator=g) + 0.5).to(half).cuda()
    beta = (torch.randn(cout, generator=g) * 0.2).to(half).cuda()
    res = torch.randn(out.shape, generator=g).to(half).cuda().contiguous(memory_format=torch.channels_last)
    for residual in (None, res):
        own = ops.group_norm_act(plain, 32, gamma, beta, 1e-5, relu=True, residual=residual, engine="hip")
        fused = ops.group_norm_act(out, 32, gamma, beta, 1e-5, relu=True, residual=residual, engine="hip", stats=out.hive_gn_stats)
        diff = (own.float() - fused.float()).abs()
        assert diff.max().item() <= 2 * _ulp(half) * max(own.float().abs().max().item(), 1.0)
        assert (diff > 0).float().mean().item() < 2e-3, "more than a few one-rounding differences"
    again = ops.conv2d(x, conv, weight=wstd, same_pad=True, gn_stats=True)
    used = ((out.numel() // cout + tile_rows - 1) // tile_rows) * 4 * cout if tile_rows else 0
    assert torch.equal(again.hive_gn_stats[0][:used], partial[:used]), "the epilogue sums are run-to-run identical"


def test_conv_gn_leaves_no_sums_behind_a_fused_epilogue(gpu_ctx, half):
    """hive_nhwc_conv_gn writes the GroupNorm sums for PLAIN epilogues (bias at most): with a ReLU or a shortcut fused into the epilogue it
    reports tile_rows 0 (the GroupNorm then makes its own pass) and is the ordinary convolution; with a bias the sums are those of the
    stored (biased, rounded) outputs."""
    import ctypes
    from hive_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(5)
    n, h, w, cin, cout = 2, 24, 32, 64, 128
    x = torch.randn(n, h, w, cin, generator=g).to(half).cuda()
    wt = (torch.randn(cout, 1, 1, cin, generator=g) * 0.2).to(half).cuda()
    bias = torch.randn(cout, generator=g).to(half).cuda()
    res = torch.randn(n, h, w, cout, generator=g).to(half).cuda()
    lib, hnd, code = gpu_ctx.lib, gpu_ctx.handle, _lib.dtype_code(half)
    nfl = int(lib.hive_nhwc_conv_gn_partial_floats(n * h * w, cout))
    ref = torch.einsum("nhwc,oc->nhwo", x.float(), wt.view(cout, cin).float())
    for relu, residual, b in ((0, None, None), (0, None, bias), (1, None, None), (0, res, None)):
        out = torch.empty(n, h, w, cout, dtype=half, device="cuda")
        partial = torch.full((nfl,), float("nan"), dtype=torch.float32, device="cuda")
        rows = ctypes.c_int(-1)
        gpu_ctx.check(lib.hive_nhwc_conv_gn(hnd, x.data_ptr(), code, n, h, w, cin, cout, 1, 1, 0, 0, h, w, wt.data_ptr(), _lib.ptr(b), relu, _lib.ptr(residual), None,
                                            out.data_ptr(), None, partial.data_ptr(), nfl, ctypes.byref(rows)))
        want = ref + (b.float() if b is not None else 0.0) + (residual.float() if residual is not None else 0.0)
        want = want.clamp_min(0.0) if relu else want
        assert (out.float() - want).abs().max().item() <= 4 * _ulp(half) * max(want.abs().max().item(), 1.0)
        if relu or residual is not None:
            assert rows.value == 0 and torch.isnan(partial).all(), "no sums (and nothing written) behind a fused ReLU / shortcut"
        else:
            assert rows.value in (128, 256)
            tiles = (n * h * w + rows.value - 1) // rows.value
            got = partial[: tiles * 4 * cout].view(tiles, 2, 2, cout).double()
            flat = out.float().view(-1, cout).double()
            per_img = h * w
            s_ref = torch.zeros(n, cout, dtype=torch.float64, device="cuda")
            s_got = torch.zeros_like(s_ref)
            for t in range(tiles):
                first = (t * rows.value) // per_img
                s_got[first] += got[t, 0, 0]
                if first + 1 < n:
                    s_got[first + 1] += got[t, 1, 0]
            for i in range(n):
                s_ref[i] = flat[i * per_img:(i + 1) * per_img].sum(0)
            assert torch.allclose(s_got, s_ref, rtol=1e-5, atol=1e-2)


@pytest.mark.parametrize("n,cin,cout,stride,h,w,with_residual", [
    (3, 64, 256, 1, 20, 24, True),      # conv3 of a stage-1 block: tiles straddle the samples
    (2, 256, 1024, 1, 30, 40, True),    # four N tiles
    (5, 256, 512, 2, 47, 61, False),    # the downsample convolution of a stage: stride 2, no shortcut, no ReLU
    (40, 64, 256, 1, 48, 64, True),     # more tiles than CUs
    (2, 64, 256, 1, 9, 13, True),       # smaller than a tile: not fused (None)
    (2, 64, 128, 1, 20, 24, True),      # 128 output channels: not fused
])
def test_conv_group_norm_two_pass_equals_the_pair(gpu_ctx, half, monkeypatch, n, cin, cout, stride, h, w, with_residual):
    """hive_nhwc_conv_gn_apply (convolution twice, its output never stored) is bit-identical to conv -> GroupNorm with the
    epilogue's statistics, and within bf16 rounding of float32 torch.  (HIVE_GN_GRAM=0: where the input is narrow the layer otherwise takes
    its statistics from the input's Gram matrices -- the same outputs to within a rounding, test_group_norm_statistics_from_the_gram_matrix.)"""
    from hive_amd.dpt import ops
    monkeypatch.setenv("HIVE_GN_GRAM", "0")
    from hive_amd.dpt.models import GroupNormAct, StdConv2dSame
    g = torch.Generator(device="cpu").manual_seed(cout + h)
    conv = StdConv2dSame(cin, cout, 1, stride=stride)
    norm = GroupNormAct(cout, apply_act=False)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g))
        norm.weight.copy_(torch.rand(cout, generator=g) + 0.5)
        norm.bias.copy_(torch.randn(cout, generator=g) * 0.2)
    conv = conv.to(memory_format=torch.channels_last).to(half).cuda().eval()
    norm = norm.to(half).cuda().eval()
    x = (torch.randn(n, cin, h, w, generator=g) + 0.3).to(half).cuda().contiguous(memory_format=torch.channels_last)
    wstd = conv.standardized_weight()
    t = ops.conv2d(x, conv, weight=wstd, same_pad=True, gn_stats=True)
    res = torch.randn(t.shape, generator=g).to(half).cuda().contiguous(memory_format=torch.channels_last) if with_residual else None
    relu = with_residual
    pair = ops.group_norm_act(t, 32, norm.weight, norm.bias, norm.eps, relu=relu, residual=res, engine="hip", stats=t.hive_gn_stats)
    fused = ops.conv_gn_act(x, conv, norm, weight=wstd, same_pad=True, relu=relu, residual=res)
    if t.shape[2] * t.shape[3] < 256 or cout % 256:
        assert fused is None
        return
    assert fused is not None and torch.equal(fused, pair)
    ref = F.group_norm(F.conv2d(x.float(), wstd.float(), None, stride), 32, norm.weight.float(), norm.bias.float(), norm.eps)
    if res is not None:
        ref = F.relu(ref + res.float())
    err = (fused.float() - ref).abs().max().item()
    assert err <= 0.04 * max(ref.abs().max().item(), 1.0), err


@pytest.mark.parametrize("n,cin,cout,stride,h,w", [
    (3, 64, 256, 1, 20, 24),      # conv3 of a stage-1 block: four waves split the pixels, HW = 480 is not a multiple of the 128-pixel tile
    (2, 128, 512, 1, 37, 45),     # four 64 x 64 blocks, one per wave; ragged chunks
    (2, 256, 1024, 1, 30, 40),    # sixteen blocks on eight waves
    (3, 256, 512, 2, 47, 61),     # the downsample convolution of a stage: stride 2 (every other pixel of every other row)
    (9, 64, 256, 1, 120, 160),    # a full-size map: several chunks per sample
])
def test_group_norm_statistics_from_the_gram_matrix(gpu_ctx, half, n, cin, cout, stride, h, w):
    """GroupNorm statistics of a 1 x 1 convolution from the input's Gram matrix (csrc/gram.hip: sum y^2 = <sum_c w_c w_c^T, sum_p x_p x_p^T>): the partial Gram
    matrices and channel sums add up to X^T X and X^T 1 of the sampled pixels (float64 reference of the same 16-bit values: 1e-5 relative -- float32
    accumulation on the matrix cores), (mean, rstd) equal those of the float64 convolution output to 2e-4 relative, and hive_nhwc_conv_gn_apply_gram equals
    the two-pass form that takes its statistics from the convolution itself to within one rounding in a few places."""
    import ctypes
    from hive_amd import _lib
    from hive_amd.dpt import ops
    from hive_amd.dpt.models import GroupNormAct, StdConv2dSame
    g = torch.Generator(device="cpu").manual_seed(cout + h)
    conv = StdConv2dSame(cin, cout, 1, stride=stride)
    norm = GroupNormAct(cout, apply_act=False)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g))
        norm.weight.copy_(torch.rand(cout, generator=g) + 0.5)
        norm.bias.copy_(torch.randn(cout, generator=g) * 0.2)
    conv = conv.to(memory_format=torch.channels_last).to(half).cuda().eval()
    norm = norm.to(half).cuda().eval()
    x = F.relu(torch.randn(n, cin, h, w, generator=g) + 0.3).to(half).cuda().contiguous(memory_format=torch.channels_last)  # (the backbone's inputs are post-ReLU)
    wstd = conv.standardized_weight().contiguous(memory_format=torch.channels_last)
    oh, ow = (h + stride - 1) // stride, (w + stride - 1) // stride
    ctx, lib, G = gpu_ctx, gpu_ctx.lib, 32
    tables = torch.empty(int(lib.hive_gn_gram_table_floats(cin, G)), dtype=torch.float32, device="cuda")
    ctx.check(lib.hive_gn_gram_prepare(ctx.handle, wstd.data_ptr(), _lib.dtype_code(half), cin, cout, G, tables.data_ptr()))
    parts = int(lib.hive_gn_gram_parts(ctx.handle, n, cin, oh, ow))
    S = torch.empty(n, parts, cin, cin, dtype=torch.float32, device="cuda")
    s = torch.empty(n, parts, cin, dtype=torch.float32, device="cuda")
    stats = torch.empty(n, G, 2, dtype=torch.float32, device="cuda")
    ctx.check(lib.hive_gn_gram_stats(ctx.handle, x.data_ptr(), _lib.dtype_code(half), n, h, w, cin, cout, stride, oh, ow, G, tables.data_ptr(), norm.eps,
                                     stats.data_ptr(), S.data_ptr(), s.data_ptr()))
    xs = x[:, :, ::stride, ::stride].permute(0, 2, 3, 1).reshape(n, oh * ow, cin).double()  # the pixels a 1 x 1 convolution of this stride reads
    S_ref, s_ref = xs.transpose(1, 2) @ xs, xs.sum(1)
    assert ((S.double().sum(1) - S_ref).abs().max() / S_ref.abs().max()).item() < 1e-5
    assert ((s.double().sum(1) - s_ref).abs().max() / s_ref.abs().max()).item() < 1e-5
    w2 = wstd.reshape(cout, cin).double()
    y = xs @ w2.t()                                                                        # [n][pixels][cout]
    yg = y.reshape(n, oh * ow, G, cout // G)
    mean_ref = yg.mean(dim=(1, 3))
    rstd_ref = 1.0 / torch.sqrt(yg.var(dim=(1, 3), unbiased=False) + norm.eps)
    scale = y.std().item()
    assert (stats[..., 0].double() - mean_ref).abs().max().item() < 2e-4 * scale
    assert ((stats[..., 1].double() - rstd_ref).abs() / rstd_ref).max().item() < 2e-4
    # the whole operation against the two-pass form
    res = torch.randn(n, cout, oh, ow, generator=g).to(half).cuda().contiguous(memory_format=torch.channels_last)
    two_pass = ops.conv_gn_act(x, conv, norm, weight=wstd, same_pad=True, relu=True, residual=res)
    assert two_pass is not None
    out = torch.empty_like(two_pass)
    scratch = torch.empty(2 * n * G, dtype=torch.float32, device="cuda")
    fused = ctypes.c_int(0)
    ctx.check(lib.hive_nhwc_conv_gn_apply_gram(ctx.handle, x.data_ptr(), _lib.dtype_code(half), n, h, w, cin, cout, stride, oh, ow, wstd.data_ptr(), tables.data_ptr(), G,
                                               norm.weight.data_ptr(), norm.bias.data_ptr(), norm.eps, res.data_ptr(), 1, out.data_ptr(), scratch.data_ptr(),
                                               scratch.numel(), ctypes.byref(fused)))
    assert fused.value == 1
    diff = (out.float() - two_pass.float()).abs()
    assert diff.max().item() <= 2 * _ulp(half) * max(two_pass.float().abs().max().item(), 1.0)
    assert (diff > 0).float().mean().item() < 5e-3, "more than a few one-rounding differences"
    again = torch.empty_like(out)
    ctx.check(lib.hive_nhwc_conv_gn_apply_gram(ctx.handle, x.data_ptr(), _lib.dtype_code(half), n, h, w, cin, cout, stride, oh, ow, wstd.data_ptr(), tables.data_ptr(), G,
                                               norm.weight.data_ptr(), norm.bias.data_ptr(), norm.eps, res.data_ptr(), 1, again.data_ptr(), scratch.data_ptr(),
                                               scratch.numel(), ctypes.byref(fused)))
    assert torch.equal(again, out), "reproducible: fixed summation orders"


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 64, 256, 120, 160), (2, 128, 512, 60, 80)])
def test_gram_statistics_with_large_means_and_outlier_channels(gpu_ctx, half, n, cin, cout, h, w):
    """the Gram form takes var = E[y^2] - mean^2 from float32 Gram matrices; inputs whose channels carry a large common component (post-ReLU maps with
    means of many sigma, a few outlier channels tens of times larger than the rest) make the gross terms large against the net variance.  (mean, rstd) against
    the float64 statistics of the same 16-bit inputs and weights: the mean to 1e-3 of the outputs' spread, rstd to 2e-3 relative -- the bound a GroupNorm output
    rounded to 8 significant bits (bfloat16) cannot see."""
    from hive_amd import _lib
    from hive_amd.dpt.models import StdConv2dSame
    g = torch.Generator(device="cpu").manual_seed(cin + h)
    conv = StdConv2dSame(cin, cout, 1)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g))
    conv = conv.to(memory_format=torch.channels_last).to(half).cuda().eval()
    mu = torch.full((cin,), 0.3)
    mu[torch.randperm(cin, generator=g)[:cin // 6]] = 5.0     # channels that sit 5 sigma above zero
    mu[torch.randperm(cin, generator=g)[:cin // 16]] = 20.0   # ... and 20 sigma
    amp = torch.ones(cin)
    amp[torch.randperm(cin, generator=g)[:2]] = 30.0           # two outlier channels
    x = F.relu((torch.randn(n, cin, h, w, generator=g) + mu[None, :, None, None]) * amp[None, :, None, None])
    x = x.to(half).cuda().contiguous(memory_format=torch.channels_last)
    wstd = conv.standardized_weight().contiguous(memory_format=torch.channels_last)
    ctx, lib, G = gpu_ctx, gpu_ctx.lib, 32
    tables = torch.empty(int(lib.hive_gn_gram_table_floats(cin, G)), dtype=torch.float32, device="cuda")
    ctx.check(lib.hive_gn_gram_prepare(ctx.handle, wstd.data_ptr(), _lib.dtype_code(half), cin, cout, G, tables.data_ptr()))
    stats = torch.empty(n, G, 2, dtype=torch.float32, device="cuda")
    ctx.check(lib.hive_gn_gram_stats(ctx.handle, x.data_ptr(), _lib.dtype_code(half), n, h, w, cin, cout, 1, h, w, G, tables.data_ptr(), 1e-5, stats.data_ptr(), None, None))
    xs = x.permute(0, 2, 3, 1).reshape(n, h * w, cin).double()
    y = xs @ wstd.reshape(cout, cin).double().t()
    yg = y.reshape(n, h * w, G, cout // G)
    mean_ref = yg.mean(dim=(1, 3))
    std_ref = torch.sqrt(yg.var(dim=(1, 3), unbiased=False) + 1e-5)
    gross = ((yg ** 2).mean(dim=(1, 3)) / std_ref ** 2).max().item()
    e_mean = ((stats[..., 0].double() - mean_ref).abs() / std_ref).max().item()
    e_rstd = ((stats[..., 1].double() * std_ref) - 1.0).abs().max().item()
    print(f"gram statistics under stress ({half}, C_in {cin}): E[y^2] / var up to {gross:.1f}; mean error {e_mean:.2e} sigma, rstd error {e_rstd:.2e} relative")
    assert e_mean < 1e-3 and e_rstd < 2e-3, (e_mean, e_rstd, gross)


def test_patch_embed_and_conv_transpose_match_torch(gpu_ctx, half):
    """DPT-Large's non-standard convolutions through the hand-written kernels: the 16 x 16 / 16 patch embedding (hive_patch_rows +
    the GEMM) and ConvTranspose2d with kernel == stride 4 and 2 (1 x 1 convolution + hive_nhwc_pixel_shuffle_bias), against float32 torch."""
    from hive_amd.dpt import ops
    g = torch.Generator(device="cpu").manual_seed(11)
    x = torch.randn(2, 3, 96, 160, generator=g).to(half).cuda().contiguous(memory_format=torch.channels_last)
    pe = nn.Conv2d(3, 1024, 16, 16)
    with torch.no_grad():
        pe.weight.copy_(torch.randn(pe.weight.shape, generator=g) * 0.05)
        pe.bias.copy_(torch.randn(1024, generator=g) * 0.1)
    pe = pe.to(memory_format=torch.channels_last).to(half).cuda()
    assert ops.patch_embed_eligible(x, pe)
    tok = ops.patch_embed(x, pe)
    ref = F.conv2d(x.float(), pe.weight.float(), pe.bias.float(), 16).flatten(2).transpose(1, 2)
    assert tok.shape == ref.shape == (2, 60, 1024)
    assert (tok.float() - ref).abs().max().item() <= 2 * _ulp(half) * ref.abs().max().item() + 0.25 * _ulp(half)
    with torch.no_grad():
        pe.bias.add_(1.0)  # the derived weights follow a parameter update
    assert (ops.patch_embed(x, pe).float() - (ref + 1.0)).abs().max().item() <= 4 * _ulp(half) * (ref.abs().max().item() + 1.0)
    for cin, s in ((256, 4), (512, 2)):
        y = torch.randn(2, cin, 6, 10, generator=g).to(half).cuda().contiguous(memory_format=torch.channels_last)
        ct = nn.ConvTranspose2d(cin, cin, s, s, 0, bias=True)
        with torch.no_grad():
            ct.weight.copy_(torch.randn(ct.weight.shape, generator=g) * (1.0 / cin) ** 0.5)
            ct.bias.copy_(torch.randn(cin, generator=g) * 0.2)
        ct = ct.to(half).cuda()
        assert ops.conv_transpose_eligible(y, ct)
        out = ops.conv_transpose(y, ct)
        ref = F.conv_transpose2d(y.float(), ct.weight.float(), ct.bias.float(), s)
        assert out.shape == ref.shape and out.is_contiguous(memory_format=torch.channels_last)
        # the 1 x 1 convolution's result is rounded to bf16 before the bias is added (two roundings)
        assert (out.float() - ref).abs().max().item() <= 4 * _ulp(half) * ref.abs().max().item() + 0.25 * _ulp(half)


def test_new_entry_points_reject_bad_arguments(gpu_ctx, half):
    """Argument checks of the round-2 additions come back as HiveError (never a launch on bad shapes)."""
    import ctypes
    from hive_amd import _lib
    lib, h = gpu_ctx.lib, gpu_ctx.handle
    x = torch.zeros(1, 64, 16, 16, device="cuda", dtype=half).contiguous(memory_format=torch.channels_last)
    w = torch.zeros(256, 64, 1, 1, device="cuda", dtype=half)
    out = torch.zeros(1, 256, 16, 16, device="cuda", dtype=half).contiguous(memory_format=torch.channels_last)
    small = torch.zeros(16, device="cuda")
    tile_rows, fused = ctypes.c_int(-1), ctypes.c_int(-1)
    with pytest.raises(_lib.HiveError, match="gn_partial holds"):
        gpu_ctx.check(lib.hive_nhwc_conv_gn(h, x.data_ptr(), _lib.dtype_code(half), 1, 16, 16, 64, 256, 1, 1, 0, 0, 16, 16, w.data_ptr(), None, 0, None, None, out.data_ptr(), None,
                                            small.data_ptr(), small.numel(), ctypes.byref(tile_rows)))
    g = torch.ones(256, device="cuda", dtype=half)
    with pytest.raises(_lib.HiveError, match="scratch holds"):
        gpu_ctx.check(lib.hive_nhwc_conv_gn_apply(h, x.data_ptr(), _lib.dtype_code(half), 1, 16, 16, 64, 256, 1, 1, 0, 0, 16, 16, w.data_ptr(), 32, g.data_ptr(), g.data_ptr(), 1e-5,
                                                  None, 1, out.data_ptr(), small.data_ptr(), small.numel(), ctypes.byref(fused)))
    # not eligible (128 output channels): reported through *fused = 0, not as an error, and nothing is launched
    w2 = torch.zeros(128, 64, 1, 1, device="cuda", dtype=half)
    gpu_ctx.check(lib.hive_nhwc_conv_gn_apply(h, x.data_ptr(), _lib.dtype_code(half), 1, 16, 16, 64, 128, 1, 1, 0, 0, 16, 16, w2.data_ptr(), 32, g.data_ptr(), g.data_ptr(), 1e-5,
                                              None, 1, out.data_ptr(), small.data_ptr(), small.numel(), ctypes.byref(fused)))
    assert fused.value == 0
    frame = torch.zeros(1, 3, 40, 48, device="cuda", dtype=half).contiguous(memory_format=torch.channels_last)
    with pytest.raises(_lib.HiveError, match="multiples of the patch size"):
        gpu_ctx.check(lib.hive_patch_rows(h, frame.data_ptr(), _lib.dtype_code(half), 1, 40, 48, 3, 16, out.data_ptr()))
    with pytest.raises(_lib.HiveError, match="pixel_shuffle_bias"):
        gpu_ctx.check(lib.hive_nhwc_pixel_shuffle_bias(h, out.data_ptr(), None, _lib.dtype_code(half), 1, 4, 4, 12, 2, x.data_ptr()))


@pytest.mark.parametrize("n,h,w", [(3, 120, 160), (2, 24, 32), (2, 37, 45)])
def test_bottleneck_gn_conv3x3_equals_the_pair(gpu_ctx, half, n, h, w):
    """hive_bneck_gn_conv3x3 (csrc/bneck.hip): conv2(relu(norm1(t))) of a 64-channel bottleneck as one kernel is bit-identical to the GroupNorm
    pass followed by the general convolution; the sums it leaves for norm2 are those of its stored outputs; maps that are not whole 32-wide
    tiles leave none (norm2 then makes its own pass)."""
    from hive_amd.dpt import ops
    from hive_amd.dpt.models import GroupNormAct, StdConv2dSame
    g = torch.Generator(device="cpu").manual_seed(h)
    conv1, conv2 = StdConv2dSame(64, 64, 1), StdConv2dSame(64, 64, 3)
    norm1 = GroupNormAct(64)
    with torch.no_grad():
        conv1.weight.copy_(torch.randn(conv1.weight.shape, generator=g))
        conv2.weight.copy_(torch.randn(conv2.weight.shape, generator=g))
        norm1.weight.copy_(torch.rand(64, generator=g) + 0.5)
        norm1.bias.copy_(torch.randn(64, generator=g) * 0.3)
    conv1 = conv1.to(memory_format=torch.channels_last).to(half).cuda().eval()
    conv2 = conv2.to(memory_format=torch.channels_last).to(half).cuda().eval()
    norm1 = norm1.to(half).cuda().eval()
    norm1.engine = "hip"
    x = (torch.randn(n, 64, h, w, generator=g) + 0.2).to(half).cuda().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        t = ops.conv2d(x, conv1, weight=conv1.standardized_weight(), same_pad=True, gn_stats=True)
        assert t.hive_gn_stats[1] > 0
        u = ops.group_norm_act(t, 32, norm1.weight, norm1.bias, norm1.eps, relu=True, engine="hip", stats=t.hive_gn_stats)
        pair = ops.conv2d(u, conv2, weight=conv2.standardized_weight(), same_pad=True, gn_stats=True)
        one = ops.bneck_gn_conv3x3(t, norm1, conv2, conv2.standardized_weight())
    assert one is not None and torch.equal(one, pair), float((one.float() - pair.float()).abs().max())
    stats = getattr(one, "hive_gn_stats", None)
    assert (stats is not None) == (w % 32 == 0 and (h * w) % (((h + 15) // 16) * (w // 32)) == 0)
    if stats is not None:
        partial, tile_rows = stats
        per_img = ((h + 15) // 16) * (w // 32)
        assert tile_rows == h * w // per_img
        got = partial[: n * per_img * 4 * 64].view(n, per_img, 2, 2, 64).double().sum(1)  # per image
        rows = one.permute(0, 2, 3, 1).reshape(n, h * w, 64).double()
        assert torch.allclose(got[:, 0, 0], rows.sum(1), rtol=1e-5, atol=1e-2) and torch.allclose(got[:, 0, 1], (rows ** 2).sum(1), rtol=1e-5, atol=1e-2)
        assert got[:, 1].abs().max().item() == 0.0
    # not a 64-channel stride-1 bottleneck: does not apply
    conv_s2 = StdConv2dSame(64, 64, 3, stride=2).to(memory_format=torch.channels_last).to(half).cuda().eval()
    assert ops.bneck_gn_conv3x3(t, norm1, conv_s2, conv_s2.standardized_weight()) is None
