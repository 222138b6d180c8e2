"""DPT-Hybrid (ViT-B/16 + ResNetV2-50 stem) monocular depth network for the MI355X.

Replaces ``dpt.models.DPTDepthModel`` of the reference's (absent) third_party/dpt (AnthonyDickson/DPT,
a fork of isl-org/DPT, with timm==0.5.4's ``vit_base_resnet50_384``); call sites
/root/reference/hive/dataset_adaptors.py:51,1366-1374,1419.  The module tree and parameter names are
those of the published checkpoints (``pretrained.model.*``, ``pretrained.act_postprocess*``,
``scratch.*``), so ``dpt_hybrid_nyu-2ce69ec7.pt`` loads with ``load_state_dict`` unchanged; without a
checkpoint (``path=None``) the network is randomly initialised -- parity unpinned.

What runs where with ``engine="hip"`` (the default) -- every layer on a hand-written gfx950 kernel behind the C ABI, in the
model's 16-bit type: bfloat16, or float16 exactly as the reference runs it (``model.half()`` + fp16 samples,
dataset_adaptors.py:1394-1401, 1415-1417):
  * ViT encoder blocks (LayerNorm, QKV / proj / MLP GEMMs, attention) and the readout projections: MFMA GEMMs with fused
    epilogues and an LDS-tiled flash attention -- ``hive_amd/csrc/vit.hip``.
  * every convolution (ResNetV2 stem and stages, reassemble, RefineNet fusion, head): implicit-GEMM MFMA kernels on channels-last
    activations with bias / ReLU / skip adds / GroupNorm statistics in the epilogue -- ``csrc/conv.hip``, ``csrc/stem.hip``.
  * GroupNorm(+residual+ReLU), x2 bilinear upsampling, max pool, and the fused depth head (upsample + conv 128->32 + ReLU +
    conv 32->1 + inversion + mm hand-off) -- ``csrc/dpt_ops.hip``, ``csrc/dpt_head.hip``.
A layer (or a tensor: float32, CPU, not channels-last) the kernels do not cover RAISES ``HiveError``: the HIP engine never drops
to a PyTorch operator.  ``engine="torch"`` is the same module tree on plain PyTorch ops: the float32 reference the numerics tests
compare against, and the float32 network of ``estimate_depth_dpt(optimize=False)``.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from hive_amd.dpt import ops as dpt_ops


# ------------------------------------------------------------------------------------------------
# ResNetV2-50 (layers 3, 4, 9; weight-standardised "same"-padded convs, GroupNorm) -- timm 0.5.4
def _same_pad(size, k, s):
    return max((math.ceil(size / s) - 1) * s + k - size, 0)


class StdConv2dSame(nn.Conv2d):
    """Conv2d with weight standardisation and TensorFlow "SAME" padding (timm ``StdConv2dSame``, eps=1e-8
    for the ViT hybrids).  The standardised weight is cached in eval mode (weights are frozen there)."""

    engine = "torch"  # set to "hip" by DPT(engine="hip"): hand-written implicit-GEMM convolution (csrc/conv.hip)

    def __init__(self, in_chs, out_chs, kernel_size, stride=1, eps=1e-8):
        super().__init__(in_chs, out_chs, kernel_size, stride=stride, padding=0, bias=False)
        self.eps = eps
        self._std_weight, self._std_stamp = None, None

    def standardized_weight(self):
        w = self.weight
        stamp = (w.data_ptr(), w._version, w.dtype, w.device)  # load_state_dict / optimiser steps write in place: _version moves
        if not self.training and self._std_weight is not None and self._std_stamp == stamp:
            return self._std_weight
        w32 = w.float()
        flat = w32.reshape(w32.shape[0], -1)
        mean = flat.mean(dim=1, keepdim=True)
        var = flat.var(dim=1, unbiased=False, keepdim=True)
        std_w = ((flat - mean) / torch.sqrt(var + self.eps)).reshape_as(w32).to(w.dtype)
        if w.dim() == 4 and w.is_contiguous(memory_format=torch.channels_last):
            std_w = std_w.contiguous(memory_format=torch.channels_last)
        if not self.training:
            self._std_weight, self._std_stamp = std_w.detach(), stamp
        return std_w

    def train(self, mode=True):
        self._std_weight = None
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        self._std_weight = None
        return super()._apply(fn, *args, **kwargs)

    def forward(self, x):
        if self.engine == "hip":
            if dpt_ops.stem_conv_eligible(x, self):
                return dpt_ops.stem_conv(x, self, self.standardized_weight())
            dpt_ops.require_conv(x, self, "weight-standardised convolution")
            return dpt_ops.conv2d(x, self, weight=self.standardized_weight(), same_pad=True, gn_stats=True)  # a GroupNorm follows every StdConv2dSame
        ih, iw = x.shape[-2:]
        kh, kw = self.kernel_size
        ph, pw = _same_pad(ih, kh, self.stride[0]), _same_pad(iw, kw, self.stride[1])
        if ph % 2 == 0 and pw % 2 == 0:
            # symmetric "SAME" padding is the convolution's own zero padding: no padded copy of the input
            return F.conv2d(x, self.standardized_weight(), None, self.stride, (ph // 2, pw // 2), self.dilation, self.groups)
        x = F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
        return F.conv2d(x, self.standardized_weight(), None, self.stride, 0, self.dilation, self.groups)


class GroupNormAct(nn.GroupNorm):
    engine = "torch"  # set to "hip" by DPT(engine="hip"): fused channels-last kernel (csrc/dpt_ops.hip)

    def __init__(self, num_channels, num_groups=32, eps=1e-5, apply_act=True):
        super().__init__(num_groups, num_channels, eps=eps)
        self.apply_act = apply_act

    def forward(self, x, residual=None):
        """relu?(group_norm(x) (+ residual)); with a residual the ReLU is always applied (bottleneck tail)."""
        return dpt_ops.group_norm_act(x, self.num_groups, self.weight, self.bias, self.eps, relu=self.apply_act or residual is not None,
                                      residual=residual, engine=self.engine, stats=getattr(x, "hive_gn_stats", None))


class MaxPool2dSame(nn.Module):
    engine = "torch"

    def __init__(self, kernel_size=3, stride=2):
        super().__init__()
        self.k, self.s = kernel_size, stride

    def forward(self, x):
        if self.engine == "hip":
            if (self.k, self.s) != (3, 2):
                dpt_ops.not_covered("max pool", f"kernel {self.k}, stride {self.s}: 3 / 2 only")
            return dpt_ops.maxpool3x3s2_same(x, engine="hip")
        ih, iw = x.shape[-2:]
        ph, pw = _same_pad(ih, self.k, self.s), _same_pad(iw, self.k, self.s)
        x = F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2), value=float("-inf"))
        return F.max_pool2d(x, self.k, self.s)


def _conv_norm(conv, norm, x, residual=None):
    """The expanding 1 x 1 convolutions of a bottleneck with their GroupNorm (+ shortcut + ReLU) as one two-pass operation
    (hive_nhwc_conv_gn_apply), or None: not the hip engine / not eligible."""
    if conv.engine != "hip" or norm.engine != "hip" or not dpt_ops.conv_eligible(x, conv):
        return None
    return dpt_ops.conv_gn_act(x, conv, norm, weight=conv.standardized_weight(), same_pad=True, relu=norm.apply_act or residual is not None,
                               residual=residual)


class DownsampleConv(nn.Module):
    def __init__(self, in_chs, out_chs, stride):
        super().__init__()
        self.conv = StdConv2dSame(in_chs, out_chs, 1, stride=stride)
        self.norm = GroupNormAct(out_chs, apply_act=False)

    def forward(self, x):
        y = _conv_norm(self.conv, self.norm, x)
        return y if y is not None else self.norm(self.conv(x))


class Bottleneck(nn.Module):
    """Non pre-activation bottleneck: 1x1 -> 3x3(stride) -> 1x1, GroupNorm after each conv, ReLU after the sum."""

    def __init__(self, in_chs, out_chs, stride, downsample):
        super().__init__()
        mid = out_chs // 4
        self.downsample = DownsampleConv(in_chs, out_chs, stride) if downsample else None
        self.conv1 = StdConv2dSame(in_chs, mid, 1)
        self.norm1 = GroupNormAct(mid)
        self.conv2 = StdConv2dSame(mid, mid, 3, stride=stride)
        self.norm2 = GroupNormAct(mid)
        self.conv3 = StdConv2dSame(mid, out_chs, 1)
        self.norm3 = GroupNormAct(out_chs, apply_act=False)

    def forward(self, x):
        shortcut = x if self.downsample is None else self.downsample(x)
        t = self.conv1(x)
        t2 = None
        if self.conv1.engine == "hip" and self.norm1.engine == "hip" and self.conv2.engine == "hip":
            # 64-channel bottlenecks: norm1 + ReLU applied while conv2 stages its input (one kernel; None elsewhere)
            t2 = dpt_ops.bneck_gn_conv3x3(t, self.norm1, self.conv2, self.conv2.standardized_weight())
        x = self.norm2(t2 if t2 is not None else self.conv2(self.norm1(t)))
        y = _conv_norm(self.conv3, self.norm3, x, residual=shortcut)
        return y if y is not None else self.norm3(self.conv3(x), residual=shortcut)  # relu(norm3(.) + shortcut)


class ResNetStage(nn.Module):
    def __init__(self, in_chs, out_chs, stride, depth):
        super().__init__()
        self.blocks = nn.Sequential(*[Bottleneck(in_chs if i == 0 else out_chs, out_chs, stride if i == 0 else 1, i == 0)
                                      for i in range(depth)])

    def forward(self, x):
        return self.blocks(x)


class ResNetV2Stem(nn.Sequential):
    def __init__(self, in_chs=3, out_chs=64):
        super().__init__()
        self.conv = StdConv2dSame(in_chs, out_chs, 7, stride=2)
        self.norm = GroupNormAct(out_chs)
        self.pool = MaxPool2dSame(3, 2)

    def forward(self, x):
        if self.conv.engine == "hip" and self.norm.engine == "hip" and self.pool.engine == "hip" and (self.pool.k, self.pool.s) == (3, 2) and self.norm.apply_act:
            # GroupNorm + ReLU + max pool in one pass behind the convolution (bit-identical to the three modules in sequence)
            y = self.conv(x)
            return dpt_ops.group_norm_relu_maxpool(y, self.norm, stats=getattr(y, "hive_gn_stats", None))
        return super().forward(x)


class ResNetV2(nn.Module):
    def __init__(self, layers=(3, 4, 9), channels=(256, 512, 1024)):
        super().__init__()
        self.stem = ResNetV2Stem()
        stages, prev = [], 64
        for i, (depth, chs) in enumerate(zip(layers, channels)):
            stages.append(ResNetStage(prev, chs, 1 if i == 0 else 2, depth))
            prev = chs
        self.stages = nn.Sequential(*stages)
        self.norm = nn.Identity()
        self.num_features = prev


class HybridEmbed(nn.Module):
    def __init__(self, embed_dim=768):
        super().__init__()
        self.backbone = ResNetV2()
        self.proj = nn.Conv2d(self.backbone.num_features, embed_dim, kernel_size=1, stride=1)


# ------------------------------------------------------------------------------------------------
# ViT-B encoder
class Attention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = (q @ k.transpose(-2, -1)) * self.scale
        attn = attn.softmax(dim=-1)
        x = (attn @ v).transpose(1, 2).reshape(B, N, C)
        return self.proj(x)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class PatchEmbed(nn.Module):
    """timm ``PatchEmbed``: one 16 x 16, stride-16 convolution (ViT-L/16 of DPT-Large)."""

    def __init__(self, embed_dim=1024, patch=16):
        super().__init__()
        self.proj = nn.Conv2d(3, embed_dim, kernel_size=patch, stride=patch)


class VisionTransformerHybrid(nn.Module):
    """``timm.vit_base_resnet50_384`` (``hybrid=True``) / ``timm.vit_large_patch16_384`` as DPT uses them
    (``forward_flex``: any input size that is a multiple of 16, position embedding resized bilinearly from its
    24 x 24 training grid)."""

    def __init__(self, embed_dim=768, depth=12, num_heads=12, train_grid=24, hybrid=True):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.patch_size = [16, 16]
        self.patch_embed = HybridEmbed(embed_dim) if hybrid else PatchEmbed(embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, train_grid * train_grid + 1, embed_dim))
        self.blocks = nn.Sequential(*[Block(embed_dim, num_heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)

    def resize_pos_embed(self, gs_h, gs_w):
        tok, grid = self.pos_embed[:, :1], self.pos_embed[0, 1:]
        gs_old = int(math.sqrt(grid.shape[0]))
        grid = grid.reshape(1, gs_old, gs_old, -1).permute(0, 3, 1, 2)
        grid = F.interpolate(grid.float(), size=(gs_h, gs_w), mode="bilinear").to(tok.dtype)
        grid = grid.permute(0, 2, 3, 1).reshape(1, gs_h * gs_w, -1)
        return torch.cat([tok, grid], dim=1)


# ------------------------------------------------------------------------------------------------
# DPT decoder
class ProjectReadout(nn.Module):
    engine = "torch"

    def __init__(self, in_features, start_index=1):
        super().__init__()
        self.start_index = start_index
        self.project = nn.Sequential(nn.Linear(2 * in_features, in_features), nn.GELU())

    def forward(self, x):
        readout = x[:, 0].unsqueeze(1).expand_as(x[:, self.start_index:])
        cat = torch.cat((x[:, self.start_index:], readout), -1)
        lin = self.project[0]
        if self.engine == "hip":
            # Linear + GELU in the hand-written GEMM (bias and erf-GELU in its epilogue, one rounding): csrc/vit.hip
            from hive_amd import _lib
            if not (cat.is_cuda and cat.dtype in dpt_ops.HALF_TYPES and lin.weight.dtype == cat.dtype and lin.out_features % 128 == 0 and lin.in_features % 64 == 0):
                dpt_ops.not_covered("readout projection", f"{cat.dtype} on {cat.device}, {lin.in_features} -> {lin.out_features}")
            key = (lin.bias.data_ptr(), lin.bias._version)
            if getattr(self, "_bias32", (None,))[0] != key:
                self._bias32 = (key, lin.bias.detach().float().contiguous())
            b, n, k = cat.shape
            cat = cat.contiguous()
            out = torch.empty((b, n, lin.out_features), dtype=cat.dtype, device=cat.device)
            ctx = _lib.default_context(cat.device.index or 0)
            ctx.check(ctx.lib.hive_vit_linear(ctx.handle, cat.data_ptr(), dpt_ops._code(cat.dtype), lin.weight.data_ptr(), self._bias32[1].data_ptr(), None,
                                              out.data_ptr(), b * n, lin.out_features, k, 1))
            return out
        return self.project(cat)


class Transpose(nn.Module):
    def __init__(self, dim0, dim1):
        super().__init__()
        self.dim0, self.dim1 = dim0, dim1

    def forward(self, x):
        return x.transpose(self.dim0, self.dim1)


class ResidualConvUnit(nn.Module):
    engine = "torch"

    def __init__(self, features):
        super().__init__()
        self.conv1 = nn.Conv2d(features, features, 3, 1, 1, bias=True)
        self.conv2 = nn.Conv2d(features, features, 3, 1, 1, bias=True)

    def forward(self, x, skip=None, x_relu=None, also_relu=False):
        # out = conv2(relu(conv1(relu(x)))) + x (+ skip), with bias / ReLU / skip adds fused behind each convolution.
        # x_relu: relu(x) if the producer of x already wrote it; also_relu: return (out, relu(out)) for the next unit.
        out = dpt_ops.conv_bias_act(F.relu(x) if x_relu is None else x_relu, self.conv1, relu=True, engine=self.engine)
        return dpt_ops.conv_bias_act(out, self.conv2, relu=False, residual=x, residual2=skip, engine=self.engine, also_relu=also_relu)


class FeatureFusionBlock(nn.Module):
    engine = "torch"

    def __init__(self, features):
        super().__init__()
        self.out_conv = nn.Conv2d(features, features, 1, 1, 0, bias=True)
        self.resConfUnit1 = ResidualConvUnit(features)
        self.resConfUnit2 = ResidualConvUnit(features)

    def forward(self, *xs, relu_of_last=None):
        """``relu_of_last``: relu(xs[-1]) if its producer already wrote it (saves the residual unit a pass)."""
        output, output_relu = xs[0], relu_of_last
        if len(xs) == 2:  # output + resConfUnit1(xs[1]), and its ReLU for resConfUnit2 from the same kernel
            output, output_relu = self.resConfUnit1(xs[1], skip=output, also_relu=True, x_relu=relu_of_last)
        output = self.resConfUnit2(output, x_relu=output_relu)
        if self.engine == "hip":
            # The reference interpolates, then applies the 1x1 out_conv.  A 1x1 convolution (per-pixel, with
            # bias) and bilinear interpolation (per-channel, weights summing to 1) commute exactly in real
            # arithmetic, so the projection runs on a quarter of the pixels; the results differ only by 16-bit
            # rounding order (covered by the tolerance of tests/test_vit_gpu.py against the torch engine).
            # (and its bias is added by the upsampling kernel while loading)
            oc = self.out_conv
            dpt_ops.require_conv(output, oc, "out_conv of a fusion block")
            low = dpt_ops.conv2d(output, oc, with_bias=False)
            return dpt_ops.upsample2x(low, engine=self.engine, bias=oc.bias)
        output = dpt_ops.upsample2x(output, engine=self.engine)  # bilinear, align_corners=True
        return self.out_conv(output)


class Interpolate(nn.Module):
    engine = "torch"

    def __init__(self, scale_factor, mode, align_corners=False):
        super().__init__()
        self.scale_factor, self.mode, self.align_corners = scale_factor, mode, align_corners

    def forward(self, x):
        if self.scale_factor == 2 and self.mode == "bilinear" and self.align_corners:
            return dpt_ops.upsample2x(x, engine=self.engine)
        if self.engine == "hip":
            dpt_ops.not_covered("Interpolate", f"scale {self.scale_factor}, {self.mode}, align_corners={self.align_corners}: x2 bilinear align_corners only")
        return F.interpolate(x, scale_factor=self.scale_factor, mode=self.mode, align_corners=self.align_corners)


BACKBONES = {
    # name: (hybrid, ViT width, depth, heads, reassemble features, hooks) -- isl-org/DPT `_make_encoder`
    "vitb_rn50_384": (True, 768, 12, 12, (256, 512, 768, 768), (0, 1, 8, 11)),
    "vitl16_384": (False, 1024, 24, 16, (256, 512, 1024, 1024), (5, 11, 17, 23)),
}


class _Pretrained(nn.Module):
    """Backbone + reassemble stages.  Hybrid: hooks = ResNet stage 0, stage 1, ViT block 8, 11 (act_postprocess1 / 2
    are identities).  ViT-L/16 (DPT-Large): four ViT hooks, the first two reassembled with transposed convolutions
    (x4, x2).  Index 2 of each act_postprocess (nn.Unflatten in isl-org/DPT) depends on the input size and holds no
    parameters."""

    def __init__(self, backbone="vitb_rn50_384"):
        super().__init__()
        hybrid, d, depth, heads, features, hooks = BACKBONES[backbone]
        self.hooks, self.hybrid, self.features = hooks, hybrid, features
        self.model = VisionTransformerHybrid(embed_dim=d, depth=depth, num_heads=heads, hybrid=hybrid)
        if hybrid:
            self.act_postprocess1 = nn.Sequential(nn.Identity(), nn.Identity(), nn.Identity())
            self.act_postprocess2 = nn.Sequential(nn.Identity(), nn.Identity(), nn.Identity())
        else:
            self.act_postprocess1 = nn.Sequential(ProjectReadout(d), Transpose(1, 2), nn.Identity(), nn.Conv2d(d, features[0], 1, 1, 0),
                                                  nn.ConvTranspose2d(features[0], features[0], 4, 4, 0, bias=True))
            self.act_postprocess2 = nn.Sequential(ProjectReadout(d), Transpose(1, 2), nn.Identity(), nn.Conv2d(d, features[1], 1, 1, 0),
                                                  nn.ConvTranspose2d(features[1], features[1], 2, 2, 0, bias=True))
        self.act_postprocess3 = nn.Sequential(ProjectReadout(d), Transpose(1, 2), nn.Identity(), nn.Conv2d(d, features[2], 1, 1, 0))
        self.act_postprocess4 = nn.Sequential(ProjectReadout(d), Transpose(1, 2), nn.Identity(), nn.Conv2d(d, features[3], 1, 1, 0),
                                              nn.Conv2d(features[3], features[3], 3, 2, 1))


class _Scratch(nn.Module):
    def __init__(self, in_shape=(256, 512, 768, 768), features=256):
        super().__init__()
        self.layer1_rn = nn.Conv2d(in_shape[0], features, 3, 1, 1, bias=False)
        self.layer2_rn = nn.Conv2d(in_shape[1], features, 3, 1, 1, bias=False)
        self.layer3_rn = nn.Conv2d(in_shape[2], features, 3, 1, 1, bias=False)
        self.layer4_rn = nn.Conv2d(in_shape[3], features, 3, 1, 1, bias=False)
        self.refinenet1 = FeatureFusionBlock(features)
        self.refinenet2 = FeatureFusionBlock(features)
        self.refinenet3 = FeatureFusionBlock(features)
        self.refinenet4 = FeatureFusionBlock(features)


class DPT(nn.Module):
    def __init__(self, head, features=256, engine="hip", backbone="vitb_rn50_384"):
        super().__init__()
        self.engine = engine
        self.pretrained = _Pretrained(backbone)
        self.scratch = _Scratch(in_shape=self.pretrained.features, features=features)
        self.scratch.output_conv = head
        self._vit_engine = None
        self._vit_stamp = None
        for m in self.modules():  # the fused channels-last glue kernels follow the engine choice
            if isinstance(m, (GroupNormAct, FeatureFusionBlock, Interpolate, ResidualConvUnit, StdConv2dSame, MaxPool2dSame, ProjectReadout)):
                m.engine = engine

    # -- ViT encoder ---------------------------------------------------------------------------
    def _run_blocks(self, tokens):
        """tokens [B, N, D] -> outputs of the hooked ViT blocks (8, 11 for the hybrid; 5, 11, 17, 23 for ViT-L/16)."""
        vit = self.pretrained.model
        hooks = self.pretrained.hooks[2:] if self.pretrained.hybrid else self.pretrained.hooks
        if self.engine == "torch":
            taps = {}
            x = tokens
            for i, blk in enumerate(vit.blocks):
                x = blk(x)
                if i in hooks:
                    taps[i] = x
            return tuple(taps[h] for h in hooks)
        if self.engine != "hip":
            raise ValueError(f"unknown engine {self.engine!r}")
        if tokens.dtype not in dpt_ops.HALF_TYPES or not tokens.is_cuda:
            # optimize=False in the reference = the network in float32 (dataset_adaptors.py:1396-1399 only halves it when
            # optimising).  The HIP engine computes in the model's 16-bit type; a float32 model is not silently down-cast
            dpt_ops.not_covered("the ViT encoder", f"{tokens.dtype} tokens on {tokens.device}: the model must be .half() / .bfloat16() on the GPU")
        # The engine packs private f32 copies of the biases / LayerNorm parameters (the matrices are used in place): rebuild it when
        # any parameter was replaced or written (load_state_dict, .to(), optimiser step).
        stamp = tuple((p.data_ptr(), p._version, p.dtype) for p in vit.blocks.parameters())
        if self._vit_engine is None or self._vit_stamp != stamp:
            from hive_amd.dpt.vit_engine import VitEngine  # raises if libhive_mi355x.so is missing
            if self._vit_engine is not None:
                self._vit_engine.close()
            self._vit_engine = VitEngine(vit)
            self._vit_stamp = stamp
        return self._vit_engine.forward(tokens, taps=hooks)

    def _conv(self, layer, x):
        """An nn.Conv2d of the reassemble / embedding stages: the hand-written kernel where it applies (1 x 1, 3 x 3 stride 2)."""
        if self.engine == "hip":
            if isinstance(layer, nn.ConvTranspose2d):  # DPT-Large: ConvTranspose2d(k = s) as conv 1 x 1 + scatter
                if not dpt_ops.conv_transpose_eligible(x, layer):
                    dpt_ops.not_covered("transposed convolution", f"kernel {tuple(layer.kernel_size)}, stride {tuple(layer.stride)}, {x.dtype}: kernel == stride only")
                return dpt_ops.conv_transpose(x, layer)
            if isinstance(layer, nn.Conv2d):
                dpt_ops.require_conv(x, layer, "reassemble convolution")
                return dpt_ops.conv2d(x, layer)
            if not isinstance(layer, nn.Identity):
                dpt_ops.not_covered(type(layer).__name__, "no kernel for this layer type")
        return layer(x)

    def forward_backbone(self, x, stages=None):
        p = self.pretrained
        vit = p.model
        b, _, h, w = x.shape
        gh, gw = h // vit.patch_size[1], w // vit.patch_size[0]
        cl = x.is_contiguous(memory_format=torch.channels_last)

        def reassemble(tap, post):
            # readout projection -> [B, D, gh, gw] -> the stage's convolutions (isl-org/DPT forward_vit)
            y = post[0](tap)  # [b, gh gw, D]
            if cl and y.is_contiguous():
                y = y.view(b, gh, gw, -1).permute(0, 3, 1, 2)  # the token matrix IS the channels-last map: no transposing copies
            else:
                y = post[1](y).reshape(b, -1, gh, gw)
                if cl:
                    y = y.contiguous(memory_format=torch.channels_last)
            for layer in post[3:]:
                y = self._conv(layer, y)
            return y

        if p.hybrid:
            feat = vit.patch_embed.backbone.stem(x)
            layer_1 = vit.patch_embed.backbone.stages[0](feat)
            layer_2 = vit.patch_embed.backbone.stages[1](layer_1)
            feat = vit.patch_embed.backbone.stages[2](layer_2)
        else:
            feat = x
        if self.engine == "hip" and not p.hybrid:
            if not dpt_ops.patch_embed_eligible(feat, vit.patch_embed.proj):
                dpt_ops.not_covered("patch embedding", f"{feat.dtype} input of {tuple(feat.shape)}: 16-bit channels-last frames, sides multiples of the patch")
            tokens = dpt_ops.patch_embed(feat, vit.patch_embed.proj)  # ViT-L/16: the 16 x 16 / 16 convolution as rows + GEMM
        else:
            tokens = self._conv(vit.patch_embed.proj, feat).flatten(2).transpose(1, 2)
        tokens = torch.cat((vit.cls_token.expand(b, -1, -1).to(tokens.dtype), tokens), dim=1)
        tokens = tokens + vit.resize_pos_embed(gh, gw).to(tokens.dtype)
        taps = self._run_blocks(tokens)
        if p.hybrid:
            tap3, tap4 = taps
        else:
            tap1, tap2, tap3, tap4 = taps
            layer_1 = reassemble(tap1, p.act_postprocess1)
            layer_2 = reassemble(tap2, p.act_postprocess2)
        layer_3 = reassemble(tap3, p.act_postprocess3)
        layer_4 = reassemble(tap4, p.act_postprocess4)
        if stages is not None:
            stages.update(tokens=tokens, tap_3=tap3, tap_4=tap4, layer_1=layer_1, layer_2=layer_2, layer_3=layer_3, layer_4=layer_4)
        return layer_1, layer_2, layer_3, layer_4

    def forward_decoder(self, x, stages=None):
        """``stages`` (optional dict) receives the intermediate maps by the names isl-org/DPT's forward gives them
        (layer_1..4, path_4..1; plus tokens and the two ViT taps) -- for the per-stage numerics tests."""
        layer_1, layer_2, layer_3, layer_4 = self.forward_backbone(x, stages)
        s = self.scratch

        def rn(conv, x):  # scratch.layerN_rn (3 x 3, no bias) -> (y, relu(y)): the residual unit behind it starts with a ReLU
            if self.engine == "hip":
                if not dpt_ops.conv3x3_eligible(x, conv):
                    dpt_ops.not_covered("scratch.layer_rn", f"{x.dtype} {tuple(x.shape)} -> {conv.out_channels}")
                return dpt_ops.conv3x3(x, conv, also_relu=True)
            return conv(x), None

        l4, l4_relu = rn(s.layer4_rn, layer_4)
        path_4 = s.refinenet4(l4, relu_of_last=l4_relu)
        l3, l3_relu = rn(s.layer3_rn, layer_3)
        path_3 = s.refinenet3(path_4, l3, relu_of_last=l3_relu)
        l2, l2_relu = rn(s.layer2_rn, layer_2)
        path_2 = s.refinenet2(path_3, l2, relu_of_last=l2_relu)
        l1, l1_relu = rn(s.layer1_rn, layer_1)
        path_1 = s.refinenet1(path_2, l1, relu_of_last=l1_relu)
        if stages is not None:
            stages.update(path_4=path_4, path_3=path_3, path_2=path_2, path_1=path_1)
        return path_1

    def forward(self, x):
        return self.scratch.output_conv(self.forward_decoder(x))


class DPTDepthModel(DPT):
    """Same constructor as the reference's ``dpt.models.DPTDepthModel`` (dataset_adaptors.py:1366-1374).

    ``backbone``: ``"vitb_rn50_384"`` (DPT-Hybrid, the one the reference instantiates) or ``"vitl16_384"`` (DPT-Large,
    BASELINE config 4; same decoder, ViT-L/16 encoder with hooks 5 / 11 / 17 / 23 as in isl-org/DPT).
    ``enable_attention_hooks`` must be False.  ``engine`` is this build's addition.
    """

    def __init__(self, path=None, non_negative=True, scale=1.0, shift=0.0, invert=False, backbone="vitb_rn50_384",
                 enable_attention_hooks=False, engine="hip", **kwargs):
        if backbone not in BACKBONES:
            raise NotImplementedError(f"backbone {backbone!r}: implemented are {sorted(BACKBONES)}")
        if enable_attention_hooks:
            raise NotImplementedError("attention hooks (visualisation) are not part of the hot path")
        features = kwargs.get("features", 256)
        head = nn.Sequential(
            nn.Conv2d(features, features // 2, kernel_size=3, stride=1, padding=1),
            Interpolate(scale_factor=2, mode="bilinear", align_corners=True),
            nn.Conv2d(features // 2, 32, kernel_size=3, stride=1, padding=1),
            nn.ReLU(True),
            nn.Conv2d(32, 1, kernel_size=1, stride=1, padding=0),
            nn.ReLU(True) if non_negative else nn.Identity(),
            nn.Identity(),
        )
        super().__init__(head, features=features, engine=engine, backbone=backbone)
        self.scale, self.shift, self.invert = scale, shift, invert
        if path is not None:
            self.load(path)

    def load(self, path):
        parameters = torch.load(path, map_location=torch.device("cpu"))
        if "optimizer" in parameters:
            parameters = parameters["model"]
        # the published checkpoints also carry the (unused) classification head / final norm of the ViT
        missing, unexpected = self.load_state_dict(parameters, strict=False)
        missing = [k for k in missing if not k.endswith("_std_weight")]
        self.load_report = (list(missing), list(unexpected))  # (missing, unexpected) keys of the last load
        self._vit_engine = None  # its packed copies of the encoder weights are stale now
        if missing:
            raise RuntimeError(f"checkpoint {path} lacks parameters: {missing[:8]}{'...' if len(missing) > 8 else ''}")

    def native(self):
        """The network as one C-ABI object (``hive_dpt_create``, csrc/dpt_net.hip): rebuilt when a parameter was replaced or
        written.  16-bit (bfloat16 / float16) models on the GPU only."""
        from hive_amd.dpt.native import NativeDPT, parameter_stamp
        cur = getattr(self, "_native", None)
        if cur is None or cur.stamp != parameter_stamp(self):
            if cur is not None:
                cur.close()
            self._native = cur = NativeDPT(self)
        return cur

    def forward_frames(self, frames_u8, max_depth=None, net_size=None):
        """uint8 frames [B, H, W, 3] in HBM -> (depth, depth_mm, depth_m) through ``hive_dpt_forward_frames``: pre-processing, the whole
        network and the depth hand-off in one C-ABI call.  ``net_size=(net_h, net_w)`` (multiples of 32; default: the frame size, which must then be
        one): frames of any other size go through the reference's bicubic / nearest resizes on the device."""
        return self.native().forward(frames_u8, max_depth=max_depth, net_size=net_size)

    def forward_head_features(self, x):
        """Everything up to (and including) the ReLU before the last 1x1 convolution: [B, 32, h, w]."""
        head = self.scratch.output_conv
        return head[3](head[2](head[1](head[0](self.forward_decoder(x)))))

    def forward(self, x, handoff=None, stages=None):
        """[B, 3, h, w] -> depth [B, h, w] in **float32**.  ``stages``: see ``forward_decoder`` (adds ``head_in``, the
        128-channel map behind ``output_conv[0]``).

        The reference returns the network's working precision (fp16 on a GPU); the last 1x1 convolution and
        the inversion ``1 / (scale * x + shift)`` run in float32 here because 16 bits cannot resolve metric depth
        (bfloat16's 8 significant bits: 3 cm steps at 7 m; float16's 11: 4 mm).  ``handoff=(max_depth,)`` additionally applies the
        reference's depth hand-off on the device -- uint16 millimetres (dataset_adaptors.py:1432-1433), read
        back as float32 metres with ``> max_depth -> 0`` (io.py:1032-1039) -- and returns (depth, depth_mm, depth_m).
        """
        head = self.scratch.output_conv
        conv = head[4]
        non_negative = isinstance(head[5], nn.ReLU)
        if self.engine == "hip":
            from hive_amd import _lib
            if not (x.is_cuda and x.dtype in dpt_ops.HALF_TYPES):
                dpt_ops.not_covered("DPTDepthModel.forward", f"{x.dtype} input on {x.device}: a .half() / .bfloat16() model on the GPU takes 16-bit channels-last input")
            ctx = _lib.default_context(x.device.index or 0)
            pre = head[2]
            first = head[0]
            if not (isinstance(head[1], Interpolate) and head[1].scale_factor == 2 and head[1].mode == "bilinear" and head[1].align_corners
                    and isinstance(head[3], nn.ReLU) and tuple(pre.weight.shape) == (32, 128, 3, 3) and pre.stride == (1, 1) and pre.padding == (1, 1)):
                dpt_ops.not_covered("the depth head", "the fused head kernel is built for Interpolate(x2) -> Conv3x3(128 -> 32) -> ReLU -> Conv1x1(32 -> 1)")
            key = (conv.weight.data_ptr(), conv.weight._version, conv.bias._version, pre.bias._version, pre.weight._version, pre.weight.dtype)
            if getattr(self, "_tail_host", (None,))[0] != key:  # one D2H per set of weights, not per forward
                self._tail_host = (key, conv.weight.detach().float().reshape(-1).cpu().numpy(), float(conv.bias.detach().float().item()),
                                   pre.bias.detach().float().cpu().numpy(),
                                   pre.weight.detach().permute(2, 3, 0, 1).contiguous())  # [ky][kx][out][in] for the fused head
            weight, bias, pre_bias, w3 = self._tail_host[1:5]
            # output_conv[0] with its bias in the epilogue; then Interpolate + conv 128 -> 32 + ReLU + conv 32 -> 1 + inversion + hand-off:
            # one HIP kernel (csrc/dpt_head.hip)
            path_1 = self.forward_decoder(x, stages)
            if not dpt_ops.conv3x3_eligible(path_1, first):
                dpt_ops.not_covered("scratch.output_conv[0]", f"{path_1.dtype} {tuple(path_1.shape)} -> {first.out_channels}")
            lo = dpt_ops.conv3x3(path_1, first)
            if stages is not None:
                stages["head_in"] = lo
            b, c, h, w = lo.shape
            depth = torch.empty((b, 2 * h, 2 * w), dtype=torch.float32, device=lo.device)
            mm = torch.empty((b, 2 * h, 2 * w), dtype=torch.int16, device=lo.device) if handoff else None
            m = torch.empty((b, 2 * h, 2 * w), dtype=torch.float32, device=lo.device) if handoff else None
            ctx.check(ctx.lib.hive_dpt_head_fused(
                ctx.handle, lo.data_ptr(), None, dpt_ops._code(lo.dtype), b, h, w, c, 32, w3.data_ptr(), pre_bias.ctypes.data, weight.ctypes.data, bias,
                int(non_negative), int(bool(self.invert)), float(self.scale), float(self.shift), depth.data_ptr(), 1.0 / 1000.0,
                float(handoff[0]) if handoff else 0.0, _lib.ptr(mm), _lib.ptr(m)))
            return (depth, mm, m) if handoff else depth
        head_in = head[0](self.forward_decoder(x, stages))
        if stages is not None:
            stages["head_in"] = head_in
        feat = head[3](head[2](head[1](head_in)))
        out = F.conv2d(feat.float(), conv.weight.float(), conv.bias.float()).squeeze(dim=1)
        if non_negative:
            out = F.relu(out)
        if self.invert:
            out = 1.0 / torch.clamp(self.scale * out + self.shift, min=1e-8)  # depth[depth < 1e-8] = 1e-8
        if handoff:
            mm = (out * 1000.0).clamp(0, 65535).to(torch.int32)
            m = mm.float() * (1.0 / 1000.0)
            m = torch.where(m > handoff[0], torch.zeros_like(m), m)
            return out, mm.to(torch.int16), m
        return out


def count_flops(height=480, width=640):
    """Algorithmic FLOPs (2 x MACs) of one DPT-Hybrid forward at height x width, by component."""
    def conv(cin, cout, k, h, w):
        return 2 * cin * cout * k * k * h * w
    h4, w4 = height // 4, width // 4
    flops = {"stem": conv(3, 64, 7, height // 2, width // 2)}
    rn = 0
    prev = 64
    for i, (depth, chs) in enumerate(zip((3, 4, 9), (256, 512, 1024))):
        s = 2 ** i
        h, w = h4 // s, w4 // s
        mid = chs // 4
        for b in range(depth):
            cin = prev if b == 0 else chs
            hin, win = (h * (2 if (i > 0 and b == 0) else 1), w * (2 if (i > 0 and b == 0) else 1))
            rn += conv(cin, mid, 1, hin, win) + conv(mid, mid, 3, h, w) + conv(mid, chs, 1, h, w)
            if b == 0:
                rn += conv(cin, chs, 1, h, w)
        prev = chs
    flops["resnet_stages"] = rn
    gh, gw = height // 16, width // 16
    n, d = gh * gw + 1, 768
    flops["patch_proj"] = conv(1024, d, 1, gh, gw)
    flops["vit_gemm"] = 12 * (2 * n * d * 3 * d + 2 * n * d * d + 2 * 2 * n * d * 4 * d)
    flops["vit_attention"] = 12 * (2 * 2 * n * n * d)
    flops["readout"] = 2 * (2 * (n - 1) * 2 * d * d)
    flops["reassemble"] = conv(d, 768, 1, gh, gw) * 2 + conv(768, 768, 3, gh // 2, gw // 2)
    f = 256
    flops["layer_rn"] = conv(256, f, 3, h4, w4) + conv(512, f, 3, h4 // 2, w4 // 2) + conv(768, f, 3, gh, gw) + \
        conv(768, f, 3, gh // 2, gw // 2)
    ref = 0
    for i, (h, w) in enumerate(((gh // 2, gw // 2), (gh, gw), (h4 // 2, w4 // 2), (h4, w4))):
        n_rcu = 1 if i == 0 else 2
        ref += n_rcu * 2 * conv(f, f, 3, h, w) + conv(f, f, 1, 2 * h, 2 * w)
    flops["refinenets"] = ref
    flops["head"] = conv(f, f // 2, 3, 2 * h4, 2 * w4) + conv(f // 2, 32, 3, height, width) + conv(32, 1, 1, height, width)
    flops["total"] = sum(flops.values())
    return flops
