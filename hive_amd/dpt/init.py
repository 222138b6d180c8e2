"""Seeded, non-degenerate random weights for the DPT networks (no checkpoint can be fetched in the build environment).

PyTorch's default initialisation leaves the depth head at ~0 behind its ReLUs and the RefineNet paths with a tiny
dynamic range: the predicted depth is a constant (~7.25 m), which makes a degenerate TSDF scene (free space only) and a
numerics comparison that says little.  ``seeded_init`` gives every stage a usable range: variance-preserving (He)
convolution / linear weights, non-trivial norm affine parameters, and a depth head whose output spans the NYU
checkpoint's working range (inverse-depth units of the reference's scale / shift,
/root/reference/hive/dataset_adaptors.py:1366-1374), i.e. depths between roughly 0.5 m and 7 m.  Used by ``bench.py``
(synthetic workload), ``__graft_entry__.smoke()`` and the numerics tests.

Deterministic for a given torch build (CPU generator).
"""
import math

import torch
import torch.nn as nn


@torch.no_grad()
def seeded_init(model, seed=1234):
    g = torch.Generator(device="cpu").manual_seed(seed)

    def normal_(p, std, mean=0.0):
        p.copy_((torch.randn(p.shape, generator=g, dtype=torch.float32) * std + mean).to(p.dtype))

    for name, m in model.named_modules():
        if isinstance(m, nn.ConvTranspose2d):  # stride == kernel: every output pixel sums in_channels products
            normal_(m.weight, math.sqrt(2.0 / m.in_channels))
            normal_(m.bias, 0.05)
        elif isinstance(m, nn.Conv2d):
            fan_in = m.in_channels * m.kernel_size[0] * m.kernel_size[1] // m.groups
            normal_(m.weight, math.sqrt(2.0 / fan_in))
            if m.bias is not None:
                normal_(m.bias, 0.05)
        elif isinstance(m, nn.Linear):
            normal_(m.weight, math.sqrt(1.0 / m.in_features))
            if m.bias is not None:
                normal_(m.bias, 0.05)
        elif isinstance(m, (nn.GroupNorm, nn.LayerNorm)):
            if name.endswith("norm3"):
                # last norm of a residual bottleneck: a small gain, as trained ResNets have (BiT zero-initialises it).
                # With a gain of 1 every block re-amplifies the rounding noise of the stream it adds to, and a bf16
                # network drifts 3 % per block from its float32 twin -- a property of the weights, not of the kernels.
                normal_(m.weight, 0.05, 0.25)
            else:
                normal_(m.weight, 0.1, 1.0)
            normal_(m.bias, 0.1)
    vit = model.pretrained.model
    normal_(vit.pos_embed, 0.2)
    normal_(vit.cls_token, 0.2)
    vit.patch_embed.proj.weight.mul_(0.25)  # tokens of O(1), so that the 12 blocks (each adds O(0.5)) shape the taps
    # residual branches of the transformer: keep the stream from growing with depth
    for blk in vit.blocks:
        blk.attn.proj.weight.mul_(0.5)
        blk.mlp.fc2.weight.mul_(0.5)
    # residual conv units add their input back: halve the second convolution so four fusion stages stay O(1)
    for name, m in model.named_modules():
        if name.endswith("conv2") and "resConfUnit" in name:
            m.weight.mul_(0.5)
        if name.endswith("out_conv"):
            m.weight.mul_(0.4)
    # depth head: 32 non-negative features -> inverse depth.  Reference scale / shift: depth = 1 / (0.000305 x + 0.1378);
    # x in [0, ~6000] spans 7.26 m .. 0.5 m.  Positive-mean weights and a bias put x in that range.
    head = model.scratch.output_conv
    head[0].weight.mul_(0.3)
    normal_(head[4].weight, 150.0, 60.0)
    head[4].bias.fill_(200.0)
    return model
