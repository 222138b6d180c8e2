#!/usr/bin/env python3
"""Golden activations for DPT-Hybrid from an INDEPENDENT implementation of the published architecture:
``transformers.DPTForDepthEstimation(DPTConfig(is_hybrid=True))`` (HuggingFace's port of isl-org/DPT +
timm's ResNetV2 / ViT hybrid), fed with the same seeded random state dict as ``hive_amd.dpt.models``.

    cd /root/repo && python tests/golden/make_dpt_golden.py        (build container only: needs `transformers`)

The reference's own network (third_party/dpt, AnthonyDickson/DPT) is absent from the snapshot and no
checkpoint is available, so this does NOT pin parity with the reference -- it catches a wrong
readout, position-embedding resize, hook index, padding rule or fusion order in this build's restatement,
which a self-comparison (HIP vs this build's own PyTorch formulation) cannot.

Stored (data only): the network input, the inverse-depth output and a few intermediate maps of the HF
model, plus the checksum of the seeded state dict the tests regenerate (``tests/dpt_weights.py``).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from dpt_weights import seeded_init, seeded_input, state_checksum  # noqa: E402
from hive_amd.dpt.models import DPTDepthModel  # noqa: E402


def to_hf_state(sd, depth=12, dim=768, hybrid=True):
    """Parameter names of the published isl-org/DPT checkpoints (hybrid / large) -> HuggingFace DPT names."""
    out = {}

    def put(dst, src):
        out[dst] = sd[src].clone()

    e = "dpt.embeddings."
    put(e + "cls_token", "pretrained.model.cls_token")
    put(e + "position_embeddings", "pretrained.model.pos_embed")
    for p in ("weight", "bias"):
        put(e + (f"projection.{p}" if hybrid else f"patch_embeddings.projection.{p}"), f"pretrained.model.patch_embed.proj.{p}")
        put(f"dpt.layernorm.{p}", f"pretrained.model.norm.{p}")
    if hybrid:
        bb_src, bb_dst = "pretrained.model.patch_embed.backbone.", e + "backbone.bit."
        put(bb_dst + "embedder.convolution.weight", bb_src + "stem.conv.weight")
        put(bb_dst + "embedder.norm.weight", bb_src + "stem.norm.weight")
        put(bb_dst + "embedder.norm.bias", bb_src + "stem.norm.bias")
        for k in sd:
            if k.startswith(bb_src + "stages.") and not k.endswith("_std_weight"):
                rest = k[len(bb_src):].replace(".blocks.", ".layers.")
                out[bb_dst + "encoder." + rest] = sd[k].clone()
    else:
        for p in ("weight", "bias"):
            for n in (1, 2):
                put(f"neck.reassemble_stage.readout_projects.{n - 1}.0.{p}", f"pretrained.act_postprocess{n}.0.project.0.{p}")
                put(f"neck.reassemble_stage.layers.{n - 1}.projection.{p}", f"pretrained.act_postprocess{n}.3.{p}")
                put(f"neck.reassemble_stage.layers.{n - 1}.resize.{p}", f"pretrained.act_postprocess{n}.4.{p}")
    for i in range(depth):
        src, dst = f"pretrained.model.blocks.{i}.", f"dpt.encoder.layer.{i}."
        qkv_w, qkv_b = sd[src + "attn.qkv.weight"], sd[src + "attn.qkv.bias"]
        for j, name in enumerate(("query", "key", "value")):
            out[dst + f"attention.attention.{name}.weight"] = qkv_w[j * dim:(j + 1) * dim].clone()
            out[dst + f"attention.attention.{name}.bias"] = qkv_b[j * dim:(j + 1) * dim].clone()
        for p in ("weight", "bias"):
            put(dst + f"attention.output.dense.{p}", src + f"attn.proj.{p}")
            put(dst + f"intermediate.dense.{p}", src + f"mlp.fc1.{p}")
            put(dst + f"output.dense.{p}", src + f"mlp.fc2.{p}")
            put(dst + f"layernorm_before.{p}", src + f"norm1.{p}")
            put(dst + f"layernorm_after.{p}", src + f"norm2.{p}")
    for p in ("weight", "bias"):
        put(f"neck.reassemble_stage.readout_projects.2.0.{p}", f"pretrained.act_postprocess3.0.project.0.{p}")
        put(f"neck.reassemble_stage.readout_projects.3.0.{p}", f"pretrained.act_postprocess4.0.project.0.{p}")
        put(f"neck.reassemble_stage.layers.2.projection.{p}", f"pretrained.act_postprocess3.3.{p}")
        put(f"neck.reassemble_stage.layers.3.projection.{p}", f"pretrained.act_postprocess4.3.{p}")
        put(f"neck.reassemble_stage.layers.3.resize.{p}", f"pretrained.act_postprocess4.4.{p}")
        for j in (0, 2, 4):
            put(f"head.head.{j}.{p}", f"scratch.output_conv.{j}.{p}")
    for n in range(1, 5):
        put(f"neck.convs.{n - 1}.weight", f"scratch.layer{n}_rn.weight")
        src, dst = f"scratch.refinenet{n}.", f"neck.fusion_stage.layers.{4 - n}."
        for p in ("weight", "bias"):
            put(dst + f"projection.{p}", src + f"out_conv.{p}")
            for u in (1, 2):
                for c in (1, 2):
                    put(dst + f"residual_layer{u}.convolution{c}.{p}", src + f"resConfUnit{u}.conv{c}.{p}")
    return out


def generate(backbone, out_name, H, W, B, compact=False):
    from transformers import DPTConfig, DPTForDepthEstimation
    import functools

    hybrid = backbone == "vitb_rn50_384"
    mine = DPTDepthModel(path=None, scale=1.0, shift=0.0, invert=False, engine="torch", backbone=backbone).eval()
    seeded_init(mine, seed=1234)
    x = seeded_input(B, H, W, seed=99).half().float()  # stored as float16: make that exact

    common = dict(layer_norm_eps=1e-6,  # timm's ViT LayerNorm eps (HF's config default is 1e-12)
                  image_size=384, readout_type="project")
    if hybrid:
        cfg = DPTConfig(is_hybrid=True, neck_hidden_sizes=[256, 512, 768, 768], reassemble_factors=[1, 1, 1, 0.5],
                        backbone_out_indices=[2, 5, 8, 11], **common)
        depth, dim = 12, 768
    else:
        cfg = DPTConfig(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                        backbone_out_indices=[5, 11, 17, 23], neck_hidden_sizes=[256, 512, 1024, 1024],
                        reassemble_factors=[4, 2, 1, 0.5], **common)
        depth, dim = 24, 1024
    hf = DPTForDepthEstimation(cfg).eval()
    missing, unexpected = hf.load_state_dict(to_hf_state(mine.state_dict(), depth, dim, hybrid), strict=False)
    assert not unexpected, unexpected
    assert not missing, missing

    captured = {}

    def grab(name):
        def hook(_m, _inp, out):
            captured[name] = out[0] if isinstance(out, (tuple, list)) else out
        return hook

    hooks = (8, 11) if hybrid else (17, 23)
    hf.dpt.encoder.layer[hooks[0]].register_forward_hook(grab("tap_3"))
    hf.dpt.encoder.layer[hooks[1]].register_forward_hook(grab("tap_4"))
    hf.neck.register_forward_hook(lambda _m, _i, out: captured.update(neck=out))
    hf.head.head[0].register_forward_hook(grab("head_in"))
    # HF's DPTModel does not pass `interpolate_pos_encoding` on to its embeddings; bind it there (harness side) so that
    # the 24 x 24 training grid is resized to this input's token grid by HF's own `_resize_pos_embed` ...
    if "interpolate_pos_encoding" in __import__("inspect").signature(hf.dpt.embeddings.forward).parameters:
        hf.dpt.embeddings.forward = functools.partial(hf.dpt.embeddings.forward, interpolate_pos_encoding=True)
    # ... and its neck assumes a square token grid when it un-flattens the tokens: hand it the grid shape
    neck_forward = hf.neck.forward
    hf.neck.forward = lambda hidden_states, patch_height=None, patch_width=None: neck_forward(hidden_states, H // 16, W // 16)
    with torch.no_grad():
        out = hf(pixel_values=x)
    inv = out.predicted_depth
    paths = captured["neck"]  # fusion-stage outputs: [path_4, path_3, path_2, path_1]
    if compact:
        # the benchmark's frame size: the maps are megabytes, so the fixture keeps sub-sampled views (every 4th pixel / 16th token) and the
        # input is regenerated by the tests from its seed (its checksum is stored)
        import hashlib
        np.savez_compressed(
            os.path.join(HERE, out_name),
            x_sha256=np.array(hashlib.sha256(x.numpy().astype(np.float16).tobytes()).hexdigest()), x_seed=np.int64(99), shape=np.array([B, H, W]),
            inv_depth_s4=inv[:, ::4, ::4].numpy().astype(np.float32), tap_3_mean=captured["tap_3"].mean(dim=2).numpy().astype(np.float32),
            tap_4_s16=captured["tap_4"][:, ::16].numpy().astype(np.float16), path_4=paths[0].numpy().astype(np.float16),
            path_1_mean_s2=paths[-1].mean(dim=1)[:, ::2, ::2].numpy().astype(np.float32),
            head_in_mean_s2=captured["head_in"].mean(dim=1)[:, ::2, ::2].numpy().astype(np.float32),
            backbone=np.array(backbone), seed=np.int64(1234), state_sha256=np.array(state_checksum(mine)),
            transformers_version=np.array(__import__("transformers").__version__), torch_version=np.array(torch.__version__))
    else:
        _save_full(out_name, x, inv, captured, paths, backbone, mine)
    # the generator checks itself: this build's PyTorch formulation against the independent one, float32
    _self_check(backbone, mine, x, inv, captured, paths)


def _save_full(out_name, x, inv, captured, paths, backbone, mine):
    np.savez_compressed(
        os.path.join(HERE, out_name),
        x=x.numpy().astype(np.float16),  # exactly representable: the tests feed x.half().float()
        inv_depth=inv.numpy().astype(np.float32),
        tap_3_mean=captured["tap_3"].mean(dim=2).numpy().astype(np.float32), tap_4=captured["tap_4"].numpy().astype(np.float16),
        path_4=paths[0].numpy().astype(np.float16), path_1_mean=paths[-1].mean(dim=1).numpy().astype(np.float32),
        head_in_mean=captured["head_in"].mean(dim=1).numpy().astype(np.float32),
        backbone=np.array(backbone), seed=np.int64(1234), state_sha256=np.array(state_checksum(mine)),
        transformers_version=np.array(__import__("transformers").__version__), torch_version=np.array(torch.__version__))


def _self_check(backbone, mine, x, inv, captured, paths):
    with torch.no_grad():
        st = {}
        mine_inv = mine(x, stages=st)
    print(backbone, "inverse depth range", float(inv.min()), float(inv.max()))
    for name, a, b in (("tap_3", st["tap_3"], captured["tap_3"]), ("tap_4", st["tap_4"], captured["tap_4"]),
                       ("path_4", st["path_4"], paths[0]), ("path_1", st["path_1"], paths[-1]),
                       ("head_in", st["head_in"], captured["head_in"]), ("inv_depth", mine_inv, inv)):
        rel = float((a - b).norm() / b.norm())
        print(f"  {name:10s} rel Frobenius error {rel:.3e}")
        assert rel < 1e-3, "this build's restatement disagrees with the independent implementation"


def main():
    # 6 x 10 token grids: exercise the non-square position-embedding resize
    generate("vitb_rn50_384", "dpt_hybrid_hf.npz", 96, 160, 2)
    generate("vitl16_384", "dpt_large_hf.npz", 96, 160, 1)
    # the benchmark's frame size (30 x 40 token grid, 1,201 tokens): pins the float32 formulation the 480 x 640 GPU tests compare against
    generate("vitb_rn50_384", "dpt_hybrid_hf_480x640.npz", 480, 640, 1, compact=True)


if __name__ == "__main__":
    main()
