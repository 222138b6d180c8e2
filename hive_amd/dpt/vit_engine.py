"""Host side of the HIP ViT encoder: packs the weights of a ``VisionTransformerHybrid`` once
(matrices in the model's 16-bit type -- bfloat16 or float16 --, f32 biases / LayerNorm affine, contiguous in HBM) and runs all
blocks through ``hive_vit_forward`` of the C ABI.  There is no fallback: construction raises if the library or the device is
missing, or if the model is not in a 16-bit type (no silent down-cast of a float32 model)."""
import ctypes

import torch

from hive_amd import _lib


class _BlockWeights(ctypes.Structure):
    _fields_ = [(name, ctypes.c_void_p) for name in
                ("ln1_g", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "ln2_g", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")]


class VitEngine:
    def __init__(self, vit, ctx=None):
        """:param vit: a ``hive_amd.dpt.models.VisionTransformerHybrid`` whose parameters live on a GPU."""
        dev = vit.cls_token.device
        if dev.type != "cuda":
            raise _lib.HiveError(_lib.ERR_DEVICE, "the HIP ViT engine needs the model on an MI355X (model.cuda()); no CPU fallback")
        self.ctx = ctx or _lib.default_context(dev.index or 0)
        self.dim, self.heads = vit.embed_dim, vit.num_heads
        self.dtype = vit.blocks[0].attn.qkv.weight.dtype
        self.code = _lib.dtype_code(self.dtype)  # raises for float32: the engine does not down-cast
        self._keep = []  # packed tensors must outlive the native handle

        def mat(p):
            if p.dtype != self.dtype:
                raise _lib.HiveError(_lib.ERR_INVALID, f"ViT matrices of mixed types ({p.dtype} beside {self.dtype})")
            t = p.detach().to(device=dev).contiguous()
            self._keep.append(t)
            return t.data_ptr()

        def vec(p):
            t = p.detach().to(device=dev, dtype=torch.float32).contiguous()
            self._keep.append(t)
            return t.data_ptr()

        blocks = (_BlockWeights * len(vit.blocks))()
        for i, blk in enumerate(vit.blocks):
            blocks[i] = _BlockWeights(vec(blk.norm1.weight), vec(blk.norm1.bias), mat(blk.attn.qkv.weight), vec(blk.attn.qkv.bias),
                                      mat(blk.attn.proj.weight), vec(blk.attn.proj.bias), vec(blk.norm2.weight), vec(blk.norm2.bias),
                                      mat(blk.mlp.fc1.weight), vec(blk.mlp.fc1.bias), mat(blk.mlp.fc2.weight), vec(blk.mlp.fc2.bias))
        self.mlp = vit.blocks[0].mlp.fc1.out_features
        self.eps = float(vit.blocks[0].norm1.eps)
        handle = ctypes.c_void_p()
        self.ctx.check(self.ctx.lib.hive_vit_create(self.ctx.handle, self.code, len(vit.blocks), self.dim, self.heads, self.mlp, self.eps,
                                                    ctypes.cast(blocks, ctypes.c_void_p), ctypes.byref(handle)))
        self.handle = handle

    def forward(self, tokens, taps):
        """tokens [B, N, D] (the engine's 16-bit type, on the GPU) -> tuple of block outputs at ``taps``."""
        B, N, D = tokens.shape
        assert D == self.dim
        if tokens.dtype != self.dtype:
            raise _lib.HiveError(_lib.ERR_INVALID, f"tokens are {tokens.dtype}, the engine was built for {self.dtype}: no silent cast")
        self.ctx.follow_torch_stream()  # the tokens were produced on torch's current stream: queue behind them
        x = tokens.contiguous()
        outs = [torch.empty_like(x) for _ in taps]
        tap_idx = (ctypes.c_int * len(taps))(*[int(t) for t in taps])
        tap_ptr = (ctypes.c_void_p * len(taps))(*[o.data_ptr() for o in outs])
        self.ctx.check(self.ctx.lib.hive_vit_forward(self.handle, x.data_ptr(), B, N, ctypes.cast(tap_idx, ctypes.c_void_p), len(taps),
                                                     ctypes.cast(tap_ptr, ctypes.c_void_p)))
        return tuple(outs)

    def weights_modified(self):
        """The packed weight tensors were overwritten in place: re-fold what ``hive_vit_create`` captured (``hive_vit_weights_modified``)."""
        self.ctx.follow_torch_stream()
        self.ctx.check(self.ctx.lib.hive_vit_weights_modified(self.handle))

    def close(self):
        if getattr(self, "handle", None) and _lib.alive():
            self.ctx.lib.hive_vit_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
