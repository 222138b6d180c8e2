"""Host side of the C-ABI network object (``hive_dpt_create / forward / destroy``, csrc/dpt_net.hip): builds the tensor table
from a ``DPTDepthModel`` (DPT-Hybrid or DPT-Large) -- standardised ResNet weights, channels-last convolution weights, matrices in the
model's 16-bit type (bfloat16, or float16 as the reference's ``model.half()``), f32 biases / LayerNorm parameters, as
``include/hive_mi355x.h`` documents -- and runs whole batches of uint8 frames through it:
frames in HBM -> depth maps in HBM without a PyTorch operator in between.  No fallback: construction raises if the library or
the device is missing."""
import ctypes

import torch
import torch.nn as nn

from hive_amd import _lib


class _Tensor(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("data", ctypes.c_void_p)]


class _Config(ctypes.Structure):
    _fields_ = [("backbone", ctypes.c_int), ("scale", ctypes.c_float), ("shift", ctypes.c_float), ("invert", ctypes.c_int),
                ("non_negative", ctypes.c_int), ("gn_eps", ctypes.c_float), ("ln_eps", ctypes.c_float),
                ("head_b3", ctypes.c_float * 32), ("head_w1", ctypes.c_float * 32), ("head_b1", ctypes.c_float), ("dtype", ctypes.c_int)]


def parameter_stamp(model):
    return tuple((p.data_ptr(), p._version, p.dtype) for p in model.parameters())


class NativeDPT:
    def __init__(self, model, ctx=None):
        """:param model: a ``hive_amd.dpt.models.DPTDepthModel`` (``vitb_rn50_384`` or ``vitl16_384``) whose parameters live on an MI355X."""
        from hive_amd.dpt.models import StdConv2dSame
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise _lib.HiveError(_lib.ERR_DEVICE, "the native DPT network needs the model on an MI355X (model.cuda()); no CPU fallback")
        self.ctx = ctx or _lib.default_context(dev.index or 0)
        self.model = model
        self.dtype = next(model.parameters()).dtype
        self.code = _lib.dtype_code(self.dtype)  # float32 (optimize=False) raises: the network object computes in the model's 16-bit type
        if any(p.dtype != self.dtype for p in model.parameters()):
            raise _lib.HiveError(_lib.ERR_INVALID, "the model's parameters are of mixed types: convert it as a whole (.half() / .bfloat16())")
        self.stamp = parameter_stamp(model)
        self._keep, names, ptrs = [], [], []

        def add(name, t):
            t = t.detach().contiguous()
            self._keep.append(t)
            names.append(name.encode())
            ptrs.append(t.data_ptr())

        bf, f32 = self.dtype, torch.float32  # bf: the model's 16-bit type
        conv_w = lambda w: w.detach().to(device=dev, dtype=bf).permute(0, 2, 3, 1)  # [C_out][ky][kx][C_in]
        mods = dict(model.named_modules())
        for name, p in model.named_parameters():
            mod = mods[name.rsplit(".", 1)[0]]
            leaf = name.rsplit(".", 1)[1]
            if isinstance(mod, StdConv2dSame):
                w = mod.standardized_weight().detach().to(device=dev, dtype=bf)
                if tuple(mod.kernel_size) == (7, 7):  # stem: (ky, (kx, c)) with a kernel row padded from 21 to 32
                    s = torch.zeros((64, 7, 32), dtype=bf, device=dev)
                    s[:, :, :21] = w.permute(0, 2, 3, 1).reshape(64, 7, 21)
                    add(name, s)
                else:
                    add(name, conv_w(w))
            elif isinstance(mod, nn.Conv2d) and leaf == "weight":
                if name == "scratch.output_conv.2.weight":
                    add(name, p.detach().to(device=dev, dtype=bf).permute(2, 3, 0, 1))  # [ky][kx][32][128] for the fused head
                elif name != "scratch.output_conv.4.weight":
                    add(name, conv_w(p))
            elif isinstance(mod, nn.ConvTranspose2d) and leaf == "weight":  # DPT-Large reassemble: rows (dy, dx, co) of the 1 x 1 form
                k, co = mod.kernel_size[0], mod.out_channels
                add(name + ".rows", p.detach().to(device=dev, dtype=bf).permute(2, 3, 1, 0).reshape(k * k * co, mod.in_channels))
            elif isinstance(mod, nn.Linear) and leaf == "weight":
                add(name, p.to(device=dev, dtype=bf))
            elif isinstance(mod, (nn.Linear, nn.LayerNorm)):  # GEMM biases and LayerNorm affine parameters: float32
                add(name, p.to(device=dev, dtype=f32))
            elif name == "pretrained.model.cls_token":
                add(name, p.to(device=dev, dtype=bf).reshape(-1))
            elif name != "pretrained.model.pos_embed":  # conv biases, GroupNorm affine: the tensor dtype
                add(name, p.to(device=dev, dtype=bf))
        head = model.scratch.output_conv
        if not model.pretrained.hybrid:  # the 16 x 16 / 16 patch embedding runs as a GEMM: float32 bias
            add("pretrained.model.patch_embed.proj.bias.f32", model.pretrained.model.patch_embed.proj.bias.to(device=dev, dtype=f32))
        cfg = _Config(0 if model.pretrained.hybrid else 1, float(model.scale), float(model.shift), int(bool(model.invert)), int(isinstance(head[5], nn.ReLU)), 1e-5,
                      float(model.pretrained.model.blocks[0].norm1.eps))
        cfg.dtype = self.code
        b3, w1 = head[2].bias.detach().float().cpu(), head[4].weight.detach().float().reshape(-1).cpu()
        for i in range(32):
            cfg.head_b3[i], cfg.head_w1[i] = float(b3[i]), float(w1[i])
        cfg.head_b1 = float(head[4].bias.detach().float().item())
        table = (_Tensor * len(names))(*[_Tensor(n, p) for n, p in zip(names, ptrs)])
        self._names = names
        handle = ctypes.c_void_p()
        self.ctx.check(self.ctx.lib.hive_dpt_create(self.ctx.handle, ctypes.byref(cfg), ctypes.cast(table, ctypes.c_void_p), len(names),
                                                    ctypes.byref(handle)))
        self.handle = handle
        self._pos = {}

    def _pos_embed(self, gh, gw):
        key = (gh, gw)
        if key not in self._pos:
            with torch.no_grad():
                pos = self.model.pretrained.model.resize_pos_embed(gh, gw)  # dpt `_resize_pos_embed` (bilinear, float32 inside)
            self._pos[key] = pos.detach().to(self.dtype).reshape(gh * gw + 1, -1).contiguous()
        return self._pos[key]

    @torch.no_grad()
    def forward(self, frames_u8, max_depth=None, want_depth=True, net_size=None):
        """frames_u8 [B, H, W, 3] uint8 on the GPU -> (depth f32 [B, H, W], depth_mm int16-viewed-uint16 or None, depth_m or None).
        ``net_size=(net_h, net_w)``: the size the network runs at (multiples of 32) when it is not the frames' -- the frames enter through the reference's
        bicubic resize and the depth maps come back through its nearest-neighbour resize, both on the device inside the one C-ABI call
        (``hive_dpt_forward_frames``; /root/reference/hive/dataset_adaptors.py:1376-1389, 1421-1426)."""
        assert frames_u8.dtype == torch.uint8 and frames_u8.is_cuda and frames_u8.dim() == 4 and frames_u8.shape[-1] == 3
        frames_u8 = frames_u8.contiguous()
        b, h, w, _ = frames_u8.shape
        net_h, net_w = (h, w) if net_size is None else (int(net_size[0]), int(net_size[1]))
        self.ctx.follow_torch_stream()
        dev = frames_u8.device
        depth = torch.empty((b, h, w), dtype=torch.float32, device=dev) if want_depth else None
        mm = torch.empty((b, h, w), dtype=torch.int16, device=dev) if max_depth is not None else None
        m = torch.empty((b, h, w), dtype=torch.float32, device=dev) if max_depth is not None else None
        pos = self._pos_embed(net_h // 16, net_w // 16)
        self.ctx.check(self.ctx.lib.hive_dpt_forward_frames(self.handle, frames_u8.data_ptr(), b, h, w, net_h, net_w, pos.data_ptr(), _lib.ptr(depth),
                                                            float(max_depth or 0.0), _lib.ptr(mm), _lib.ptr(m)))
        return depth, mm, m

    def weights_modified(self):
        """The tensors of the table were overwritten IN PLACE (same pointers): refresh what the object derived from them at creation / first use (the folded
        LayerNorm weights, the Gram tables) -- ``hive_dpt_weights_modified``.  (``DPTDepthModel.native()`` does not need it: it rebuilds the object when a
        parameter's version or address changed.)"""
        self.ctx.follow_torch_stream()
        self.ctx.check(self.ctx.lib.hive_dpt_weights_modified(self.handle))

    def arena_bytes(self):
        """Bytes of the activation arena (high-water mark of the largest forward so far)."""
        n = ctypes.c_int64(0)
        self.ctx.check(self.ctx.lib.hive_dpt_arena_bytes(self.handle, ctypes.byref(n)))
        return int(n.value)

    def close(self):
        if getattr(self, "handle", None) and _lib.alive():
            self.ctx.lib.hive_dpt_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
