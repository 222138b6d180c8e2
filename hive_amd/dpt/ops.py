"""The layers of DPT's convolutional parts on the hand-written kernels (``csrc/conv.hip``, ``stem.hip``, ``dpt_ops.hip``,
``vit.hip``): implicit-GEMM convolutions with fused epilogues, GroupNorm [+ residual] [+ ReLU], bilinear x2 upsampling.

``engine="hip"`` means EVERY layer runs a HIP kernel, on 16-bit (bfloat16, or float16 as the reference's ``model.half()``,
/root/reference/hive/dataset_adaptors.py:1394-1401) channels-last CUDA tensors; a layer the kernels do not cover raises
``HiveError`` -- there is no per-layer drop to a PyTorch operator.  ``engine="torch"`` is the plain PyTorch formulation: the
float32 / CPU reference the numerics tests compare against (and ``estimate_depth_dpt(optimize=False)``'s float32 network)."""
import os

import torch
import torch.nn.functional as F

from hive_amd import _lib

HALF_TYPES = (torch.float16, torch.bfloat16)


def not_covered(what, why):
    """engine="hip" and a layer outside what the kernels cover: fail loudly."""
    raise _lib.HiveError(_lib.ERR_INVALID, f"engine='hip': {what} is not covered by the HIP kernels ({why}); the HIP engine never drops to a "
                                           f"PyTorch operator -- build the model with engine='torch' for the PyTorch formulation")


def _why_not_map(x, channels_multiple=8):
    """None if ``x`` is a 16-bit channels-last CUDA activation the kernels take, else the reason."""
    if not x.is_cuda:
        return "the tensor is not on the GPU"
    if x.dtype not in HALF_TYPES:
        return f"dtype {x.dtype}: the kernels compute in float16 / bfloat16"
    if x.dim() != 4 or not x.is_contiguous(memory_format=torch.channels_last):
        return "the activation must be a 4-d tensor in channels_last memory format"
    if x.shape[1] % channels_multiple:
        return f"{x.shape[1]} channels: a multiple of {channels_multiple} is needed"
    return None


def _hip_eligible(x):
    return _why_not_map(x) is None


def _code(dtype):
    return _lib.BF16 if dtype == torch.bfloat16 else _lib.F16


def group_norm_act(x, num_groups, weight, bias, eps, relu=True, residual=None, engine="torch", stats=None):
    """relu?(group_norm(x) (+ residual)).  x: [N, C, H, W].  ``stats``: the (partial sums, tile rows) the convolution that wrote
    ``x`` left (``conv2d(..., gn_stats=True)``): the statistics pass over ``x`` is skipped."""
    c = x.shape[1]
    if engine == "hip":
        why = _why_not_map(x) or (None if (c & (c - 1)) == 0 and c <= 2048 else f"{c} channels: a power of two <= 2048 is needed") or \
            (None if weight.dtype == x.dtype else f"affine parameters are {weight.dtype}, the tensor {x.dtype}")
        if why:
            not_covered("group_norm", why)
        if residual is not None:
            assert residual.shape == x.shape and residual.dtype == x.dtype
            residual = residual.contiguous(memory_format=torch.channels_last)
        n, _, h, w = x.shape
        out = torch.empty_like(x)  # preserves channels_last
        ctx = _lib.default_context(x.device.index or 0)
        partial, tile_rows = stats if stats is not None else (None, 0)
        ctx.check(ctx.lib.hive_nhwc_group_norm_stats(ctx.handle, x.data_ptr(), _code(x.dtype), n, h * w, c, num_groups, weight.data_ptr(),
                                                     bias.data_ptr(), float(eps), _lib.ptr(residual), int(bool(relu)), out.data_ptr(),
                                                     _lib.ptr(partial), int(tile_rows)))
        return out
    y = F.group_norm(x, num_groups, weight, bias, eps)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


def _module_cache(module):
    """Derived (re-laid-out) weights live ON the module that owns the parameters: the cache dies with its owner and cannot be
    hit by another object (a process-global dict keyed by id(module) can, once the id is recycled)."""
    cache = module.__dict__.get("_hive_cache")
    if cache is None:
        cache = module.__dict__["_hive_cache"] = {}
    return cache


def _conv3x3_weight(conv):
    """[C_out][3][3][C_in] view of the weights = the weight tensor in channels-last memory format (what
    ``model.to(memory_format=torch.channels_last)`` already stores)."""
    w = conv.weight
    if w.is_contiguous(memory_format=torch.channels_last):
        return w
    stamp = (w.data_ptr(), w._version)
    cache = _module_cache(conv)
    hit = cache.get("channels_last")
    if hit is None or hit[0] != stamp:
        hit = cache["channels_last"] = (stamp, w.detach().contiguous(memory_format=torch.channels_last))
    return hit[1]


def _why_not_conv(x, conv, kernels=(1, 3), cout_multiple=64):
    why = _why_not_map(x, 64)
    if why:
        return why
    k = conv.kernel_size
    if not (k[0] == k[1] and k[0] in kernels):
        return f"kernel {tuple(k)}"
    if not (conv.stride[0] == conv.stride[1] and conv.stride[0] in (1, 2)):
        return f"stride {tuple(conv.stride)}"
    if tuple(conv.dilation) != (1, 1) or conv.groups != 1:
        return "dilation / groups"
    if conv.in_channels % 64 or conv.out_channels % cout_multiple or x.shape[1] != conv.in_channels:
        return f"{conv.in_channels} -> {conv.out_channels} channels: multiples of 64 -> {cout_multiple} are needed"
    if conv.weight.dtype != x.dtype:
        return f"weights are {conv.weight.dtype}, the tensor {x.dtype}"
    if not (conv.padding[0] == conv.padding[1] and conv.padding[0] < k[0]):
        return f"padding {tuple(conv.padding)}"
    return None


def conv3x3_eligible(x, conv):
    """The implicit-GEMM kernel (csrc/conv.hip) with the decoder's fused tail covers 3 x 3, stride 1, padding 1 on 16-bit channels-last with
    C_in % 64 == 0 and C_out % 128 == 0 -- every 3 x 3 convolution of DPT's decoder."""
    return (_why_not_conv(x, conv, kernels=(3,), cout_multiple=128) is None and tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (1, 1))


def conv3x3(x, conv, relu=False, residual=None, residual2=None, also_relu=False, with_bias=True):
    """relu?(conv(x) (+ bias) (+ residual) (+ residual2)) [and relu of it] in ONE HIP kernel; see ``conv3x3_eligible``."""
    n, _, h, w = x.shape
    out = torch.empty((n, conv.out_channels, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    out_relu = torch.empty_like(out) if also_relu else None
    for r in (residual, residual2):
        assert r is None or (r.shape == out.shape and r.dtype == out.dtype and r.is_contiguous(memory_format=torch.channels_last))
    bias = conv.bias if (with_bias and conv.bias is not None) else None
    ctx = _lib.default_context(x.device.index or 0)
    ctx.check(ctx.lib.hive_nhwc_conv3x3(ctx.handle, x.data_ptr(), _code(x.dtype), n, h, w, conv.in_channels, conv.out_channels,
                                        _conv3x3_weight(conv).data_ptr(), _lib.ptr(bias), int(bool(relu)), _lib.ptr(residual),
                                        _lib.ptr(residual2), out.data_ptr(), _lib.ptr(out_relu)))
    return (out, out_relu) if also_relu else out


def conv_geometry(conv, ih, iw, same_pad=False):
    """(kernel, stride, pad_top, pad_left, out_h, out_w) of a square-kernel convolution on an ih x iw input.  ``same_pad``:
    timm's StdConv2dSame (TensorFlow "SAME": total padding max((ceil(i / s) - 1) s + k - i, 0), the odd pixel at the bottom /
    right); otherwise nn.Conv2d's symmetric ``padding``."""
    k, st = conv.kernel_size[0], conv.stride[0]
    if same_pad:
        oh, ow = -(-ih // st), -(-iw // st)
        ph, pw = max((oh - 1) * st + k - ih, 0), max((ow - 1) * st + k - iw, 0)
        return k, st, ph // 2, pw // 2, oh, ow
    p = conv.padding[0]
    return k, st, p, p, (ih + 2 * p - k) // st + 1, (iw + 2 * p - k) // st + 1


def conv_eligible(x, conv):
    """csrc/conv.hip covers square kernels 1 / 3, stride 1 / 2, 16-bit channels-last, C_in % 64 == 0, C_out % 64 == 0, no groups /
    dilation -- every convolution of DPT-Hybrid except the 7 x 7 stem (its own kernel)."""
    return _why_not_conv(x, conv) is None


def require_conv(x, conv, what="convolution"):
    why = _why_not_conv(x, conv)
    if why:
        not_covered(what, why)


def conv2d(x, conv, weight=None, same_pad=False, relu=False, residual=None, residual2=None, also_relu=False, with_bias=True, gn_stats=False):
    """relu?(conv(x) (+ bias) (+ residuals)) through hive_nhwc_conv (see ``conv_eligible``).  ``weight``: use this tensor
    instead of ``conv.weight`` (the standardised weight of a StdConv2dSame), [C_out, C_in, k, k] in channels-last memory format.
    ``gn_stats``: also leave the per-tile channel sums of the output for the GroupNorm that follows (hive_nhwc_conv_gn); they ride
    on the returned tensor as ``out.hive_gn_stats = (partial, tile_rows)`` for ``group_norm_act(..., stats=)``."""
    n, _, ih, iw = x.shape
    k, st, pt, pl, oh, ow = conv_geometry(conv, ih, iw, same_pad)
    w = weight if weight is not None else _conv3x3_weight(conv)
    if not w.is_contiguous(memory_format=torch.channels_last):
        w = w.contiguous(memory_format=torch.channels_last)
    out = torch.empty((n, conv.out_channels, oh, ow), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    out_relu = torch.empty_like(out) if also_relu else None
    for r in (residual, residual2):
        assert r is None or (r.shape == out.shape and r.dtype == out.dtype and r.is_contiguous(memory_format=torch.channels_last))
    bias = conv.bias if (with_bias and conv.bias is not None) else None
    ctx = _lib.default_context(x.device.index or 0)
    if gn_stats:
        import ctypes
        partial = torch.empty(int(ctx.lib.hive_nhwc_conv_gn_partial_floats(n * oh * ow, conv.out_channels)), dtype=torch.float32, device=x.device)
        tile_rows = ctypes.c_int(0)
        ctx.check(ctx.lib.hive_nhwc_conv_gn(ctx.handle, x.data_ptr(), _code(x.dtype), n, ih, iw, conv.in_channels, conv.out_channels, k, st, pt, pl, oh, ow,
                                            w.data_ptr(), _lib.ptr(bias), int(bool(relu)), _lib.ptr(residual), _lib.ptr(residual2), out.data_ptr(),
                                            _lib.ptr(out_relu), partial.data_ptr(), partial.numel(), ctypes.byref(tile_rows)))
        out.hive_gn_stats = (partial, tile_rows.value)
    else:
        ctx.check(ctx.lib.hive_nhwc_conv(ctx.handle, x.data_ptr(), _code(x.dtype), n, ih, iw, conv.in_channels, conv.out_channels, k, st, pt, pl, oh, ow,
                                         w.data_ptr(), _lib.ptr(bias), int(bool(relu)), _lib.ptr(residual), _lib.ptr(residual2), out.data_ptr(),
                                         _lib.ptr(out_relu)))
    return (out, out_relu) if also_relu else out


def conv_gn_act(x, conv, norm, weight=None, same_pad=False, relu=True, residual=None):
    """relu?(group_norm(conv(x)) (+ residual)) through hive_nhwc_conv_gn_apply (the convolution's output never reaches memory), or
    None where that is not eligible (the caller then runs ``conv2d`` + ``group_norm_act``).  ``norm``: the nn.GroupNorm."""
    import ctypes
    if not (conv_eligible(x, conv) and conv.bias is None and norm.weight.dtype == x.dtype):
        return None
    n, _, ih, iw = x.shape
    k, st, pt, pl, oh, ow = conv_geometry(conv, ih, iw, same_pad)
    cout, g = conv.out_channels, norm.num_groups
    if cout % 256 or cout % g or (cout // g) % 8 or oh * ow < 256:
        return None
    w = weight if weight is not None else _conv3x3_weight(conv)
    if not w.is_contiguous(memory_format=torch.channels_last):
        w = w.contiguous(memory_format=torch.channels_last)
    out = torch.empty((n, cout, oh, ow), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    assert residual is None or (residual.shape == out.shape and residual.dtype == out.dtype and residual.is_contiguous(memory_format=torch.channels_last))
    ctx = _lib.default_context(x.device.index or 0)
    scratch = torch.empty(int(ctx.lib.hive_nhwc_conv_gn_partial_floats(n * oh * ow, cout)) + 2 * n * g, dtype=torch.float32, device=x.device)
    fused = ctypes.c_int(0)
    # 1 x 1 convolutions whose input is narrow enough: GroupNorm statistics from the input's Gram matrices (csrc/gram.hip) -- the rule of the network object
    # (csrc/dpt_net.hip conv_norm), so that both orchestrations stay bit-identical; HIVE_GN_GRAM=0 switches it off
    cin = conv.in_channels
    out_elems = n * oh * ow * cout
    if k == 1 and os.environ.get("HIVE_GN_GRAM", "1") != "0" and not getattr(ctx, "deterministic", False) and g == 32 and ((cin in (64, 128) and out_elems >= 150_000_000) or
                                                                               (cin == 256 and st == 2 and out_elems >= 250_000_000)):
        def tables():
            t = torch.empty(int(ctx.lib.hive_gn_gram_table_floats(cin, g)), dtype=torch.float32, device=x.device)
            ctx.check(ctx.lib.hive_gn_gram_prepare(ctx.handle, w.data_ptr(), _code(x.dtype), cin, cout, g, t.data_ptr()))
            return t
        tab = _cached(conv, ("gram", x.dtype, weight is not None), [conv.weight], tables)  # (the standardised weight is a function of conv.weight)
        ctx.check(ctx.lib.hive_nhwc_conv_gn_apply_gram(ctx.handle, x.data_ptr(), _code(x.dtype), n, ih, iw, cin, cout, st, oh, ow, w.data_ptr(), tab.data_ptr(), g,
                                                       norm.weight.data_ptr(), norm.bias.data_ptr(), float(norm.eps), _lib.ptr(residual), int(bool(relu)),
                                                       out.data_ptr(), scratch.data_ptr(), scratch.numel(), ctypes.byref(fused)))
        if fused.value:
            return out
    ctx.check(ctx.lib.hive_nhwc_conv_gn_apply(ctx.handle, x.data_ptr(), _code(x.dtype), n, ih, iw, conv.in_channels, cout, k, st, pt, pl, oh, ow, w.data_ptr(), g,
                                              norm.weight.data_ptr(), norm.bias.data_ptr(), float(norm.eps), _lib.ptr(residual), int(bool(relu)),
                                              out.data_ptr(), scratch.data_ptr(), scratch.numel(), ctypes.byref(fused)))
    return out if fused.value else None


def _cached(layer, what, params, build):
    """Weights re-laid-out for the kernels, kept on ``layer`` and rebuilt when a parameter changes."""
    stamp = tuple((p.data_ptr(), p._version, p.dtype, p.device) for p in params)
    cache = _module_cache(layer)
    hit = cache.get(what)
    if hit is None or hit[0] != stamp:
        with torch.no_grad():
            hit = cache[what] = (stamp, build())
    return hit[1]


def patch_embed_eligible(x, conv):
    k = conv.kernel_size
    return (x.is_cuda and x.dtype in HALF_TYPES and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and type(conv) is torch.nn.Conv2d
            and k[0] == k[1] and tuple(conv.stride) == tuple(k) and tuple(conv.padding) == (0, 0) and conv.groups == 1 and conv.weight.dtype == x.dtype
            and x.shape[2] % k[0] == 0 and x.shape[3] % k[0] == 0 and (k[0] * k[0] * conv.in_channels) % 64 == 0 and conv.out_channels % 128 == 0)


def patch_embed(x, conv):
    """timm ``PatchEmbed.proj`` (Conv2d(C, D, P, P)) -> tokens [N, (H/P)(W/P), D]: hive_patch_rows + the hand-written GEMM."""
    n, c, h, w = x.shape
    p, d = conv.kernel_size[0], conv.out_channels
    wmat, bias = _cached(conv, "patch", (conv.weight, conv.bias) if conv.bias is not None else (conv.weight,), lambda: (
        conv.weight.permute(0, 2, 3, 1).reshape(d, p * p * c).contiguous(),
        (conv.bias.float() if conv.bias is not None else torch.zeros(d, device=x.device)).contiguous()))
    m = n * (h // p) * (w // p)
    cols = torch.empty((m, p * p * c), dtype=x.dtype, device=x.device)
    out = torch.empty((n, (h // p) * (w // p), d), dtype=x.dtype, device=x.device)
    ctx = _lib.default_context(x.device.index or 0)
    ctx.check(ctx.lib.hive_patch_rows(ctx.handle, x.data_ptr(), _code(x.dtype), n, h, w, c, p, cols.data_ptr()))
    ctx.check(ctx.lib.hive_vit_linear(ctx.handle, cols.data_ptr(), _code(x.dtype), wmat.data_ptr(), bias.data_ptr(), None, out.data_ptr(), m, d, p * p * c, 0))
    return out


def conv_transpose_eligible(x, layer):
    k = layer.kernel_size
    return (x.is_cuda and x.dtype in HALF_TYPES and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
            and isinstance(layer, torch.nn.ConvTranspose2d) and k[0] == k[1] and tuple(layer.stride) == tuple(k) and tuple(layer.padding) == (0, 0)
            and tuple(layer.output_padding) == (0, 0) and layer.groups == 1 and tuple(layer.dilation) == (1, 1) and layer.weight.dtype == x.dtype
            and layer.in_channels % 64 == 0 and layer.out_channels % 64 == 0 and k[0] <= 8)


def conv_transpose(x, layer):
    """ConvTranspose2d with kernel == stride (DPT-Large's reassemble stages): a 1 x 1 convolution to s s C_out channels in (dy, dx, co)
    order (hive_nhwc_conv) and the scatter + bias of hive_nhwc_pixel_shuffle_bias."""
    n, cin, h, w = x.shape
    s_, cout = layer.kernel_size[0], layer.out_channels
    wmat = _cached(layer, "convT", (layer.weight,), lambda: layer.weight.permute(2, 3, 1, 0).reshape(s_ * s_ * cout, cin).contiguous())  # [(dy, dx, co)][ci]
    tmp = torch.empty((n * h * w, s_ * s_ * cout), dtype=x.dtype, device=x.device)
    out = torch.empty((n, cout, h * s_, w * s_), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    ctx = _lib.default_context(x.device.index or 0)
    ctx.check(ctx.lib.hive_nhwc_conv(ctx.handle, x.data_ptr(), _code(x.dtype), n, h, w, cin, s_ * s_ * cout, 1, 1, 0, 0, h, w, wmat.data_ptr(), None, 0, None, None,
                                     tmp.data_ptr(), None))
    ctx.check(ctx.lib.hive_nhwc_pixel_shuffle_bias(ctx.handle, tmp.data_ptr(), _lib.ptr(layer.bias), _code(x.dtype), n, h, w, cout, s_, out.data_ptr()))
    return out


def stem_conv_eligible(x, conv):
    return (x.is_cuda and x.dtype in HALF_TYPES and x.dim() == 4 and x.shape[1] == 3 and x.is_contiguous(memory_format=torch.channels_last)
            and tuple(conv.kernel_size) == (7, 7) and tuple(conv.stride) == (2, 2) and conv.in_channels == 3 and conv.out_channels == 64
            and conv.bias is None and x.shape[2] >= 7 and x.shape[3] >= 7)


def stem_conv(x, conv, weight):
    """The 7 x 7 / 2 "SAME" stem convolution (csrc/stem.hip).  ``weight``: the standardised [64, 3, 7, 7] weights."""
    stamp = (weight.data_ptr(), weight._version, weight.dtype)
    cache = _module_cache(conv)
    hit = cache.get("stem")
    if hit is None or hit[0] != stamp:
        w = torch.zeros((64, 7, 32), dtype=x.dtype, device=weight.device)
        w[:, :, :21] = weight.detach().permute(0, 2, 3, 1).reshape(64, 7, 21)  # (ky, (kx, c)), a kernel row padded to 32
        hit = cache["stem"] = (stamp, w.contiguous())
    n, _, h, w_ = x.shape
    oh, ow = (h + 1) // 2, (w_ + 1) // 2
    out = torch.empty((n, 64, oh, ow), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    ctx = _lib.default_context(x.device.index or 0)
    # the statistics of the GroupNorm behind the convolution come out of its epilogue (hive_resnet_stem_conv_gn), as for every other
    # StdConv2dSame; they ride on the tensor as ``out.hive_gn_stats = (partial, tile_rows)`` (tile_rows 0: none were written)
    import ctypes
    partial = torch.empty(int(ctx.lib.hive_nhwc_conv_gn_partial_floats(n * oh * ow, 64)), dtype=torch.float32, device=x.device)
    tile_rows = ctypes.c_int(0)
    ctx.check(ctx.lib.hive_resnet_stem_conv_gn(ctx.handle, x.data_ptr(), _code(x.dtype), n, h, w_, hit[1].data_ptr(), out.data_ptr(), partial.data_ptr(),
                                               partial.numel(), ctypes.byref(tile_rows)))
    if tile_rows.value:
        out.hive_gn_stats = (partial, tile_rows.value)
    return out


def bneck_gn_conv3x3(t, norm, conv, weight):
    """conv(relu(norm(t))) for the 64-channel bottlenecks (hive_bneck_gn_conv3x3: the GroupNorm + ReLU applied while the 3 x 3 convolution
    stages its input; csrc/bneck.hip), or None where it does not apply (then: ``group_norm_act`` + ``conv2d``).  ``t`` is conv1's output
    with the sums its epilogue left (``t.hive_gn_stats``); ``weight``: conv's standardised [64, 64, 3, 3] weights.  The result carries
    ``hive_gn_stats`` for the GroupNorm behind it."""
    import ctypes
    stats = getattr(t, "hive_gn_stats", None)
    if (stats is None or _why_not_map(t) or t.shape[1] != 64 or conv.in_channels != 64 or conv.out_channels != 64 or tuple(conv.kernel_size) != (3, 3)
            or tuple(conv.stride) != (1, 1) or conv.bias is not None or not norm.apply_act or norm.num_groups != 32 or norm.weight.dtype != t.dtype):
        return None
    w = weight if weight.is_contiguous(memory_format=torch.channels_last) else weight.contiguous(memory_format=torch.channels_last)
    n, c, h, wd = t.shape
    out = torch.empty_like(t)  # preserves channels_last
    ctx = _lib.default_context(t.device.index or 0)
    partial = torch.empty(int(ctx.lib.hive_nhwc_conv_gn_partial_floats(n * h * wd, 64)), dtype=torch.float32, device=t.device)
    tile_rows, fused = ctypes.c_int(0), ctypes.c_int(0)
    ctx.check(ctx.lib.hive_bneck_gn_conv3x3(ctx.handle, t.data_ptr(), _code(t.dtype), n, h, wd, c, stats[0].data_ptr(), int(stats[1]), norm.weight.data_ptr(),
                                            norm.bias.data_ptr(), float(norm.eps), w.data_ptr(), out.data_ptr(), partial.data_ptr(), partial.numel(),
                                            ctypes.byref(tile_rows), ctypes.byref(fused)))
    if not fused.value:
        return None
    if tile_rows.value:
        out.hive_gn_stats = (partial, tile_rows.value)
    return out


def group_norm_relu_maxpool(x, norm, stats=None):
    """MaxPool2dSame(3, 2)(relu(norm(x))) in one pass (hive_nhwc_group_norm_relu_maxpool): the ResNetV2 stem behind its convolution.
    ``stats``: ``x.hive_gn_stats`` of the convolution that wrote ``x``, or None (own statistics pass)."""
    why = _why_not_map(x) or (None if norm.weight.dtype == x.dtype else f"affine parameters are {norm.weight.dtype}, the tensor {x.dtype}")
    n, c, h, w = x.shape
    if not why and ((c & (c - 1)) or c > 2048):
        why = f"{c} channels: a power of two <= 2048 is needed"
    if why:
        not_covered("group_norm + max pool", why)
    out = torch.empty((n, c, (h + 1) // 2, (w + 1) // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    ctx = _lib.default_context(x.device.index or 0)
    partial, tile_rows = stats if stats is not None else (None, 0)
    ctx.check(ctx.lib.hive_nhwc_group_norm_relu_maxpool(ctx.handle, x.data_ptr(), _code(x.dtype), n, h, w, c, norm.num_groups, norm.weight.data_ptr(),
                                                        norm.bias.data_ptr(), float(norm.eps), out.data_ptr(), _lib.ptr(partial), int(tile_rows)))
    return out


def maxpool3x3s2_same(x, engine="torch"):
    """MaxPool2dSame(3, 2) of the ResNetV2 stem (``engine="hip"``; raises where the kernel does not apply)."""
    if engine == "hip":
        why = _why_not_map(x)
        if why:
            not_covered("max_pool 3 x 3 / 2", why)
        n, c, h, w = x.shape
        out = torch.empty((n, c, (h + 1) // 2, (w + 1) // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        ctx = _lib.default_context(x.device.index or 0)
        ctx.check(ctx.lib.hive_nhwc_maxpool3x3s2(ctx.handle, x.data_ptr(), _code(x.dtype), n, h, w, c, out.data_ptr()))
        return out
    return None


def conv_bias_act(x, conv, relu=False, residual=None, residual2=None, engine="torch", also_relu=False):
    """relu?(conv(x) (+ residual) (+ residual2)) for an ``nn.Conv2d`` with bias.  HIP engine: the decoder's 3 x 3
    convolutions run in the hand-written implicit-GEMM kernel with bias, skip connections and ReLU in its epilogue.
    ``also_relu=True`` returns ``(y, relu(y))``: the second tensor is what the next residual unit feeds to its first convolution."""
    if engine == "hip":
        why = _why_not_conv(x, conv, kernels=(3,), cout_multiple=128) or (None if conv3x3_eligible(x, conv) else "3 x 3 / stride 1 / padding 1 only") or \
            (None if conv.bias is None or conv.bias.dtype == x.dtype else f"bias is {conv.bias.dtype}, the tensor {x.dtype}")
        if why:
            not_covered("convolution of a residual unit", why)
        if residual is not None:
            residual = residual.contiguous(memory_format=torch.channels_last)
        if residual2 is not None:
            residual2 = residual2.contiguous(memory_format=torch.channels_last)
        return conv3x3(x, conv, relu=relu, residual=residual, residual2=residual2, also_relu=also_relu)
    y = conv(x)
    if residual is not None:
        y = y + residual
    if residual2 is not None:
        y = y + residual2
    y = F.relu(y) if relu else y
    return (y, F.relu(y)) if also_relu else y


def upsample2x(x, engine="torch", bias=None):
    """interpolate(x (+ bias per channel), scale_factor=2, mode="bilinear", align_corners=True).  ``bias`` folds the bias
    pass of the convolution that produced ``x`` into the load (x + b is rounded to the tensor dtype first, as a separate
    add would round it)."""
    if engine == "hip":
        why = _why_not_map(x) or (None if bias is None or bias.dtype == x.dtype else f"bias is {bias.dtype}, the tensor {x.dtype}")
        if why:
            not_covered("bilinear x2 upsampling", why)
        n, c, h, w = x.shape
        out = torch.empty((n, c, 2 * h, 2 * w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        ctx = _lib.default_context(x.device.index or 0)
        ctx.check(ctx.lib.hive_nhwc_upsample2x(ctx.handle, x.data_ptr(), _lib.ptr(bias), _code(x.dtype), n, h, w, c, out.data_ptr()))
        return out
    if bias is not None:
        x = x + bias.view(1, -1, 1, 1)
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
