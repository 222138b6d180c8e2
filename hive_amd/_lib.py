"""ctypes binding of ``libhive_mi355x.so`` (the C ABI in ``include/hive_mi355x.h``).

There is no CPU fallback: if the shared library is missing, or no gfx950 device is visible when a
context is requested, this module raises -- it never routes to numpy or to the test oracle.
"""
import atexit
import ctypes
import os
import sys
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HIVE_AMD_LIB") or os.path.join(_HERE, "lib", "libhive_mi355x.so")  # override: kernel experiments

OK, ERR_INVALID, ERR_DEVICE, ERR_NOMEM, ERR_EMPTY, ERR_STATE = 0, -1, -2, -3, -4, -5
MEM_HOST, MEM_DEVICE = 0, 1
ROUND_HALF_EVEN, ROUND_HALF_AWAY = 0, 1
F32, F16, BF16 = 0, 1, 2
ABI_VERSION = 2  # include/hive_mi355x.h HIVE_ABI_VERSION: 2 = the network entry points take the 16-bit dtype (HIVE_F16 / HIVE_BF16)

c_void_p, c_int, c_int64, c_float, c_double = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double
P = ctypes.POINTER

# name -> (restype, argtypes); every symbol declared in include/hive_mi355x.h
SIGNATURES = {
    "hive_abi_version": (c_int, []),
    "hive_ctx_create": (c_int, [c_int, c_void_p, P(c_void_p)]),
    "hive_ctx_destroy": (c_int, [c_void_p]),
    "hive_ctx_get_stream": (c_int, [c_void_p, P(c_void_p)]),
    "hive_ctx_synchronize": (c_int, [c_void_p]),
    "hive_last_error": (ctypes.c_char_p, [c_void_p]),
    "hive_ctx_release_stream": (c_int, [c_void_p]),
    "hive_ctx_set_stream": (c_int, [c_void_p, c_void_p]),
    "hive_ctx_set_round_mode": (c_int, [c_void_p, c_int]),
    "hive_ctx_set_deterministic": (c_int, [c_void_p, c_int]),
    "hive_ctx_launch_stats": (c_int, [c_void_p, P(c_int64), P(c_int64), c_int]),
    "hive_ctx_set_timing": (c_int, [c_void_p, c_int]),
    "hive_ctx_last_kernel_ms": (c_int, [c_void_p, P(c_float)]),
    "hive_ctx_kernel_time_total": (c_int, [c_void_p, P(c_int), P(c_float)]),
    "hive_tsdf_dims": (c_int, [c_void_p, c_double, c_void_p]),
    "hive_tsdf_create": (c_int, [c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_void_p, P(c_void_p)]),
    "hive_tsdf_create_slab": (c_int, [c_void_p, c_void_p, c_double, c_int64, c_int64, c_void_p, c_void_p, c_void_p, P(c_void_p)]),
    "hive_tsdf_slab_info": (c_int, [c_void_p, P(c_int64), P(c_int64)]),
    "hive_tsdf_destroy": (c_int, [c_void_p]),
    "hive_tsdf_set_round_mode": (c_int, [c_void_p, c_int]),
    "hive_tsdf_reset": (c_int, [c_void_p]),
    "hive_tsdf_planes_modified": (c_int, [c_void_p]),
    "hive_tsdf_info": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, P(c_float), P(c_float)]),
    "hive_tsdf_device_ptrs": (c_int, [c_void_p, P(c_void_p), P(c_void_p), P(c_void_p)]),
    "hive_tsdf_integrate": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_float, c_int,
                                    P(ctypes.c_uint64)]),
    "hive_tsdf_integrate_batch": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_float,
                                          c_int]),
    "hive_tsdf_last_batch_groups": (c_int, [c_void_p, c_void_p, c_int, P(c_int)]),
    "hive_tsdf_last_sweep_items": (c_int, [c_void_p, P(ctypes.c_uint64), P(c_int)]),
    "hive_tsdf_stats": (c_int, [c_void_p, P(c_int64), P(c_int64)]),
    "hive_tsdf_get_volume": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "hive_tsdf_set_volume": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "hive_tsdf_extract_mesh": (c_int, [c_void_p, P(c_int64), P(c_int64)]),
    "hive_tsdf_copy_mesh": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "hive_tsdf_copy_mesh_voxel_coords": (c_int, [c_void_p, c_void_p]),
    "hive_tsdf_accum_reset": (c_int, [c_void_p, c_void_p]),
    "hive_tsdf_accum_integrate": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_float,
                                          c_int]),
    "hive_tsdf_accum_finalize": (c_int, [c_void_p, c_void_p]),
    "hive_tsdf_accum_finalize_to": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "hive_tsdf_accum_from_volume": (c_int, [c_void_p, c_void_p]),
    "hive_tsdf_accum_from_volume_sharded": (c_int, [c_void_p, c_void_p, c_int, c_int64]),
    "hive_tsdf_set_volume_range": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "hive_view_frustum": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "hive_view_frustum_batch": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "hive_depth_apply_mask": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "hive_depth_mm_to_m": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_float, c_void_p]),
    "hive_unproject": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                               c_void_p, c_void_p, c_int64, P(c_int64)]),
    "hive_image2world": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_double, c_int, c_void_p]),
    "hive_project": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_double, c_int, c_void_p, c_void_p,
                             c_void_p]),
    "hive_project_bbox": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "hive_grid_mesh": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_double, c_double, c_int, c_void_p, c_int64, P(c_int64), P(c_int64)]),
    "hive_fg_frame_mesh": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, c_void_p, c_int64, c_void_p, c_int64,
                                   c_void_p, P(c_int64), P(c_int64), c_void_p]),
    "hive_filter_faces": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_double, c_double, c_int, c_void_p, P(c_int64)]),
    "hive_texture_window": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_double, c_int, c_void_p, c_void_p]),
    "hive_dilate_mask": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "hive_dilate_mask_se": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "hive_depth_apply_mask_se": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "hive_vit_create": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, P(c_void_p)]),
    "hive_vit_weights_modified": (c_int, [c_void_p]),
    "hive_vit_destroy": (c_int, [c_void_p]),
    "hive_vit_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "hive_vit_layernorm": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float]),
    "hive_vit_linear": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int]),
    "hive_vit_qkv": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int]),
    "hive_vit_attention": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int]),
    "hive_dpt_preprocess": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_float, c_int, c_void_p]),
    "hive_dpt_head_tail": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_int, c_void_p, c_float, c_int, c_int, c_float,
                                   c_float, c_void_p, c_float, c_float, c_void_p, c_void_p]),
    "hive_dpt_head_fused": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float,
                                    c_int, c_int, c_float, c_float, c_void_p, c_float, c_float, c_void_p, c_void_p]),
    "hive_nhwc_group_norm": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_int,
                                     c_void_p]),
    "hive_nhwc_bias_act": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "hive_nhwc_upsample2x": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "hive_nhwc_conv3x3": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                  c_void_p, c_void_p]),
    "hive_nhwc_conv": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                               c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "hive_nhwc_conv_gn_partial_floats": (c_int64, [c_int64, c_int]),
    "hive_nhwc_conv_gn": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                  c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, P(c_int)]),
    "hive_nhwc_conv_gn_apply": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                        c_void_p, c_void_p, c_float, c_void_p, c_int, c_void_p, c_void_p, c_int64, P(c_int)]),
    "hive_nhwc_group_norm_stats": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_int,
                                           c_void_p, c_void_p, c_int]),
    "hive_gn_gram_table_floats": (c_int64, [c_int, c_int]),
    "hive_gn_gram_prepare": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "hive_gn_gram_parts": (c_int, [c_void_p, c_int, c_int, c_int, c_int]),
    "hive_gn_gram_stats": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_float, c_void_p, c_void_p,
                                   c_void_p]),
    "hive_nhwc_conv_gn_apply_gram": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                             c_void_p, c_float, c_void_p, c_int, c_void_p, c_void_p, c_int64, P(c_int)]),
    "hive_patch_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "hive_nhwc_pixel_shuffle_bias": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "hive_resnet_stem_conv": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "hive_nhwc_maxpool3x3s2": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "hive_bneck_gn_conv3x3": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                                      c_int64, P(c_int), P(c_int)]),
    "hive_resnet_stem_conv_gn": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int64, P(c_int)]),
    "hive_nhwc_group_norm_relu_maxpool": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                                  c_int]),
    "hive_dpt_create": (c_int, [c_void_p, c_void_p, c_void_p, c_int, P(c_void_p)]),
    "hive_dpt_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    "hive_dpt_forward_frames": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    "hive_dpt_resize_preprocess": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_int, c_void_p]),
    "hive_depth_resize_nearest": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "hive_dpt_arena_bytes": (c_int, [c_void_p, P(c_int64)]),
    "hive_dpt_weights_modified": (c_int, [c_void_p]),
    "hive_dpt_destroy": (c_int, [c_void_p]),
    "hive_depth_quantize": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p]),
}

_lib = None
_lock = threading.Lock()
_shutdown = False


@atexit.register
def _mark_shutdown():
    # Registered after torch's own atexit hooks, so it runs before them: from here on native handles are
    # left to process teardown instead of calling into a HIP runtime that is being torn down.
    global _shutdown
    _shutdown = True


def alive():
    """False once the interpreter is exiting: destructors must not call the library any more."""
    return not _shutdown and not sys.is_finalizing()


class HiveError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libhive_mi355x error {code}: {message}")
        self.code = code


def load():
    """Load the shared library (after torch, so that both share one HIP runtime: torch's bundled
    libamdhip64.so.7 has the same soname as ROCm's)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
                f"`make -C hive_amd/csrc`. hive_amd has no CPU fallback.")
        try:
            import torch  # noqa: F401  (loads torch's HIP runtime first)
        except ImportError:
            pass
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        if lib.hive_abi_version() != ABI_VERSION:
            raise ImportError(f"{LIB_PATH}: ABI version {lib.hive_abi_version()} != {ABI_VERSION} (rebuild: make -C hive_amd/csrc)")
        _lib = lib
        return _lib


def dtype_code(torch_dtype):
    """hive_dtype of a torch 16-bit type; raises for anything the network kernels do not compute in."""
    name = str(torch_dtype)
    if name == "torch.bfloat16":
        return BF16
    if name == "torch.float16":
        return F16
    raise HiveError(ERR_INVALID, f"{name}: the network kernels compute in float16 or bfloat16 (no silent down-cast)")


def ptr(a):
    """Raw address of a numpy array / torch tensor / int / None."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return a.data_ptr()


def check(rc, ctx_handle=None):
    if rc != OK:
        msg = load().hive_last_error(ctx_handle)
        raise HiveError(rc, msg.decode() if msg else "")


class Context:
    """One ``hive_ctx`` = one (GPU, stream).  By default it rides on torch's current stream of the
    device so that hive kernels and torch kernels are ordered with each other (for torch's default
    stream that is HIP's null stream, handle 0).  ``stream="own"`` creates a private non-blocking
    stream; an int is taken as a ``hipStream_t``."""

    def __init__(self, device=0, stream="torch"):
        lib = load()
        self.device = int(device)
        handle = c_void_p()
        self._follow_torch = stream == "torch"
        self._stream = None
        self.deterministic = False
        self._side_stream = None  # the torch.cuda.Stream this context was made for (DepthFusionStream.side_stream_context)
        if stream == "torch":
            import torch
            if not torch.cuda.is_available():
                raise HiveError(ERR_DEVICE, "no HIP device visible to torch; hive_amd needs an MI355X (no CPU fallback)")
            self._stream = int(torch.cuda.current_stream(self.device).cuda_stream)
            stream_ptr = c_void_p(self._stream)
        elif stream == "own":
            stream_ptr = c_void_p(-1)  # HIVE_STREAM_OWN
        elif stream == "own_low":
            stream_ptr = c_void_p(-2)  # HIVE_STREAM_OWN_LOW: a private stream of the lowest dispatch priority
        else:
            stream_ptr = c_void_p(int(stream or 0))
        check(lib.hive_ctx_create(self.device, stream_ptr, ctypes.byref(handle)))
        self.handle = handle
        self.lib = lib

    def check(self, rc):
        check(rc, self.handle)

    def follow_torch_stream(self):
        """Re-bind to torch's *current* stream of the device if it changed since the last call (``with
        torch.cuda.stream(s)``): hive kernels must queue on the stream the surrounding torch ops use, or the
        two are unordered.  A no-op for contexts created with an explicit or private stream."""
        if self._follow_torch:
            import torch
            cur = int(torch.cuda.current_stream(self.device).cuda_stream)
            if cur != self._stream:
                self.check(self.lib.hive_ctx_set_stream(self.handle, c_void_p(cur)))
                self._stream = cur
        return self

    def stream_handle(self):
        """The hipStream_t (int) this context issues on."""
        h = c_void_p()
        self.check(self.lib.hive_ctx_get_stream(self.handle, ctypes.byref(h)))
        return int(h.value or 0)

    def torch_stream(self):
        """The torch stream object of a context that does NOT follow torch's current stream (a private or explicit hipStream_t):
        whoever issues torch ops or collectives that must be ordered with this context's kernels runs them inside
        ``with torch.cuda.stream(ctx.torch_stream())``.  None for a context that follows torch."""
        if self._follow_torch:
            return None
        if self._side_stream is None:
            import torch
            self._side_stream = torch.cuda.ExternalStream(self.stream_handle(), device=self.device)
            # torch now refers to the stream beyond this context's life (the caching allocator records an event on it whenever a tensor used there is
            # freed; process groups cache the streams they synchronised with): the library must not destroy it with the context
            self.check(self.lib.hive_ctx_release_stream(self.handle))
        return self._side_stream

    def synchronize(self):
        self.check(self.lib.hive_ctx_synchronize(self.handle))

    def set_round_mode(self, mode):
        self.check(self.lib.hive_ctx_set_round_mode(self.handle, int(mode)))

    def set_deterministic(self, enabled):
        """No split-K, no key split in attention, no Gram-matrix GroupNorm statistics: see ``hive_ctx_set_deterministic`` in include/hive_mi355x.h."""
        self.check(self.lib.hive_ctx_set_deterministic(self.handle, int(bool(enabled))))
        self.deterministic = bool(enabled)

    def launch_stats(self, reset=False):
        """(split-K launches, four-stage-ring launches) since creation / the last reset."""
        a, b = c_int64(0), c_int64(0)
        self.check(self.lib.hive_ctx_launch_stats(self.handle, ctypes.byref(a), ctypes.byref(b), int(bool(reset))))
        return int(a.value), int(b.value)

    def set_timing(self, enabled):
        self.check(self.lib.hive_ctx_set_timing(self.handle, int(bool(enabled))))

    def kernel_time_total(self):
        n, ms = c_int(0), c_float(0)
        self.check(self.lib.hive_ctx_kernel_time_total(self.handle, ctypes.byref(n), ctypes.byref(ms)))
        return n.value, ms.value

    def close(self):
        if getattr(self, "handle", None) and alive():
            self.lib.hive_ctx_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}
_tls = threading.local()


def default_context(device=None):
    """Per-thread, per-device default context (the reference calls the geometric functions from a
    ThreadPool, hive/pipeline.py:491; a hive_ctx is not re-entrant)."""
    if device is None:
        import torch
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    cache = getattr(_tls, "ctx", None)
    if cache is None:
        cache = _tls.ctx = {}
    if device not in cache:
        cache[device] = Context(device)
    return cache[device].follow_torch_stream()
