"""Depth estimation driver: ``estimate_depth_dpt`` of /root/reference/hive/dataset_adaptors.py:1346-1435,
and the on-device depth+fusion stream that keeps DPT -> TSDF inside HBM.
"""
import os

import numpy as np
import torch

from hive_amd import _lib
from hive_amd.dpt import transforms as dpt_transforms
from hive_amd.dpt.models import DPTDepthModel

NET_W, NET_H = 640, 480  # hard-coded in the reference (dataset_adaptors.py:1363-1364)
DPT_SCALE, DPT_SHIFT = 0.000305, 0.1378  # NYU fine-tuned DPT-Hybrid (dataset_adaptors.py:1368-1369)


def build_model(weights_path=None, device="cuda", dtype=torch.bfloat16, engine="hip", init_seed=None):
    """DPT-Hybrid-NYU as the reference constructs it (dataset_adaptors.py:1366-1374, 1394-1401), in
    channels-last 16-bit on the GPU.  ``weights_path=None`` gives a randomly initialised network: PyTorch's default
    initialisation (a constant ~7.25 m depth map), or, with ``init_seed``, the seeded non-degenerate weights of
    ``hive_amd.dpt.init`` (depth maps spanning ~1-7 m -- what ``bench.py`` and the numerics tests use)."""
    model = DPTDepthModel(path=weights_path, scale=DPT_SCALE, shift=DPT_SHIFT, invert=True, backbone="vitb_rn50_384",
                          non_negative=True, enable_attention_hooks=False, engine=engine)
    if weights_path is None and init_seed is not None:
        from hive_amd.dpt.init import seeded_init
        seeded_init(model, seed=init_seed)
    model.eval()
    model = model.to(memory_format=torch.channels_last)
    if dtype is not None:
        model = model.to(dtype)
    return model.to(device)


def make_transform():
    """Resize(640, 480, keep aspect, multiple of 32, "minimal", cubic) -> Normalize(.5, .5) -> PrepareForNet
    (dataset_adaptors.py:1376-1392)."""
    return dpt_transforms.Compose([
        dpt_transforms.Resize(NET_W, NET_H, resize_target=None, keep_aspect_ratio=True, ensure_multiple_of=32,
                              resize_method="minimal", image_interpolation_method=dpt_transforms.INTER_CUBIC),
        dpt_transforms.NormalizeImage(mean=[0.5, 0.5, 0.5], std=[0.5, 0.5, 0.5]),
        dpt_transforms.PrepareForNet(),
    ])


def network_size(frame_h, frame_w):
    """(net_h, net_w) the reference runs the network at for frames of this size: its ``Resize(640, 480, keep_aspect_ratio=True,
    ensure_multiple_of=32, resize_method="minimal")`` (dataset_adaptors.py:1363-1387) -- 480 x 640 frames stay as they are, 1080 x 1920 become 480 x 864."""
    net_w, net_h = dpt_transforms.Resize(NET_W, NET_H, resize_target=None, keep_aspect_ratio=True, ensure_multiple_of=32,
                                         resize_method="minimal").get_size(int(frame_w), int(frame_h))
    return int(net_h), int(net_w)


def resize_preprocess_on_device(frames_u8, net_size, dtype=torch.bfloat16, ctx=None):
    """uint8 [B, H, W, 3] on the GPU -> the network input [B, 3, net_h, net_w] (channels-last) through the reference's cv2.INTER_CUBIC resize +
    normalisation (dataset_adaptors.py:1376-1392, 1407-1417), one HIP kernel (``hive_dpt_resize_preprocess``).  ``dtype=torch.float32`` for checks."""
    assert frames_u8.dtype == torch.uint8 and frames_u8.is_cuda and frames_u8.dim() == 4 and frames_u8.shape[-1] == 3
    frames_u8 = frames_u8.contiguous()
    b, h, w, _ = frames_u8.shape
    net_h, net_w = int(net_size[0]), int(net_size[1])
    ctx = ctx or _lib.default_context(frames_u8.device.index or 0)
    code = _lib.F32 if dtype == torch.float32 else _lib.dtype_code(dtype)
    out = torch.empty((b, 3, net_h, net_w), dtype=dtype, device=frames_u8.device, memory_format=torch.channels_last)
    ctx.check(ctx.lib.hive_dpt_resize_preprocess(ctx.handle, frames_u8.data_ptr(), b, h, w, net_h, net_w, 0.5, 0.5, code, out.data_ptr()))
    return out


def resize_depth_nearest(depth, frame_size, max_depth=None, ctx=None):
    """depth f32 [B, h, w] on the GPU -> (depth [B, H, W], depth_mm, depth_m): torch's ``interpolate(mode="nearest")`` back to the frame size
    (dataset_adaptors.py:1421-1426) with the uint16-mm hand-off (``hive_depth_resize_nearest``); mm / m are None without ``max_depth``."""
    assert depth.dtype == torch.float32 and depth.is_cuda and depth.dim() == 3
    depth = depth.contiguous()
    b, h, w = depth.shape
    H, W = int(frame_size[0]), int(frame_size[1])
    ctx = ctx or _lib.default_context(depth.device.index or 0)
    out = torch.empty((b, H, W), dtype=torch.float32, device=depth.device)
    mm = torch.empty((b, H, W), dtype=torch.int16, device=depth.device) if max_depth is not None else None
    m = torch.empty((b, H, W), dtype=torch.float32, device=depth.device) if max_depth is not None else None
    ctx.check(ctx.lib.hive_depth_resize_nearest(ctx.handle, depth.data_ptr(), b, h, w, H, W, 1.0 / 1000.0, float(max_depth or 0.0), out.data_ptr(), _lib.ptr(mm),
                                                _lib.ptr(m)))
    return out, mm, m


def preprocess_on_device(frames_u8, dtype=torch.bfloat16, ctx=None):
    """uint8 [B, H, W, 3] on the GPU -> normalised channels-last network input [B, 3, H, W]:
    ``((x / 255) - 0.5) / 0.5`` (dataset_adaptors.py:1407 + NormalizeImage), one fused HIP kernel."""
    assert frames_u8.dtype == torch.uint8 and frames_u8.is_cuda and frames_u8.dim() == 4 and frames_u8.shape[-1] == 3
    frames_u8 = frames_u8.contiguous()
    b, h, w, _ = frames_u8.shape
    ctx = ctx or _lib.default_context(frames_u8.device.index or 0)
    out = torch.empty((b, 3, h, w), dtype=dtype, device=frames_u8.device, memory_format=torch.channels_last)
    ctx.check(ctx.lib.hive_dpt_preprocess(ctx.handle, frames_u8.data_ptr(), frames_u8.numel(), 0.5, 0.5, _lib.dtype_code(dtype), out.data_ptr()))
    return out


def _write_png16(path, depth_mm_u16):
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(depth_mm_u16, dtype=np.uint16)).save(path)  # uint16 -> 16-bit PNG


def estimate_depth_dpt(rgb_dataset, output_path: str, weights_filename='dpt_hybrid_nyu.pt', optimize=True, batch_size=8, dtype=None):
    """Estimate a depth map for every frame of ``rgb_dataset`` and write it as ``%06d.png`` (16-bit,
    millimetres) into ``output_path`` -- the side effect on disk is the contract
    (dataset_adaptors.py:1346-1435).  Frames are batched on the GPU (the reference runs them one by
    one); frames whose size is not the network's 640 x 480 go through the reference's resize rule and a
    nearest-neighbour resize of the prediction back to the frame size (:1421-1426).

    ``optimize=True`` is the reference's ``model.to(memory_format=channels_last); model.half()`` (:1394-1401): the network in
    float16 on the HIP engine (``dtype=torch.bfloat16`` selects the bfloat16 kernels instead: wider range, 3 bits less
    precision).  ``optimize=False`` is the reference's float32 network: PyTorch operators (``engine="torch"``), stated here, not
    a silent substitute -- the HIP engine computes in 16-bit types only.
    """
    weights_dir = os.environ.get('WEIGHTS_PATH', 'weights')
    model_path = os.path.join(weights_dir, weights_filename)
    if not os.path.isfile(model_path):
        model_path = os.path.join('weights', weights_filename)
    if not os.path.isfile(model_path):
        raise FileNotFoundError(f"DPT weights not found: {model_path} (set WEIGHTS_PATH)")
    if not torch.cuda.is_available():
        raise _lib.HiveError(_lib.ERR_DEVICE, "estimate_depth_dpt needs an MI355X; hive_amd has no CPU fallback")
    if optimize:
        dtype = dtype or torch.float16
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError(f"optimize=True runs the HIP engine in float16 or bfloat16, not {dtype}")
        model = build_model(model_path, dtype=dtype, engine="hip")
    else:
        dtype = None
        model = build_model(model_path, dtype=None, engine="torch")
    os.makedirs(output_path, exist_ok=True)
    transform = make_transform()
    n = len(rgb_dataset)
    # The PNG encoder (zlib, one core per file) is the slowest stage by far next to ~1 ms of GPU work per frame: the files of batch i are written by a small
    # thread pool while the GPU works on batch i + 1 (zlib releases the GIL); at most two batches of depth maps are in flight.
    from concurrent.futures import ThreadPoolExecutor
    from hive_amd.utils import usable_cores
    writers = ThreadPoolExecutor(max_workers=max(1, min(32, usable_cores() - 1)))  # (a 640 x 480 depth map takes 20-60 ms of one core to compress)
    pending = []

    def write_batch(start, depth_mm):
        return [writers.submit(_write_png16, os.path.join(output_path, f"{start + j:06d}.png"), depth_mm[j]) for j in range(len(depth_mm))]

    try:
        with torch.no_grad():
            for start in range(0, n, batch_size):
                images = [np.asarray(rgb_dataset[i]) for i in range(start, min(n, start + batch_size))]
                same = all(im.shape == images[0].shape and im.ndim == 3 and im.shape[2] == 3 and im.dtype == np.uint8 for im in images)
                if same and dtype is not None:
                    # uint8 frames of one size: resize (if any), normalisation, network, nearest resize back and the uint16-mm quantisation in ONE C-ABI call
                    frames = torch.from_numpy(np.stack(images)).cuda()
                    h, w = images[0].shape[:2]
                    _, mm, _ = model.forward_frames(frames, max_depth=65.535, net_size=network_size(h, w))
                    depth_mm = mm.cpu().numpy().view(np.uint16)
                else:  # optimize=False (PyTorch float32) or frames that are not uint8 RGB of one size: the host transform, frame by frame
                    preds = []
                    for im in images:
                        sample = torch.from_numpy(transform({"image": im / 255.0})["image"][None]).cuda().contiguous(memory_format=torch.channels_last)
                        if dtype is not None:
                            sample = sample.to(dtype)
                        prediction = model(sample)
                        if prediction.shape[-2:] != im.shape[:2]:
                            prediction = torch.nn.functional.interpolate(prediction.unsqueeze(1), size=im.shape[:2], mode="nearest").squeeze(1)
                        preds.append((prediction[0] * 1000.0).clamp(0, 65535).to(torch.int32).cpu().numpy().astype(np.uint16))
                    depth_mm = preds
                pending.append(write_batch(start, depth_mm))
                while len(pending) > 4:
                    for f in pending.pop(0):
                        f.result()
        for batch in pending:
            for f in batch:
                f.result()  # (re-raises a writer's exception)
    finally:
        writers.shutdown(wait=True)


class DepthFusionStream:
    """The hot path as one device-resident stream: uint8 frames in HBM -> DPT-Hybrid depth -> uint16-mm
    hand-off (same arithmetic as the PNG round trip, no file) -> TSDF integrate, batch by batch.

    ``accumulate=True`` integrates into the 5 accumulator planes instead of the running-average volume, for
    frame-sharded multi-GPU fusion (``hive_amd.distributed``).
    """

    def __init__(self, model, volume, cam_intr, max_depth=10.0, accumulate=False, native=True, overlap=False, net_size=None):
        """``overlap=True``: the TSDF sweeps of batch i run on a second HIP stream while the network already works on batch
        i + 1.  The network's MFMA kernels leave the vector ALUs idle and whole CUs idle in the tails of their tile rounds; the sweep
        is vector-ALU work with 8 KB of LDS.  ``volume`` must then live on a context of that stream
        (``side_stream_context()`` makes one); ``step`` returns the depth maps, and ``self.last_done`` is an event that fires when the
        sweeps of the MOST RECENT step are done (None before the first step and without overlap; overwritten by every step -- whoever
        recycles frame buffers takes it right after the call); ``join()`` orders the caller's stream behind all sweeps queued so far
        (before reading or merging the volume)."""
        self.model = model
        self.native = native  # run the network as one hive_dpt_forward_frames call (a 16-bit model on the HIP engine)
        self.net_size = None if net_size is None else (int(net_size[0]), int(net_size[1]))  # (net_h, net_w); None: the reference's rule for the frame size
        self.volume = volume
        self.K = np.ascontiguousarray(cam_intr, dtype=np.float32)
        self.max_depth = float(max_depth)
        self.dtype = next(model.parameters()).dtype
        self.accum = None
        self.side = None
        self.last_done = None
        if overlap:
            vctx = volume._ctx
            if vctx._follow_torch or not vctx._side_stream:
                raise ValueError("overlap=True needs a volume created on DepthFusionStream.side_stream_context(device)")
            self.side = vctx._side_stream
        if accumulate:
            self.accum = torch.empty(5 * volume.num_voxels, dtype=torch.float32, device="cuda")
            volume.accum_reset(self.accum)

    @staticmethod
    def side_stream_context(device=0):
        """A context (and its torch stream) for the volume of an overlapping stream: ``TSDFVolume(..., ctx=this)``."""
        ctx = _lib.Context(device, stream="own_low")  # lowest dispatch priority: the sweeps fill what the network leaves idle
        ctx.torch_stream()  # (creates ctx._side_stream, the torch view of that stream)
        return ctx

    def join(self):
        """Order the caller's current stream behind every sweep queued so far (no-op without overlap)."""
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)

    @torch.no_grad()
    def depth(self, frames_u8):
        """[B, H, W, 3] uint8 (GPU) -> (depth_m f32 [B, H, W] after the hand-off, depth_mm int16-viewed-as-uint16).  Frames of any size: the
        network runs at the reference's size for them (``network_size``: 480 x 640 frames as they are, 1080 x 1920 at 480 x 864), resizes on the device."""
        h, w = int(frames_u8.shape[1]), int(frames_u8.shape[2])
        net = self.net_size or network_size(h, w)
        if self.native and self.dtype in (torch.bfloat16, torch.float16) and getattr(self.model, "engine", None) == "hip":
            _, mm, m = self.model.forward_frames(frames_u8, max_depth=self.max_depth, net_size=net)  # ONE C-ABI call: hive_dpt_forward_frames
            return m, mm
        x = preprocess_on_device(frames_u8, self.dtype) if net == (h, w) else resize_preprocess_on_device(frames_u8, net, self.dtype)
        if net == (h, w):
            _, mm, m = self.model(x, handoff=(self.max_depth,))
            return m, mm
        _, mm, m = resize_depth_nearest(self.model(x), (h, w), self.max_depth)
        return m, mm

    @torch.no_grad()
    def step(self, frames_u8, poses_c2w, obs_weight=1.0):
        """One batch through the whole path; returns the depth maps that were integrated (with ``overlap``: that are being
        integrated on the side stream -- ``self.last_done`` fires when they are)."""
        depth_m, _ = self.depth(frames_u8)
        if self.side is not None:
            main = torch.cuda.current_stream()
            self.side.wait_stream(main)  # the depth maps (and the uploaded frames) are complete
            with torch.cuda.stream(self.side):
                self._integrate(frames_u8, depth_m, poses_c2w, obs_weight)
                self.last_done = torch.cuda.Event()
                self.last_done.record(self.side)
            for t in (frames_u8, depth_m):  # the caching allocator must not hand these to the main stream while the sweeps read them
                t.record_stream(self.side)
            return depth_m
        self._integrate(frames_u8, depth_m, poses_c2w, obs_weight)
        self.last_done = None
        return depth_m

    def _integrate(self, frames_u8, depth_m, poses_c2w, obs_weight):
        if self.accum is None:
            self.volume.integrate_batch(frames_u8, depth_m, self.K, poses_c2w, obs_weight=obs_weight)
        else:
            for i in range(frames_u8.shape[0]):
                self.volume.accum_integrate(self.accum, frames_u8[i], depth_m[i], self.K, poses_c2w[i], obs_weight=obs_weight)
