"""Frame-sharded multi-GPU fusion (new design -- the reference is single-process).

One process per GPU.  Frames are independent units for DPT, so rank r takes the contiguous block
``[r * T / W, (r + 1) * T / W)`` with no communication.  For the shared static-scene TSDF each rank
fuses its frames into its own volume and contributes the sums ``[num, w, r, g, b]`` (they commute, unlike
running averages): either converted from its volumes at the end (``fuse_sharded(volume)``) or accumulated
directly (``accum_integrate``); ONE all-reduce (RCCL over xGMI; ``backend="nccl"`` is RCCL on ROCm) over
the 5 N floats merges them -- issued as ONE reduce-scatter + ONE all-gather (``fuse_sharded``) -- after which every rank holds the
merged volume.

Parity (stated): GIVEN THE SAME DEPTH MAPS the merged tsdf equals the sequential running average up to float32 re-association
(abs 1e-5), weights are exact (small integers), colours differ by at most 2 levels (the sequential
reference rounds and clamps after every frame), the set of observed voxels is exact.

Caveat on the depth maps themselves: a frame's DPT depth is not bit-independent of the batch it was computed in.  Launches that do not
fill the chip split long K loops (another, fixed, order of float32 additions), the GroupNorm statistics of the bottlenecks' 1 x 1 convolutions switch
to their Gram-matrix form above a size threshold, both thresholds depend on the batch size and on the device's CU count, and a GroupNorm's partial
sums are cut at tile boundaries counted from the start of the batch.  Frame-sharded runs with other per-rank batch sizes than the one-GPU run
therefore integrate depth maps that differ from the one-GPU run's within the network's stated tolerance (median < 1 mm in float16, < 9 mm in
bfloat16: tools/diag_splitk_bound.py) before the merge's own tolerance applies.  ``Context.set_deterministic(True)`` (hive_ctx_set_deterministic)
removes the two threshold effects; the tile-boundary effect (~1e-7 relative in (mean, rstd)) remains unless the batch composition is kept.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise ``torch.distributed`` from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_ADDR / MASTER_PORT).  Returns (rank, world_size, local_rank).  Single process: (0, 1, 0), no group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:  # HIVE_DIST_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs than ranks
            backend = os.environ.get("HIVE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_range(num_items, rank, world):
    """Contiguous block of ``num_items`` owned by ``rank``: sizes differ by at most one, order preserved."""
    base, rem = divmod(num_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def allreduce_accumulators(accum, chunk_elems=1 << 28):
    """Sum the accumulator planes over all ranks, in place.  One logical all-reduce per sequence, issued
    in large chunks (1 GiB of float32) so that a 1024^3 volume (21 GB of planes) does not need one giant
    staging buffer; with 7 direct xGMI links per GPU large messages are what the links want."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return accum
    flat = accum.view(-1)
    for start in range(0, flat.numel(), chunk_elems):
        dist.all_reduce(flat[start:start + chunk_elems], op=dist.ReduceOp.SUM)
    return accum


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    """Max of a Python float over all ranks (for timing)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device="cpu" if _host_staged() else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _host_staged():
    """True when the process group cannot move device tensors itself (gloo rehearsals of the N > 1 path on a box with fewer GPUs
    than ranks, ``HIVE_DIST_BACKEND=gloo``): collectives then go through host copies.  RCCL (backend "nccl") never does."""
    return dist.is_initialized() and dist.get_backend() == "gloo"


def _collective():
    """Collectives are issued whenever a process group exists -- also a single-rank one (the one-GPU RCCL test runs the very
    calls the 8-GPU job makes); without a group the helpers below copy locally."""
    return dist.is_initialized()


def _reduce_scatter(out, inp):
    if _host_staged() and inp.is_cuda:
        o, i = out.cpu(), inp.cpu()
        dist.reduce_scatter_tensor(o, i, op=dist.ReduceOp.SUM)
        out.copy_(o)
    else:
        dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM)


def _all_gather(out, inp):
    if _host_staged() and inp.is_cuda:
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, inp.cpu())
        out.copy_(o)
    else:
        dist.all_gather_into_tensor(out, inp)


class VoxelPartition:
    """Equal, contiguous shares of a volume's N voxels over the ranks: rank r owns [r * chunk, r * chunk + count_of(r)).  Buffers
    that go through reduce-scatter / all-gather are laid out PIECE-major, ``[world][planes][chunk]``, so that each is ONE
    collective over one contiguous tensor (a rank's piece holds all planes of its share)."""

    def __init__(self, n_voxels, world=None, rank=None, align=256):
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank() if dist.is_initialized() else 0)
        self.n = int(n_voxels)
        per = -(-self.n // self.world)
        self.chunk = -(-per // align) * align
        self.padded = self.chunk * self.world
        self.first = self.rank * self.chunk
        self.count = self.count_of(self.rank)

    def count_of(self, rank):
        return max(0, min(self.chunk, self.n - rank * self.chunk))


def shard_layout(planes, part):
    """planes [P, N] -> [world, P, chunk] (piece-major, zero past N): the generic (torch) form of what
    ``hive_tsdf_accum_from_volume_sharded`` writes directly; used for raw accumulator tensors and by the CPU tests."""
    p = planes.shape[0]
    out = torch.zeros((part.world, p, part.chunk), dtype=planes.dtype, device=planes.device)
    for r in range(part.world):
        c = part.count_of(r)
        if c:
            out[r, :, :c] = planes[:, r * part.chunk:r * part.chunk + c]
    return out


def reduce_scatter_pieces(pieces, part):
    """pieces [world, P, chunk] -> this rank's summed share [P, chunk]: ONE reduce-scatter.  Each GPU sends (W - 1) / W of the
    buffer -- over xGMI's point-to-point links that is W - 1 concurrent transfers of 1 / W of the data each, instead of the two
    passes of an all-reduce."""
    assert pieces.dim() == 3 and pieces.shape[0] == part.world and pieces.shape[2] == part.chunk and pieces.is_contiguous()
    out = torch.empty(tuple(pieces.shape[1:]), dtype=pieces.dtype, device=pieces.device)
    if _collective():
        _reduce_scatter(out.view(-1), pieces.view(-1))
    else:
        out.copy_(pieces[0])
    return out


def all_gather_pieces(mine, part):
    """mine [P, chunk] (this rank's share of P result planes) -> [world, P, chunk] on every rank: ONE all-gather into a separate
    output buffer (no aliasing of input and output)."""
    assert mine.dim() == 2 and mine.shape[1] == part.chunk and mine.is_contiguous()
    out = torch.empty((part.world,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
    if _collective():
        _all_gather(out.view(-1), mine.view(-1))
    else:
        out[0].copy_(mine)
    return out


def fuse_sharded(volume, stream_or_accum=None):
    """Merge the per-rank fusions into the shared static-scene volume, on every rank, with TWO collectives:
    ONE reduce-scatter of the 5 accumulator planes (piece-major ``[world][5][chunk]``) -> every rank folds ITS 1 / W of the
    voxels (`accum_finalize_range`) -> ONE all-gather of the 3 result planes (``[world][3][chunk]``) -> the pieces are copied into
    place.  Against all-reduce + full finalize on every rank this moves 8 / 10 of the bytes ((5 + 3) N (W - 1) / W instead of
    2 x 5 N (W - 1) / W floats per GPU) and divides the finalize pass by W.

    Parity with the sequential one-GPU fusion of the SAME depth maps: tsdf <= 1e-5, weights exact, colours +-2 (module docstring; depth maps computed in
    other batch sizes than the one-GPU run's differ within the network's tolerance first).

    ``stream_or_accum=None`` (what ``bench.py --gpus N`` does): every rank fused its own frames with the ordinary
    ``integrate`` (same kernel and cost as on one GPU); its volumes are converted to the sums
    ``[tsdf * w, w, r * w, g * w, b * w]`` straight into the piece-major buffer.  Otherwise: the accumulators of a
    ``DepthFusionStream(accumulate=True)`` or a raw accumulator tensor filled with ``accum_integrate`` (sums of the raw
    observations, 40 B / voxel / frame)."""
    # a volume whose context does not follow torch's current stream (the side stream of an overlapping DepthFusionStream, Context(stream="own"), an
    # explicit hipStream_t) has its kernels on THAT stream: the collectives and the torch buffers go there too, in order with them
    side = volume._ctx.torch_stream()
    if side is not None:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            _fuse_sharded(volume, stream_or_accum)
        torch.cuda.current_stream().wait_stream(side)
        return volume
    return _fuse_sharded(volume, stream_or_accum)


def _fuse_sharded(volume, stream_or_accum):
    n = volume.num_voxels
    part = VoxelPartition(n)
    accum = getattr(stream_or_accum, "accum", stream_or_accum)
    if accum is None:
        pieces = torch.empty((part.world, 5, part.chunk), dtype=torch.float32, device="cuda")
        volume.accum_from_volume_sharded(pieces, part.world, part.chunk)
    else:
        pieces = shard_layout(accum.view(5, n), part)
    mine = reduce_scatter_pieces(pieces, part)  # [5, chunk]
    del pieces
    share = torch.zeros((3, part.chunk), dtype=torch.float32, device="cuda")  # tsdf, weight, colour of this rank's voxels
    volume.accum_finalize_range(mine, part.chunk, part.count, [share[0], share[1], share[2]])
    del mine
    gathered = all_gather_pieces(share, part)  # [world, 3, chunk]
    for r in range(part.world):
        volume.set_volume_range(r * part.chunk, part.count_of(r), tsdf=gathered[r, 0], weight=gathered[r, 1], color=gathered[r, 2])
    return volume


def allreduce_bounds(vol_bnds):
    """Union of every rank's scene bounds (3, 2): element-wise min of the lower and max of the upper corners -- exact in any order, so the
    frame-sharded run gets the bounds (and with them the voxel size and the grid) of the one-GPU run over the whole frame set."""
    import numpy as np
    vol_bnds = np.asarray(vol_bnds, np.float64)
    if not _collective():
        return vol_bnds
    device = "cpu" if _host_staged() else "cuda"
    lo = torch.tensor(vol_bnds[:, 0], dtype=torch.float64, device=device)
    hi = torch.tensor(vol_bnds[:, 1], dtype=torch.float64, device=device)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return np.stack([lo.cpu().numpy(), hi.cpu().numpy()], axis=1)


def tsdf_fusion_fg_bg_sharded(dataset, options=None, num_frames=-1, frame_set=None, instance_id=0, chunk_frames=None):
    """BASELINE config 5's multi-GPU form: the dynamic path (background volume = depth with the dilated instance masks zeroed, foreground volume
    = the complement; ``hive_amd.fusion.tsdf_fusion_fg_bg``) with the frames sharded over the ranks.  Every rank decodes and fuses ITS contiguous
    block of the frame set into its own pair of volumes -- on the grid of the whole set: the ranks' scene bounds are all-reduced (min / max, exact)
    before the volumes are created -- and each of the two volumes is merged once with ``fuse_sharded`` (reduce-scatter of its sums, all-gather of
    the result).  Returns {"bg": TSDFVolume, "fg": TSDFVolume}, the merged volumes, on every rank.  Parity with one GPU: as ``fuse_sharded``
    (tsdf <= 1e-5, weights exact, colours +-2) -- for the dataset's OWN depth maps, which this driver reads from disk; depth maps estimated per rank by
    the DPT network carry the batch-size caveat of this module's docstring on top."""
    import numpy as np
    from hive_amd import fusion
    from hive_amd.options import BackgroundMeshOptions
    options = options or BackgroundMeshOptions()
    frame_set = fusion._resolve_frames(dataset, num_frames, frame_set)
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_range(len(frame_set), rank, world)
    mine = list(frame_set[lo:hi])
    chunk = int(chunk_frames or fusion.CHUNK_FRAMES)
    local = np.zeros((3, 2))  # (the reference starts from zeros: the origin is always inside, hive/fusion.py:48)
    if mine:
        local = fusion.scene_bounds(fusion.frame_chunks(fusion._RawFrames(dataset), mine, False, chunk, with_color=False), dataset.camera_matrix)
    vol_bnds = allreduce_bounds(local)
    volumes = fusion.tsdf_fusion_fg_bg(dataset, options, frame_set=mine, instance_id=instance_id, chunk_frames=chunk_frames, vol_bnds=vol_bnds)
    for vol in volumes.values():
        fuse_sharded(vol)
    return volumes


# ------------------------------------------------------------------------------------------------
# Bit-exact mode (the alternative): the FRAMES are all-gathered (2.15 MB each: 323 MB for 150 VGA frames),
# the VOLUME is sharded in x-slabs, and every rank integrates every frame, in sequence order, into its slab.  A voxel's update
# reads only its own pixel, and a slab voxel keeps the world position it has in the whole grid (hive_tsdf_create_slab), so
# each slab is bit for bit the slice of the sequential single-GPU volume -- colours included, which the sum-based merge
# above cannot give.  Depth estimation stays frame-sharded (the expensive part); integrate work per rank is 1 / W of the
# voxels of every frame.
def allgather_frames(local, counts):
    """local: this rank's frames [n_r, ...] (contiguous block of the sequence, rank order = sequence order); counts[r] = n_r.
    Returns all frames [sum(counts), ...] in sequence order on every rank."""
    world = len(counts)
    if not _collective():
        assert world == 1
        return local
    most = max(counts)
    if local.shape[0] == most:
        padded = local.contiguous()
    else:
        padded = torch.zeros((most,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded[:local.shape[0]].copy_(local)
    gathered = torch.empty((world * most,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    _all_gather(gathered, padded)
    if all(c == most for c in counts):
        return gathered
    return torch.cat([gathered[r * most:r * most + counts[r]] for r in range(world)], dim=0)


def allgather_slabs(planes, x_ranges, row_elems):
    """planes: this rank's slab of P planes, [P, (x1 - x0) * row_elems]; x_ranges[r] = (x0, x1) of rank r.  ONE all-gather of
    piece-major ``[world][P][most]``; returns that buffer and the element count of every rank's slab."""
    world = len(x_ranges)
    counts = [(b - a) * row_elems for a, b in x_ranges]
    most = max(counts)
    p = planes.shape[0]
    mine = torch.zeros((p, most), dtype=planes.dtype, device=planes.device)
    mine[:, :planes.shape[1]].copy_(planes)
    out = torch.empty((world, p, most), dtype=planes.dtype, device=planes.device)
    if _collective():
        _all_gather(out.view(-1), mine.view(-1))
    else:
        assert world == 1
        out[0].copy_(mine)
    return out, counts


class ExactSlabFusion:
    """The bit-exact multi-GPU fusion on the MI355X: one x-slab ``TSDFVolume`` per rank."""

    def __init__(self, vol_bnds, voxel_size, ctx=None, **volume_kwargs):
        from hive_amd import fusion
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.vol_bnds, self.voxel_size, self._kwargs, self._ctx = vol_bnds, voxel_size, volume_kwargs, ctx
        dims = fusion.volume_dims(vol_bnds, voxel_size)
        self.dims = tuple(int(d) for d in dims)
        self.x_ranges = [shard_range(self.dims[0], r, self.world) for r in range(self.world)]
        self.slab = fusion.TSDFVolume(vol_bnds, voxel_size, ctx=ctx, x_range=self.x_ranges[self.rank], **volume_kwargs)

    def integrate(self, color_local, depth_local, cam_intr, poses_all, counts, obs_weight=1.0):
        """color_local u8 [n_r, H, W, 3] / depth_local f32 [n_r, H, W]: this rank's block of the sequence (device tensors,
        e.g. its DPT depth maps); poses_all [T, 4, 4] for the whole sequence; counts[r] = frames of rank r."""
        color = allgather_frames(color_local, counts)
        depth = allgather_frames(depth_local, counts)
        assert depth.shape[0] == len(poses_all)
        self.slab.integrate_batch(color, depth, cam_intr, poses_all, obs_weight=obs_weight)

    def gather(self):
        """The whole volume on every rank (for marching cubes): all-gather of the slabs' three planes."""
        from hive_amd import fusion
        row = self.dims[1] * self.dims[2]
        full = fusion.TSDFVolume(self.vol_bnds, self.voxel_size, ctx=self._ctx, **self._kwargs)
        gathered, counts = allgather_slabs(torch.stack(self.slab.device_tensors()), self.x_ranges, row)  # [world][tsdf, weight, colour][most]
        for r, (x0, _) in enumerate(self.x_ranges):
            full.set_volume_range(x0 * row, counts[r], tsdf=gathered[r, 0], weight=gathered[r, 1], color=gathered[r, 2])
        return full
