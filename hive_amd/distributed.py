"""Frame-sharded multi-GPU fusion (new design -- the reference is single-process).

One process per GPU.  Frames are independent units for DPT, so rank r takes the contiguous block
``[r * T / W, (r + 1) * T / W)`` with no communication.  For the shared static-scene TSDF each rank
fuses its frames into its own volume and contributes the sums ``[num, w, r, g, b]`` (they commute, unlike
running averages): either converted from its volumes at the end (``fuse_sharded(volume)``) or accumulated
directly (``accum_integrate``); ONE all-reduce (RCCL over xGMI; ``backend="nccl"`` is RCCL on ROCm) over
the 5 N floats merges them, after which every rank folds the sums into its volume.

Parity (stated): the merged tsdf equals the sequential running average up to float32 re-association
(abs 1e-5), weights are exact (small integers), colours differ by at most 2 levels (the sequential
reference rounds and clamps after every frame), the set of observed voxels is exact.
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
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def fuse_sharded(volume, stream_or_accum=None):
    """Merge the per-rank fusions into the shared static-scene volume, on every rank: ONE all-reduce of the
    accumulator planes, then ``finalize``.

    ``stream_or_accum=None`` (what ``bench.py --gpus N`` does): every rank fused its own frames with the ordinary
    ``integrate`` (same kernel and cost as on one GPU); its volumes are converted to the sums
    ``[tsdf * w, w, r * w, g * w, b * w]`` here.  Otherwise: the accumulators of a ``DepthFusionStream(accumulate=True)``
    or a raw accumulator tensor filled with ``accum_integrate`` (sums of the raw observations, 40 B / voxel / frame)."""
    accum = getattr(stream_or_accum, "accum", stream_or_accum)
    if accum is None:
        accum = torch.empty(5 * volume.num_voxels, dtype=torch.float32, device="cuda")
        volume.accum_from_volume(accum)
    allreduce_accumulators(accum)
    volume.accum_finalize(accum)
    return volume
