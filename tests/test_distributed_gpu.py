"""The RCCL leg of the frame-sharded path on the one GPU a test box has: a single-rank ``nccl`` process group
(``backend="nccl"`` is RCCL on ROCm) runs the very collectives ``bench.py --gpus N`` issues -- the chunked in-place
all-reduce over views of the accumulator planes, the float64 MAX all-reduce of the timing and the barrier -- on
device memory, and the accumulate -> all-reduce -> finalize path must reproduce the sequential volume within the
tolerance stated in hive_amd/distributed.py.  (World size 2 is covered on CPU with gloo, tests/test_distributed_cpu.py.)"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_group():
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    yield dist
    dist.destroy_process_group()


def test_rccl_collectives_on_accumulator_views(gpu_ctx, nccl_group):
    import torch
    from hive_amd import distributed as hdist, fusion, synthetic
    dist = nccl_group
    seq = synthetic.make_sequence(num_frames=4, height=120, width=160, yaw_step_deg=20.0)
    bounds = synthetic.room_bounds()
    ref = fusion.TSDFVolume(bounds, 0.04, ctx=gpu_ctx)
    vol = fusion.TSDFVolume(bounds, 0.04, ctx=gpu_ctx)
    accum = torch.empty(5 * vol.num_voxels, dtype=torch.float32, device="cuda")  # as DepthFusionStream(accumulate=True) allocates it
    vol.accum_reset(accum)
    for i in range(4):
        ref.integrate(seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i])
        vol.accum_integrate(accum, seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i])
    before = accum.clone()
    # the exact calls of allreduce_accumulators (it returns early for one rank): in-place SUM over chunk views
    flat = accum.view(-1)
    chunk = 1 << 20
    for start in range(0, flat.numel(), chunk):
        dist.all_reduce(flat[start:start + chunk], op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    assert torch.equal(accum, before)  # one rank: the sum is the identity, bit for bit
    hdist.barrier()
    assert hdist.max_over_ranks(1.25, device=torch.device("cuda", 0)) == 1.25
    hdist.fuse_sharded(vol, accum)
    t_ref, c_ref = ref.get_volume()
    t_acc, c_acc = vol.get_volume()
    assert np.abs(t_ref - t_acc).max() <= 1e-5
    observed = t_ref != 1.0
    assert observed.any() and np.array_equal(observed, t_acc != 1.0)
    for sh in (0, 8, 16):
        a = (c_ref.astype(np.int64) >> sh) & 255
        b = (c_acc.astype(np.int64) >> sh) & 255
        assert np.abs(a - b).max() <= 2
    # the other contribution mode (bench.py --gpus N): volumes -> sums on the device, bit-exact against the numpy
    # statement of the same products, and with one rank fuse_sharded(volume) must give the volume back
    import oracle
    planes = torch.empty(5 * ref.num_voxels, dtype=torch.float32, device="cuda")
    ref.accum_from_volume(planes)
    ora = oracle.TSDFVolume(bounds, 0.04)
    for i in range(4):
        ora.integrate(seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i])
    assert np.array_equal(ora._tsdf, t_ref) and np.array_equal(ora._color, c_ref)
    assert np.array_equal(planes.cpu().numpy().reshape(5, *t_ref.shape), oracle.AccumVolume.planes_from_volume(ora))
    # the piece-major form the merge sends (hive_tsdf_accum_from_volume_sharded) == the generic re-layout of those planes, for a
    # partition into 3 uneven pieces
    part3 = hdist.VoxelPartition(ref.num_voxels, world=3, rank=0, align=256)
    pieces = torch.full((3, 5, part3.chunk), float("nan"), dtype=torch.float32, device="cuda")
    ref.accum_from_volume_sharded(pieces, 3, part3.chunk)
    assert torch.equal(pieces, hdist.shard_layout(planes.view(5, -1), part3))
    hdist.fuse_sharded(ref)
    t2, c2 = ref.get_volume()
    assert np.abs(t2 - t_ref).max() <= 1e-6 and np.array_equal(c2, c_ref)


def test_rccl_runs_the_merge_collectives(gpu_ctx, nccl_group, monkeypatch):
    """The collectives of both merge modes are ISSUED on RCCL with device tensors even at world size 1 (no short-circuit while a
    process group exists): reduce_scatter_tensor + all_gather_into_tensor of `fuse_sharded`, and the all-gathers of
    `ExactSlabFusion` (frames, slabs) -- counted by wrapping torch.distributed's entry points -- and the results are exact."""
    import torch
    import torch.distributed as tdist
    from hive_amd import distributed as hdist, fusion, synthetic
    calls = {"reduce_scatter_tensor": 0, "all_gather_into_tensor": 0}
    for name in calls:
        orig = getattr(tdist, name)

        def wrapped(out, inp, *a, _orig=orig, _name=name, **k):
            assert out.is_cuda and inp.is_cuda and out.data_ptr() != inp.data_ptr(), "collectives run on separate device buffers"
            calls[_name] += 1
            return _orig(out, inp, *a, **k)
        monkeypatch.setattr(tdist, name, wrapped)
    seq = synthetic.make_sequence(num_frames=6, height=120, width=160, yaw_step_deg=6.0)
    color_d, depth_d = torch.from_numpy(seq["color"]).cuda(), torch.from_numpy(seq["depth"]).cuda()
    bounds = synthetic.room_bounds()
    ref = fusion.TSDFVolume(bounds, 0.04, ctx=gpu_ctx)
    ref.integrate_batch(color_d, depth_d, seq["K"], seq["poses"])
    want = [t.clone() for t in ref.device_tensors()]
    # sum mode: one rank's volume through reduce-scatter -> finalize -> all-gather must come back (tsdf * w / w: within one ulp)
    hdist.fuse_sharded(ref)
    assert calls == {"reduce_scatter_tensor": 1, "all_gather_into_tensor": 1}
    got = ref.device_tensors()
    assert torch.equal(got[1], want[1]) and torch.equal(got[2], want[2]) and float((got[0] - want[0]).abs().max()) <= 1e-6
    # exact mode: frames all-gathered (2 calls), slabs all-gathered (1 call), bit-identical
    fus = hdist.ExactSlabFusion(bounds, 0.04, ctx=gpu_ctx)
    fus.integrate(color_d, depth_d, seq["K"], seq["poses"], [6])
    full = fus.gather()
    assert calls == {"reduce_scatter_tensor": 1, "all_gather_into_tensor": 4}
    assert fus.slab.last_batch_groups() == [4, 2]
    for a, b in zip(full.device_tensors(), want):
        assert torch.equal(a, b)


def test_rccl_merge_of_a_volume_on_the_overlap_side_stream(gpu_ctx, nccl_group):
    """The combination the default `bench.py --gpus N` job uses at N > 1: the volume lives on the lowest-priority side stream of an overlapping
    DepthFusionStream (`side_stream_context`), its sweeps are queued there behind depth maps made on torch's stream, and `fuse_sharded` must issue its
    RCCL collectives (and its torch buffers) on THAT stream, in order with the sweeps -- no synchronize in between.  Single-rank RCCL group: the very calls
    of the 8-GPU job.  The merged volume must equal the same frames fused on the ordinary stream (weights and colours exactly, tsdf to tsdf * w / w)."""
    import torch
    from hive_amd import depth as depth_mod, distributed as hdist, fusion, synthetic
    seq = synthetic.make_sequence(num_frames=8, height=120, width=160, yaw_step_deg=4.0)
    color_d, depth_d = torch.from_numpy(seq["color"]).cuda(), torch.from_numpy(seq["depth"]).cuda()
    bounds = synthetic.room_bounds()
    ref = fusion.TSDFVolume(bounds, 0.04, ctx=gpu_ctx)
    ref.integrate_batch(color_d, depth_d, seq["K"], seq["poses"])
    want = [t.clone() for t in ref.device_tensors()]
    vctx = depth_mod.DepthFusionStream.side_stream_context(0)
    side = vctx.torch_stream()
    vol = fusion.TSDFVolume(bounds, 0.04, ctx=vctx)
    main = torch.cuda.current_stream()
    for rep in range(3):  # (repeated: an ordering bug between the streams shows as a volume merged before its sweeps landed)
        vol.reset()
        scaled = depth_d * 1.0  # produced on torch's stream just before the hand-over, as a network's depth maps are
        side.wait_stream(main)
        with torch.cuda.stream(side):
            vol.integrate_batch(color_d, scaled, seq["K"], seq["poses"])
        scaled.record_stream(side)
        hdist.fuse_sharded(vol)  # no synchronize: the merge must order itself behind the sweeps
        got = vol.device_tensors()
        torch.cuda.synchronize()
        assert torch.equal(got[1], want[1]) and torch.equal(got[2], want[2]) and float((got[0] - want[0]).abs().max()) <= 1e-6, rep
    assert vol.stats()[0] == 8


# ---- BASELINE config 5's multi-GPU form: the dynamic path with the frames sharded over TWO ranks (both on this box's one GPU, collectives
# through gloo: what is exercised is the sharding, the all-reduced bounds and the two merges, not RCCL) --------------------------------------
class _ShardedFake:
    """What tsdf_fusion_fg_bg reads of a HiveDataset, built from the seeded synthetic sequence in every rank."""

    def __init__(self, n):
        from hive_amd import synthetic
        from hive_amd.geometric import Trajectory
        self.seq = synthetic.make_sequence(num_frames=n, height=60, width=80, yaw_step_deg=25.0)
        self.masks = list(synthetic.ellipse_masks(n, 60, 80, num_objects=2, seed=5))
        self.num_frames = n
        self.camera_matrix = self.seq["K"].astype(np.float64)
        self.camera_trajectory = Trajectory(synthetic.trajectory_rows_world_to_cam(self.seq["poses"]).astype(np.float64))
        self.has_inpainted_frame_data = False
        self.mask_dataset = self.masks
        self.rgb_dataset = self.bg_rgb_dataset = [c.copy() for c in self.seq["color"]]
        self.depth_dataset = self.bg_depth_dataset = [d.copy() for d in self.seq["depth"]]


def _fg_bg_worker(rank, world, port, out_path):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HIVE_DIST_BACKEND="gloo")
    import torch
    from hive_amd import distributed as hdist
    from hive_amd.options import BackgroundMeshOptions
    hdist.init_from_env()
    torch.cuda.set_device(0)
    options = BackgroundMeshOptions(sdf_voxel_size=0.05, sdf_max_voxels=400_000, depth_mask_dilation_iterations=2)
    vols = hdist.tsdf_fusion_fg_bg_sharded(_ShardedFake(7), options)  # 7 frames: blocks of 4 + 3
    if rank == 0:
        out = {}
        for name, vol in vols.items():
            t, c, w = vol.get_volume(with_weight=True)
            out[name + "_tsdf"], out[name + "_color"], out[name + "_weight"] = t, c, w
            out[name + "_bnds"] = vol._vol_bnds
        np.savez(out_path, **out)
    hdist.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_dynamic_path_frame_sharded(gpu_ctx, tmp_path):
    """`tsdf_fusion_fg_bg_sharded` on two ranks == `tsdf_fusion_fg_bg` on one (the same grid: bounds all-reduced exactly; weights exact,
    tsdf <= 1e-5, colours +-2: the merge adds the ranks' sums where one GPU rounds after every frame)."""
    import socket
    import torch.multiprocessing as mp
    from hive_amd import fusion
    from hive_amd.options import BackgroundMeshOptions
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "sharded.npz")
    mp.start_processes(_fg_bg_worker, args=(2, port, out), nprocs=2, join=True, start_method="spawn")
    got = np.load(out)
    options = BackgroundMeshOptions(sdf_voxel_size=0.05, sdf_max_voxels=400_000, depth_mask_dilation_iterations=2)
    ref = fusion.tsdf_fusion_fg_bg(_ShardedFake(7), options)
    for name in ("bg", "fg"):
        t, c, w = ref[name].get_volume(with_weight=True)
        assert np.array_equal(got[name + "_bnds"], ref[name]._vol_bnds) and got[name + "_tsdf"].shape == t.shape, name
        assert np.array_equal(got[name + "_weight"], w), f"{name}: weights (and the observed set) must be exact"
        assert np.abs(got[name + "_tsdf"] - t).max() <= 1e-5, name
        for sh in (0, 8, 16):
            a, b = (got[name + "_color"].astype(np.int64) >> sh) & 255, (c.astype(np.int64) >> sh) & 255
            assert np.abs(a - b).max() <= 2, name
    assert got["fg_weight"].max() > 0 and (got["bg_weight"] > 0).sum() > (got["fg_weight"] > 0).sum()
