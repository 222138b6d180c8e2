"""N > 1 path on CPU: two gloo ranks shard the frames, accumulate with the CPU oracle, all-reduce the
accumulator planes through ``hive_amd.distributed`` and finalize; the result must match the sequential
reference semantics within the tolerance stated in DESIGN.md §7 (tsdf 1e-5, weight exact, colour +-2,
observed-voxel set exact)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path, mode="accumulate"):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import oracle
    from hive_amd import distributed as hdist, synthetic
    r, w, _ = hdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    seq = synthetic.make_sequence(num_frames=8, height=60, width=80, yaw_step_deg=45.0)
    lo, hi = hdist.shard_range(8, rank, world)
    acc = oracle.AccumVolume(synthetic.room_bounds(), 0.16)
    if mode == "accumulate":  # sums of the raw observations (accum_integrate)
        for i in range(lo, hi):
            acc.integrate(seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i])
    else:  # "volume": the rank fuses its shard the ordinary way, then converts its volumes to sums (fuse_sharded(volume))
        own = oracle.TSDFVolume(synthetic.room_bounds(), 0.16)
        for i in range(lo, hi):
            own.integrate(seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i])
        acc.accum[...] = oracle.AccumVolume.planes_from_volume(own)
    t = torch.from_numpy(acc.accum)
    hdist.allreduce_accumulators(t, chunk_elems=10_000)  # several chunks on purpose
    hdist.barrier()
    vol = acc.finalize()
    elapsed = hdist.max_over_ranks(float(rank + 1))
    assert elapsed == float(world)
    if rank == 0:
        np.savez(out_path, tsdf=vol._tsdf, weight=vol._weight, color=vol._color)
    torch.distributed.destroy_process_group()


def test_shard_range_partitions():
    from hive_amd.distributed import shard_range
    for n in (0, 1, 7, 150, 151):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("mode", ["accumulate", "volume"])
def test_two_rank_frame_sharded_fusion_matches_sequential(tmp_path, oracle_lib, mode):
    import torch.multiprocessing as mp
    from hive_amd import synthetic
    out = str(tmp_path / "rank0.npz")
    mp.start_processes(_worker, args=(2, _free_port(), out, mode), nprocs=2, join=True, start_method="spawn")
    got = np.load(out)
    seq = synthetic.make_sequence(num_frames=8, height=60, width=80, yaw_step_deg=45.0)
    ref = oracle_lib.TSDFVolume(synthetic.room_bounds(), 0.16)
    for i in range(8):
        ref.integrate(seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i])
    assert np.array_equal(got["weight"], ref._weight), "weights (and the observed set) must be exact"
    np.testing.assert_allclose(got["tsdf"], ref._tsdf, rtol=0, atol=1e-5)

    def unpack(c):
        c = c.astype(np.int64)
        return np.stack([c & 255, (c >> 8) & 255, c >> 16], axis=-1)
    assert np.abs(unpack(got["color"]) - unpack(ref._color)).max() <= 2


def test_single_process_helpers_are_noops():
    import torch
    from hive_amd import distributed as hdist
    t = torch.arange(10, dtype=torch.float32)
    assert hdist.allreduce_accumulators(t) is t
    assert hdist.max_over_ranks(3.5) == 3.5
    hdist.barrier()


def _worker_reduce_scatter(rank, world, port, out_path):
    """fuse_sharded's collective shape with the oracle standing in for the HIP kernels: piece-major planes -> ONE reduce-scatter
    -> fold own share -> ONE all-gather of the three result planes."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import ctypes
    import torch
    import oracle
    from hive_amd import distributed as hdist, synthetic
    hdist.init_from_env(backend="gloo")
    seq = synthetic.make_sequence(num_frames=8, height=60, width=80, yaw_step_deg=45.0)
    lo, hi = hdist.shard_range(8, rank, world)
    own = oracle.TSDFVolume(synthetic.room_bounds(), 0.16, use_gpu=False)
    for i in range(lo, hi):
        own.integrate(seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i])
    n = own._tsdf.size
    part = hdist.VoxelPartition(n, align=64)
    assert part.padded >= n and part.padded % world == 0 and part.chunk % 64 == 0
    # piece-major [world][5][chunk]: what hive_tsdf_accum_from_volume_sharded writes on the GPU
    pieces = hdist.shard_layout(torch.from_numpy(oracle.AccumVolume.planes_from_volume(own).reshape(5, n)), part)
    mine = hdist.reduce_scatter_pieces(pieces, part)  # ONE reduce-scatter
    assert mine.shape == (5, part.chunk)
    share = [np.zeros(part.chunk, np.float32) for _ in range(3)]
    acc = np.ascontiguousarray(mine.numpy())
    oracle.lib().oracle_tsdf_accum_finalize(acc.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(part.chunk), share[0].ctypes.data_as(ctypes.c_void_p),
                                            share[1].ctypes.data_as(ctypes.c_void_p), share[2].ctypes.data_as(ctypes.c_void_p), own.round_mode)
    gathered = hdist.all_gather_pieces(torch.from_numpy(np.stack(share)), part)  # ONE all-gather -> [world][3][chunk]
    assert gathered.shape == (world, 3, part.chunk)
    outs = [torch.cat([gathered[r, k, :part.count_of(r)] for r in range(world)]) for k in range(3)]
    assert all(o.numel() == n for o in outs)
    if rank == 0:
        np.savez(out_path, tsdf=outs[0][:n].numpy(), weight=outs[1][:n].numpy(), color=outs[2][:n].numpy())
    torch.distributed.destroy_process_group()


def test_two_rank_reduce_scatter_merge_matches_sequential(tmp_path, oracle_lib):
    """reduce-scatter -> fold own share -> all-gather == what the all-reduce merge gives (bit for bit at two ranks: a sum of two
    terms has one order), and the sequential reference within the stated tolerance."""
    import torch.multiprocessing as mp
    from hive_amd import synthetic
    out = str(tmp_path / "rank0.npz")
    mp.start_processes(_worker_reduce_scatter, args=(2, _free_port(), out), nprocs=2, join=True, start_method="spawn")
    got = np.load(out)
    seq = synthetic.make_sequence(num_frames=8, height=60, width=80, yaw_step_deg=45.0)
    ref = oracle_lib.TSDFVolume(synthetic.room_bounds(), 0.16, use_gpu=False)
    halves = [oracle_lib.TSDFVolume(synthetic.room_bounds(), 0.16, use_gpu=False) for _ in range(2)]
    for i in range(8):
        ref.integrate(seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i])
        halves[i // 4].integrate(seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i])
    merged = oracle_lib.AccumVolume(synthetic.room_bounds(), 0.16, use_gpu=False)
    merged.accum[...] = oracle_lib.AccumVolume.planes_from_volume(halves[0]) + oracle_lib.AccumVolume.planes_from_volume(halves[1])
    exp = merged.finalize()
    assert np.array_equal(got["tsdf"], exp._tsdf.reshape(-1)) and np.array_equal(got["color"], exp._color.reshape(-1))
    assert np.array_equal(got["weight"], ref._weight.reshape(-1)), "weights (and the observed set) must be exact"
    np.testing.assert_allclose(got["tsdf"], ref._tsdf.reshape(-1), rtol=0, atol=1e-5)


def _worker_exact(rank, world, port, out_path):
    """The bit-exact mode's collective shape: frames all-gathered (uneven blocks), x-slabs (uneven), slabs all-gathered."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import oracle
    from hive_amd import distributed as hdist, synthetic
    hdist.init_from_env(backend="gloo")
    T = 7
    seq = synthetic.make_sequence(num_frames=T, height=60, width=80, yaw_step_deg=50.0)
    counts = [b - a for a, b in (hdist.shard_range(T, r, world) for r in range(world))]
    lo, hi = hdist.shard_range(T, rank, world)
    # this rank only HAS its own block of frames (on the GPU: the depth maps its DPT shard produced)
    color = hdist.allgather_frames(torch.from_numpy(seq["color"][lo:hi].copy()), counts)
    depth = hdist.allgather_frames(torch.from_numpy(seq["depth"][lo:hi].copy()), counts)
    assert color.shape[0] == T and depth.shape[0] == T
    bounds, voxel = np.array([[0.0, 5.12], [0.0, 5.12], [0.0, 5.12]]), 0.155  # 34 x 34 x 34: slabs of 17 x-rows
    full = oracle.TSDFVolume(bounds, voxel)
    X, Y, Z = (int(d) for d in full._vol_dim)
    x_ranges = [hdist.shard_range(X, r, world) for r in range(world)]
    x0, x1 = x_ranges[rank]
    for i in range(T):  # the oracle has no slab form: integrate the whole grid, KEEP only this rank's slab (voxels are independent)
        full.integrate(color[i].numpy(), depth[i].numpy(), seq["K"], seq["poses"][i])
    slabs = [torch.from_numpy(np.ascontiguousarray(a[x0:x1])) for a in (full._tsdf, full._weight, full._color)]
    poison = [torch.full_like(s, float("nan")) for s in slabs]  # what the other ranks hold is NOT available here
    del full
    pieces, counts = hdist.allgather_slabs(torch.stack([s.reshape(-1) for s in slabs]), x_ranges, Y * Z)  # ONE all-gather: [world][3][most]
    assert counts == [(b - a) * Y * Z for a, b in x_ranges]
    gathered = [torch.cat([pieces[r, k, :counts[r]] for r in range(world)]) for k in range(3)]
    assert all(g.numel() == X * Y * Z for g in gathered) and not any(torch.isnan(g).any() for g in gathered) and len(poison) == 3
    if rank == 1:
        np.savez(out_path, tsdf=gathered[0].numpy(), weight=gathered[1].numpy(), color=gathered[2].numpy(), dims=np.array([X, Y, Z]))
    torch.distributed.destroy_process_group()


def test_two_rank_exact_slab_mode_is_bit_identical_to_sequential(tmp_path, oracle_lib):
    import torch.multiprocessing as mp
    from hive_amd import synthetic
    out = str(tmp_path / "rank1.npz")
    mp.start_processes(_worker_exact, args=(2, _free_port(), out), nprocs=2, join=True, start_method="spawn")
    got = np.load(out)
    seq = synthetic.make_sequence(num_frames=7, height=60, width=80, yaw_step_deg=50.0)
    ref = oracle_lib.TSDFVolume(np.array([[0.0, 5.12], [0.0, 5.12], [0.0, 5.12]]), 0.155)
    for i in range(7):
        ref.integrate(seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i])
    assert tuple(got["dims"]) == tuple(int(d) for d in ref._vol_dim)
    for name, arr in (("tsdf", ref._tsdf), ("weight", ref._weight), ("color", ref._color)):
        assert np.array_equal(got[name], arr.reshape(-1)), f"{name}: the slab mode must be BIT-identical to the sequential fusion"


def _bounds_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    from hive_amd import distributed as hdist
    hdist.init_from_env(backend="gloo")
    local = np.array([[-1.5 - rank, 0.25 + rank], [0.0, 2.0 - 3.0 * rank], [-0.125 * (rank + 1), 7.0]])
    merged = hdist.allreduce_bounds(local)
    if rank == 0:
        np.save(out_path, merged)
    hdist.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_scene_bounds_union(tmp_path):
    """`allreduce_bounds` (the frame-sharded dynamic path, hive_amd.distributed.tsdf_fusion_fg_bg_sharded): element-wise min / max over the ranks, exact."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "bounds.npy")
    mp.start_processes(_bounds_worker, args=(2, _free_port(), out), nprocs=2, join=True, start_method="spawn")
    assert np.array_equal(np.load(out), np.array([[-2.5, 1.25], [0.0, 2.0], [-0.25, 7.0]]))


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the driver's plain command form) must start two fresh rank processes itself --
    before anything touches the GPU -- and hand their exit status on.  Without an MI355X both ranks get as far as the rendezvous (gloo) and then refuse to
    run (no CPU fallback): two refusals, a non-zero status, and not the old "launch with torch.distributed.run" exit."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HIVE_DIST_BACKEND"] = "gloo"
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""  # (also on a GPU box: this test is about the launch, not the job)
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True,
                         timeout=300)
    out = run.stdout + run.stderr
    assert run.returncode != 0
    assert out.count("bench.py needs an MI355X") == 2, out[-2000:]
    assert "launch with torch.distributed.run" not in out
