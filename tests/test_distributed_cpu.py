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
