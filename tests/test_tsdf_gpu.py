"""Parity of the HIP TSDF path (through the C ABI) against the CPU oracle: bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _volumes_equal(vol, ora):
    tsdf, color, weight = vol.get_volume(with_weight=True)
    assert np.array_equal(weight, ora._weight), "weight volume differs"
    assert np.array_equal(color, ora._color), "colour volume differs"
    assert np.array_equal(tsdf, ora._tsdf), "tsdf volume differs"


@pytest.mark.parametrize("round_mode", [0, 1])
@pytest.mark.parametrize("voxel", [0.08, 0.0641])  # 64^3 (vector path) and 80^3 with Z % 4 == 0; see below for odd Z
def test_integrate_bit_exact(gpu_ctx, oracle_lib, small_sequence, round_mode, voxel):
    from hive_amd import fusion, synthetic
    seq = small_sequence
    vol = fusion.TSDFVolume(synthetic.room_bounds(), voxel, ctx=gpu_ctx, round_mode=round_mode)
    ora = oracle_lib.TSDFVolume(synthetic.room_bounds(), voxel, round_mode=round_mode)
    assert np.array_equal(vol._vol_dim, ora._vol_dim)
    assert np.array_equal(vol._vol_origin, ora._vol_origin)
    for i in range(seq["depth"].shape[0]):
        n = vol.integrate(seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i], return_n_updated=True)
        ora.integrate(seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i])
        assert n == ora.last_n_updated, f"frame {i}: N_upd {n} != oracle {ora.last_n_updated}"
    _volumes_equal(vol, ora)


def test_integrate_odd_dims_row_tails(gpu_ctx, oracle_lib, small_sequence):
    """Z not a multiple of 4: rows are not 16-byte aligned and the last quad of every row reaches into the next row -- its tail
    is excluded from the tests and from the access (the volumes data-dependent bounds give, hive/fusion.py:37-76, are rarely
    multiples of 4); ragged X / Y / Z, an observation weight of 2."""
    from hive_amd import fusion
    seq = small_sequence
    bnds = np.array([[0.3, 4.9], [0.0, 5.12], [0.1, 5.0]])
    vol = fusion.TSDFVolume(bnds, 0.07, ctx=gpu_ctx)
    ora = oracle_lib.TSDFVolume(bnds, 0.07)
    assert vol._vol_dim[2] % 4 != 0
    for i in range(4):
        n = vol.integrate(seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i], obs_weight=2.0, return_n_updated=True)
        ora.integrate(seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i], obs_weight=2.0)
        assert n == ora.last_n_updated
    _volumes_equal(vol, ora)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_integrate_random_poses_weights_and_ragged_images(gpu_ctx, oracle_lib, seed):
    """Random camera poses (incl. inside / outside / behind the volume), non-integer observation weights,
    image sizes that are not multiples of 4 (scalar pack kernel), noisy depth with zeros: exercises the
    shared-reciprocal divisions and every inclusion test against the oracle, bit for bit."""
    from scipy.spatial.transform import Rotation
    from hive_amd import fusion
    rng = np.random.default_rng(seed)
    H, W = [(37, 53), (48, 64), (61, 45)][seed]
    K = np.array([[40.0 + 7 * seed, 0, W / 2 - 0.5], [0, 42.0, H / 2 - 0.5], [0, 0, 1]], np.float32)
    bnds = np.array([[-1.0, 1.1], [-0.8, 1.0], [0.2, 2.6]])
    voxel = [0.03, 0.05, 0.041][seed]
    vol = fusion.TSDFVolume(bnds, voxel, ctx=gpu_ctx)
    ora = oracle_lib.TSDFVolume(bnds, voxel)
    for f in range(12):
        pose = np.eye(4)
        pose[:3, :3] = Rotation.from_rotvec(rng.normal(scale=0.6, size=3)).as_matrix()
        pose[:3, 3] = rng.uniform(-1.5, 1.5, 3) + np.array([0, 0, -0.5])
        depth = rng.uniform(0.2, 4.0, (H, W)).astype(np.float32)
        depth[rng.random((H, W)) < 0.15] = 0.0
        color = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        w = float(np.float32(rng.choice([1.0, 0.37, 2.5, 1e-3, 123.456])))
        n = vol.integrate(color, depth, K, pose, obs_weight=w, return_n_updated=True)
        ora.integrate(color, depth, K, pose, obs_weight=w)
        assert n == ora.last_n_updated
    assert ora._weight.max() > 0, "scene never observed"
    _volumes_equal(vol, ora)


def test_full_size_512_vga(gpu_ctx, oracle_lib):
    """BASELINE.json's full size (640 x 480 into 512^3): one frame bit-exact against the C oracle, then
    size-independent properties -- integrating the same observation again leaves tsdf and colour unchanged
    and doubles the weights (running average of equal values), N_upd is the number of weighted voxels,
    and a checksum of the volume is reproducible across an independent second volume."""
    from hive_amd import fusion, synthetic
    seq = synthetic.make_sequence(num_frames=2, yaw_step_deg=40.0)
    vol = fusion.TSDFVolume(synthetic.room_bounds(), 0.01, ctx=gpu_ctx)
    assert tuple(vol.vol_dim) == (512, 512, 512)
    n1 = vol.integrate(seq["color"][0], seq["depth"][0], seq["K"], seq["poses"][0], return_n_updated=True)
    tsdf, color, weight = vol.get_volume(with_weight=True)
    ora = oracle_lib.TSDFVolume(synthetic.room_bounds(), 0.01)
    ora.integrate(seq["color"][0], seq["depth"][0], seq["K"], seq["poses"][0])
    assert n1 == ora.last_n_updated == int(np.count_nonzero(weight))
    assert np.array_equal(weight, ora._weight) and np.array_equal(tsdf, ora._tsdf) and np.array_equal(color, ora._color)
    del ora
    n2 = vol.integrate(seq["color"][0], seq["depth"][0], seq["K"], seq["poses"][0], return_n_updated=True)
    tsdf2, color2, weight2 = vol.get_volume(with_weight=True)
    assert n2 == n1
    assert np.array_equal(tsdf2, tsdf) and np.array_equal(color2, color) and np.array_equal(weight2, 2 * weight)
    # second frame into two independent volumes (one via the batch entry point): identical bits
    vol.integrate(seq["color"][1], seq["depth"][1], seq["K"], seq["poses"][1])
    other = fusion.TSDFVolume(synthetic.room_bounds(), 0.01, ctx=gpu_ctx)
    idx = [0, 0, 1]
    other.integrate_batch(seq["color"][idx], seq["depth"][idx], seq["K"], seq["poses"][idx])
    a, b = vol.get_volume(with_weight=True), other.get_volume(with_weight=True)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    verts, faces, norms, colors = vol.get_mesh()
    assert len(verts) > 10000 and faces.max() == len(verts) - 1 and faces.min() == 0
    assert np.isfinite(verts).all() and np.abs(np.linalg.norm(norms, axis=1) - 1).max() < 1e-4


def test_full_size_1024_cubed_1080p(gpu_ctx, oracle_lib):
    """BASELINE config 4's TSDF side: a 1920 x 1080 frame into a 1024^3 volume (5 mm voxels, 12.9 GB of volumes in HBM).
    One frame bit-exact against the C oracle (26 GB of host arrays), then the size-independent properties on the device:
    the same observation again leaves tsdf / colour unchanged and doubles the weights; N_upd = weighted voxels."""
    import torch
    from hive_amd import fusion, synthetic
    seq = synthetic.make_sequence(num_frames=1, height=1080, width=1920, yaw_step_deg=40.0)
    vol = fusion.TSDFVolume(synthetic.room_bounds(), 0.005, ctx=gpu_ctx)
    assert tuple(vol.vol_dim) == (1024, 1024, 1024) and vol.num_voxels == 2 ** 30
    args = (seq["color"][0], seq["depth"][0], seq["K"], seq["poses"][0])
    n1 = vol.integrate(*args, return_n_updated=True)
    ora = oracle_lib.TSDFVolume(synthetic.room_bounds(), 0.005)
    ora.integrate(*args)
    assert n1 == ora.last_n_updated and n1 > 50_000_000
    tsdf, color, weight = vol.get_volume(with_weight=True)
    assert np.array_equal(weight, ora._weight) and np.array_equal(tsdf, ora._tsdf) and np.array_equal(color, ora._color)
    assert int(np.count_nonzero(weight)) == n1
    del ora, tsdf, color, weight
    t1, w1, c1 = vol.device_tensors()
    n2 = vol.integrate(*args, return_n_updated=True)
    t2, w2, c2 = vol.device_tensors()
    assert n2 == n1 and torch.equal(t1, t2) and torch.equal(c1, c2) and torch.equal(w2, 2 * w1)


def test_integrate_batch_and_device_inputs(gpu_ctx, oracle_lib, small_sequence):
    import torch
    from hive_amd import fusion, synthetic
    seq = small_sequence
    ora = oracle_lib.TSDFVolume(synthetic.room_bounds(), 0.08)
    for i in range(8):
        ora.integrate(seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i])
    # host batch
    vol = fusion.TSDFVolume(synthetic.room_bounds(), 0.08, ctx=gpu_ctx)
    vol.integrate_batch(seq["color"], seq["depth"], seq["K"], seq["poses"])
    _volumes_equal(vol, ora)
    # device tensors, caller-owned storage
    n = int(np.prod(ora._vol_dim))
    storage = tuple(torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(3))
    vol2 = fusion.TSDFVolume(synthetic.room_bounds(), 0.08, ctx=gpu_ctx, storage=storage)
    color_d = torch.from_numpy(seq["color"]).cuda()
    depth_d = torch.from_numpy(seq["depth"]).cuda()
    vol2.integrate_batch(color_d, depth_d, seq["K"], seq["poses"])
    _volumes_equal(vol2, ora)
    torch.cuda.synchronize()
    assert np.array_equal(storage[0].cpu().numpy().reshape(ora._tsdf.shape), ora._tsdf)


def test_empty_and_invalid_inputs(gpu_ctx, oracle_lib, small_sequence):
    from hive_amd import fusion, synthetic
    from hive_amd._lib import HiveError
    seq = small_sequence
    vol = fusion.TSDFVolume(synthetic.room_bounds(), 0.16, ctx=gpu_ctx)
    # all-invalid depth: nothing is written, volume stays at its initial state
    n = vol.integrate(seq["color"][0], np.zeros_like(seq["depth"][0]), seq["K"], seq["poses"][0], return_n_updated=True)
    assert n == 0
    tsdf, color, weight = vol.get_volume(with_weight=True)
    assert (tsdf == 1).all() and (color == 0).all() and (weight == 0).all()
    with pytest.raises(ValueError):
        vol.get_mesh()
    # camera looking away from the volume
    pose = seq["poses"][0].copy()
    pose[:3, 3] = [100.0, 100.0, 100.0]
    assert vol.integrate(seq["color"][0], seq["depth"][0], seq["K"], pose, return_n_updated=True) == 0
    with pytest.raises(AssertionError):
        vol.integrate(seq["color"][0][:, :-1], seq["depth"][0], seq["K"], seq["poses"][0])
    with pytest.raises(HiveError):
        fusion.TSDFVolume(np.array([[0, 1.0], [0, 1.0], [0, 1.0]]), -1.0, ctx=gpu_ctx)
    with pytest.raises(AssertionError):
        fusion.TSDFVolume(np.zeros((2, 3)), 0.1, ctx=gpu_ctx)


def test_pixel_ties_round_modes(gpu_ctx, oracle_lib):
    """Axis-aligned camera with cx = 319.5-style half-pixel principal point: voxel centres project
    exactly onto .5 pixel ties; the two rounding modes must differ from each other and each must
    match the oracle."""
    from hive_amd import fusion
    H, W = 32, 32
    K = np.array([[16.0, 0, 15.5], [0, 16.0, 15.5], [0, 0, 1]], np.float32)
    pose = np.eye(4)
    pose[:3, 3] = [0.5, 0.5, -1.0]
    rng = np.random.default_rng(7)
    depth = np.full((H, W), 2.0, np.float32) + rng.integers(0, 3, (H, W)).astype(np.float32) * 0.25
    color = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    bnds = np.array([[0.0, 1.0], [0.0, 1.0], [0.0, 2.0]])
    # the mode belongs to the volume: both volumes live on ONE context and are integrated alternately.
    # use_gpu picks the reference library's path: True = its CUDA kernel (roundf), False = its numpy path (np.round)
    vols = [fusion.TSDFVolume(bnds, 0.0625, ctx=gpu_ctx, use_gpu=False), fusion.TSDFVolume(bnds, 0.0625, ctx=gpu_ctx, use_gpu=True)]
    oras = [oracle_lib.TSDFVolume(bnds, 0.0625, round_mode=rm) for rm in (0, 1)]
    assert [v.round_mode for v in vols] == [0, 1]
    for _ in range(3):
        for vol, ora in zip(vols, oras):
            vol.integrate(color, depth, K, pose)
            ora.integrate(color, depth, K, pose)
    for vol, ora in zip(vols, oras):
        _volumes_equal(vol, ora)
    assert not np.array_equal(vols[0].get_volume()[0], vols[1].get_volume()[0]), "tie case not exercised"


def test_accumulate_finalize_matches_oracle(gpu_ctx, oracle_lib, small_sequence):
    import torch
    from hive_amd import fusion, synthetic
    seq = small_sequence
    vol = fusion.TSDFVolume(synthetic.room_bounds(), 0.08, ctx=gpu_ctx)
    acc = torch.empty(5 * vol.num_voxels, dtype=torch.float32, device="cuda")
    vol.accum_reset(acc)
    ora = oracle_lib.AccumVolume(synthetic.room_bounds(), 0.08)
    for i in range(8):
        vol.accum_integrate(acc, seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i])
        ora.integrate(seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i])
    torch.cuda.synchronize()
    assert np.array_equal(acc.cpu().numpy().reshape(ora.accum.shape), ora.accum)
    vol.accum_finalize(acc)
    _volumes_equal(vol, ora.finalize())


def test_mesh_matches_oracle(gpu_ctx, oracle_lib, small_sequence):
    from hive_amd import fusion, synthetic
    seq = small_sequence
    vol = fusion.TSDFVolume(synthetic.room_bounds(), 0.08, ctx=gpu_ctx)
    ora = oracle_lib.TSDFVolume(synthetic.room_bounds(), 0.08)
    for i in range(8):
        vol.integrate(seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i])
        ora.integrate(seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i])
    v, f, n, c, vv = vol.get_mesh(return_voxel_coords=True)
    ov, of, on, oc, ovv = ora.get_mesh(return_voxel_coords=True)
    assert v.shape == ov.shape and f.shape == of.shape
    assert np.array_equal(f, of), "face indices (integer work) must be bit-exact"
    assert np.array_equal(c, oc), "vertex colours must be bit-exact"
    assert np.array_equal(vv, ovv) and np.array_equal(v, ov), "vertex positions differ"
    np.testing.assert_allclose(n, on, rtol=0, atol=1e-6)
    pc = vol.get_point_cloud()
    assert pc.shape == (v.shape[0], 6)


def test_x_slab_volumes_are_slices_of_the_whole_volume(gpu_ctx, oracle_lib, small_sequence):
    """The bit-exact multi-GPU mode's kernel side (hive_tsdf_create_slab): three uneven x-slabs of one grid, every frame
    integrated into each -- concatenated they are bit for bit the whole volume, and the oracle's."""
    from hive_amd import _lib, fusion, synthetic
    seq = small_sequence
    bounds, voxel = synthetic.room_bounds(), 0.0641  # 80^3
    whole = fusion.TSDFVolume(bounds, voxel, ctx=gpu_ctx)
    ora = oracle_lib.TSDFVolume(bounds, voxel)
    X = int(whole.vol_dim[0])
    cuts = [0, 27, 28, X]  # slabs of 27, 1 and 52 rows
    slabs = [fusion.TSDFVolume(bounds, voxel, ctx=gpu_ctx, x_range=(a, b)) for a, b in zip(cuts, cuts[1:])]
    assert [int(s.vol_dim[0]) for s in slabs] == [27, 1, X - 28] and all(np.array_equal(s._vol_origin, whole._vol_origin) for s in slabs)
    for i in range(seq["depth"].shape[0]):
        args = (seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i])
        whole.integrate(*args)
        ora.integrate(*args)
        n_parts = [s.integrate(*args, return_n_updated=True) for s in slabs]
        assert sum(n_parts) == ora.last_n_updated
    _volumes_equal(whole, ora)
    parts = [s.get_volume(with_weight=True) for s in slabs]
    for k, ref in enumerate((ora._tsdf, ora._color, ora._weight)):
        assert np.array_equal(np.concatenate([p[k] for p in parts], axis=0), ref)
    with pytest.raises(_lib.HiveError):
        slabs[0].get_mesh()  # marching cubes needs the gathered volume
    with pytest.raises(_lib.HiveError):
        fusion.TSDFVolume(bounds, voxel, ctx=gpu_ctx, x_range=(10, X + 1))


def test_exact_slab_fusion_single_rank_and_device_volume_copies(gpu_ctx, oracle_lib, fusable_sequence):
    """hive_amd.distributed.ExactSlabFusion with one rank (no process group): frames 'all-gathered', integrated (the fused sweep:
    10 frames 9 degrees apart -> sweeps of 4 + 4 + 2, asserted), slabs 'all-gathered' into a whole volume whose mesh can be
    extracted; set_volume_device / device_tensors round trip."""
    import torch
    from hive_amd import distributed as hdist, synthetic
    seq = fusable_sequence
    fus = hdist.ExactSlabFusion(synthetic.room_bounds(), 0.08, ctx=gpu_ctx)
    ora = oracle_lib.TSDFVolume(synthetic.room_bounds(), 0.08)
    n = seq["depth"].shape[0]
    fus.integrate(torch.from_numpy(seq["color"]).cuda(), torch.from_numpy(seq["depth"]).cuda(), seq["K"], seq["poses"], [n])
    assert fus.slab.last_batch_groups() == [4, 4, 2], "the fused sweep did not run on the slab"
    for i in range(n):
        ora.integrate(seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i])
    full = fus.gather()
    _volumes_equal(full, ora)
    verts, faces, _, _ = full.get_mesh()
    o_verts, o_faces, _, _ = ora.get_mesh()
    assert np.array_equal(faces, o_faces) and np.array_equal(verts, o_verts)


@pytest.mark.parametrize("round_mode", [0, 1])
def test_multi_frame_sweep_is_bit_identical(gpu_ctx, oracle_lib, round_mode):
    """hive_tsdf_integrate_batch on device frames fuses up to four consecutive frames per sweep (volume loaded and stored once,
    frames applied to the registers in order): bit-identical to the C oracle's serial loop (hive/fusion.py:113-124) -- 7 frames
    8 degrees apart = sweeps of 4 + 3 (asserted) into 128^3, both rounding modes; the single-frame kernel on the same frames
    with alternating observation weights beside it."""
    import torch
    from hive_amd import fusion, synthetic
    seq = synthetic.make_sequence(num_frames=7, height=120, width=160, yaw_step_deg=8.0, seed=5)
    ora = oracle_lib.TSDFVolume(synthetic.room_bounds(), 0.04, round_mode=round_mode)
    for i in range(7):
        ora.integrate(seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i], obs_weight=1.0 + 0.5 * (i % 2))
    color_d, depth_d = torch.from_numpy(seq["color"]).cuda(), torch.from_numpy(seq["depth"]).cuda()
    fused = fusion.TSDFVolume(synthetic.room_bounds(), 0.04, ctx=gpu_ctx, round_mode=round_mode)
    one = fusion.TSDFVolume(synthetic.room_bounds(), 0.04, ctx=gpu_ctx, round_mode=round_mode)
    for i in range(7):
        one.integrate(color_d[i], depth_d[i], seq["K"], seq["poses"][i], obs_weight=1.0 + 0.5 * (i % 2))
    ora2 = oracle_lib.TSDFVolume(synthetic.room_bounds(), 0.04, round_mode=round_mode)
    for i in range(7):
        ora2.integrate(seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i])
    fused.integrate_batch(color_d, depth_d, seq["K"], seq["poses"])
    assert fused.last_batch_groups() == [4, 3]
    _volumes_equal(one, ora)
    _volumes_equal(fused, ora2)


@pytest.mark.parametrize("round_mode", [0, 1])
def test_multi_frame_sweep_bench_configuration_vs_oracle(gpu_ctx, oracle_lib, round_mode):
    """The kernel and the configuration bench.py times: 640 x 480 frames of the bench trajectory (2.4 degree steps) into 512^3,
    eight device-resident frames through integrate_batch = two sweeps of FOUR (asserted), against the C oracle's eight serial
    integrates, bit for bit, both rounding modes (the oracle runs its x planes on the host's cores: same bits as one thread)."""
    import torch
    from hive_amd import fusion, synthetic
    seq = synthetic.make_sequence(num_frames=8, yaw_step_deg=2.4)
    oracle_lib.set_threads(0)
    ora = oracle_lib.TSDFVolume(synthetic.room_bounds(), 0.01, round_mode=round_mode)
    for i in range(8):
        ora.integrate(seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i])
    vol = fusion.TSDFVolume(synthetic.room_bounds(), 0.01, ctx=gpu_ctx, round_mode=round_mode)
    assert tuple(vol.vol_dim) == (512, 512, 512)
    vol.integrate_batch(torch.from_numpy(seq["color"]).cuda(), torch.from_numpy(seq["depth"]).cuda(), seq["K"], seq["poses"])
    assert vol.last_batch_groups() == [4, 4]
    _volumes_equal(vol, ora)
    assert float(ora._weight.max()) == 8.0


def _poses_yaw(degrees, centre_offsets=None):
    """Camera-to-world poses on the room's circle at the given yaw angles (degrees); optional extra translation per frame."""
    from hive_amd import synthetic
    poses = []
    for k, deg in enumerate(degrees):
        p = synthetic.circular_trajectory(2, yaw_step_deg=float(deg))[1]
        if centre_offsets is not None:
            p = p.copy()
            p[:3, 3] += np.asarray(centre_offsets[k], np.float64)
        poses.append(p)
    return np.stack(poses)


def _render(poses, height=120, width=160, seed=11, zero_frames=()):
    from hive_amd import synthetic
    rng = np.random.default_rng(seed)
    K = synthetic.scaled_intrinsics(height, width)
    color = np.empty((len(poses), height, width, 3), np.uint8)
    depth = np.empty((len(poses), height, width), np.float32)
    for i, pose in enumerate(poses):
        d, pts = synthetic.raycast_room_depth(pose, K, height, width, 0.32, 4.80)
        d = d.copy()
        d[rng.random(d.shape) < 0.02] = 0.0
        if i in zero_frames:
            d[:] = 0.0
        color[i], depth[i] = synthetic.room_colour(pts, rng), d
    return color, depth, K


@pytest.mark.parametrize("case", ["five", "distance", "angle", "zero_depth", "weight", "odd_z"])
def test_multi_frame_grouping_edge_cases(gpu_ctx, oracle_lib, case):
    """How hive_tsdf_integrate_batch forms its sweeps (tsdf.hip `fusable`), each case bit for bit against the C oracle's serial
    loop and with the expected grouping asserted: five frames (4 + 1), a group broken by the camera moving more than a quarter
    of the volume's longest side, a group broken by the 36 degree limit to its FIRST frame, an all-zero depth map inside a
    group, an observation weight other than 1, and Z % 4 != 0 (rows of any length fuse since round 3: the same 4 + 1 sweeps, the rows' tails masked)."""
    import torch
    from hive_amd import fusion, synthetic
    bounds, voxel, obs_w, zero = synthetic.room_bounds(), 0.04, 1.0, ()
    if case == "five":
        poses, groups = _poses_yaw([0, 3, 6, 9, 12]), [4, 1]
    elif case == "distance":  # frame 2 is 1.5 m away from frame 0 (> 5.12 / 4): the group closes before it
        poses, groups = _poses_yaw([0, 2, 4, 6, 8, 10], [(0, 0, 0), (0, 0, 0), (0, -1.5, 0), (0, -1.5, 0), (0, -1.5, 0), (0, -1.5, 0)]), [2, 4]
    elif case == "angle":  # 0, 20, 40: 40 degrees from the group's first frame -> [2, ...]; 40, 60, 70, 75: all within 36 of 40
        poses, groups = _poses_yaw([0, 20, 40, 60, 70, 75]), [2, 4]
    elif case == "zero_depth":
        poses, groups, zero = _poses_yaw([0, 3, 6, 9, 12, 15]), [4, 2], (1, 5)
    elif case == "weight":
        poses, groups, obs_w = _poses_yaw([0, 3, 6, 9]), [4], 0.37
    else:  # Z = ceil(5.0 / 0.04) = 125 or 126: not a multiple of 4
        poses, groups = _poses_yaw([0, 3, 6, 9, 12]), [4, 1]
        bounds = np.array([[0.0, 5.12], [0.04, 5.12], [0.0, 5.0]])
    color, depth, K = _render(poses, zero_frames=zero)
    ora = oracle_lib.TSDFVolume(bounds, voxel)
    for i in range(len(poses)):
        ora.integrate(color[i], depth[i], K, poses[i], obs_weight=obs_w)
    vol = fusion.TSDFVolume(bounds, voxel, ctx=gpu_ctx)
    if case == "odd_z":
        assert vol.vol_dim[2] % 4 != 0
    vol.integrate_batch(torch.from_numpy(color).cuda(), torch.from_numpy(depth).cuda(), K, poses, obs_weight=obs_w)
    assert vol.last_batch_groups() == groups
    _volumes_equal(vol, ora)
    assert ora._weight.max() > 0


def test_multi_frame_sweep_full_size(gpu_ctx):
    """The same at 640 x 480 into 512^3: six consecutive frames of the bench sequence, fused sweep vs six single sweeps."""
    import torch
    from hive_amd import fusion, synthetic
    seq = synthetic.make_sequence(num_frames=6, yaw_step_deg=2.4)
    color_d, depth_d = torch.from_numpy(seq["color"]).cuda(), torch.from_numpy(seq["depth"]).cuda()
    fused = fusion.TSDFVolume(synthetic.room_bounds(), 0.01, ctx=gpu_ctx)
    one = fusion.TSDFVolume(synthetic.room_bounds(), 0.01, ctx=gpu_ctx)
    fused.integrate_batch(color_d, depth_d, seq["K"], seq["poses"])
    for i in range(6):
        one.integrate(color_d[i], depth_d[i], seq["K"], seq["poses"][i])
    for a, b in zip(fused.device_tensors(), one.device_tensors()):
        assert torch.equal(a, b)
    assert float(fused.device_tensors()[1].max()) == 6.0


def test_multi_frame_sweep_on_x_slabs(gpu_ctx, oracle_lib, fusable_sequence):
    """The fused sweep on x-slab volumes (what the bit-exact multi-GPU mode runs on every rank): ten device frames 9 degrees
    apart through integrate_batch into three uneven slabs -- sweeps of 4 + 4 + 2 on EVERY slab (asserted) -- == the oracle's whole
    volume, bit for bit."""
    import torch
    from hive_amd import fusion, synthetic
    seq = fusable_sequence
    bounds, voxel = synthetic.room_bounds(), 0.0641  # 80^3
    ora = oracle_lib.TSDFVolume(bounds, voxel)
    n = seq["depth"].shape[0]
    for i in range(n):
        ora.integrate(seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i])
    color_d, depth_d = torch.from_numpy(seq["color"]).cuda(), torch.from_numpy(seq["depth"]).cuda()
    X = int(ora._vol_dim[0])
    cuts = [0, 27, 28, X]
    parts = []
    for a, b in zip(cuts, cuts[1:]):
        slab = fusion.TSDFVolume(bounds, voxel, ctx=gpu_ctx, x_range=(a, b))
        slab.integrate_batch(color_d, depth_d, seq["K"], seq["poses"])
        assert slab.last_batch_groups() == [4, 4, 2], "the fused sweep did not run on the slab"
        parts.append(slab.get_volume(with_weight=True))
    for k, ref in enumerate((ora._tsdf, ora._color, ora._weight)):
        assert np.array_equal(np.concatenate([p[k] for p in parts], axis=0), ref)
    assert ora._weight.max() >= 8


def test_fast_colour_update_falls_back_when_weights_are_not_whole_numbers(gpu_ctx, oracle_lib, fusable_sequence):
    """The sweep's division-free colour update (update_voxels FASTC) needs every weight of the volume to be a whole number of unit
    observations; the library tracks that (hive_tsdf::unit_weights).  Each way of breaking it -- an observation weight other than 1, planes
    written through hive_tsdf_set_volume, caller-owned planes written directly and announced with hive_tsdf_planes_modified -- must
    send the NEXT unit-weight sweeps down the exact-division path: bit for bit against the oracle from the same starting planes; and a
    reset restores the fast path (same bits either way, which is the point)."""
    import torch
    from hive_amd import fusion, synthetic
    seq = fusable_sequence
    color_d, depth_d = torch.from_numpy(seq["color"]).cuda(), torch.from_numpy(seq["depth"]).cuda()
    bounds, voxel = synthetic.room_bounds(), 0.04

    def run_oracle(ora, frames, w=1.0):
        for i in frames:
            ora.integrate(seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i], obs_weight=w)

    # (a) a sweep with obs_weight 0.5 in between
    vol, ora = fusion.TSDFVolume(bounds, voxel, ctx=gpu_ctx), oracle_lib.TSDFVolume(bounds, voxel)
    vol.integrate_batch(color_d[:4], depth_d[:4], seq["K"], seq["poses"][:4])
    vol.integrate_batch(color_d[4:6], depth_d[4:6], seq["K"], seq["poses"][4:6], obs_weight=0.5)
    vol.integrate_batch(color_d[6:10], depth_d[6:10], seq["K"], seq["poses"][6:10])
    run_oracle(ora, range(4)), run_oracle(ora, range(4, 6), 0.5), run_oracle(ora, range(6, 10))
    _volumes_equal(vol, ora)
    # (b) planes set by the caller: weights scaled by 0.75 (no longer whole numbers), through set_volume
    t, c, w = vol.get_volume(with_weight=True)
    vol.set_volume(tsdf=t, color=c, weight=w * np.float32(0.75))
    ora._weight = (ora._weight * np.float32(0.75)).astype(np.float32)
    vol.integrate_batch(color_d[:4], depth_d[:4], seq["K"], seq["poses"][:4])
    run_oracle(ora, range(4))
    _volumes_equal(vol, ora)
    # (c) caller-owned planes written directly, announced with planes_modified
    n = int(np.prod(vol.vol_dim))
    storage = tuple(torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(3))
    ext, ora2 = fusion.TSDFVolume(bounds, voxel, ctx=gpu_ctx, storage=storage), oracle_lib.TSDFVolume(bounds, voxel)
    ext.integrate_batch(color_d[:4], depth_d[:4], seq["K"], seq["poses"][:4])
    run_oracle(ora2, range(4))
    storage[1].mul_(0.5)
    ext.planes_modified()
    ora2._weight = (ora2._weight * np.float32(0.5)).astype(np.float32)
    ext.integrate_batch(color_d[4:8], depth_d[4:8], seq["K"], seq["poses"][4:8])
    run_oracle(ora2, range(4, 8))
    _volumes_equal(ext, ora2)
    # (d) reset: whole numbers again
    ext.reset()
    ora3 = oracle_lib.TSDFVolume(bounds, voxel)
    ext.integrate_batch(color_d, depth_d, seq["K"], seq["poses"])
    run_oracle(ora3, range(10))
    _volumes_equal(ext, ora3)


def test_clip_of_rows_parallel_to_the_image_plane(gpu_ctx, oracle_lib):
    """Regression (round 4): at a yaw of exactly 90 degrees the voxel rows run parallel to the image plane (cam_z = az + 6e-17 tz), and with the
    frame's deepest pixel on a wall exactly at max depth the far cut's alpha is 0: -alpha / beta cut those rows at tz <= 0 although all
    of their voxels sit ON the truncation boundary and update.  Foreground-masked depth (sparse: small max depth) of frame 3 of the 30-degree
    sequence into 512^3, single-frame kernel and a fused pair, N_upd and weights against the oracle."""
    import torch
    from hive_amd import fusion, synthetic
    seq = synthetic.make_sequence(num_frames=6, yaw_step_deg=30.0)
    masks = synthetic.ellipse_masks(6, 480, 640, num_objects=3, seed=9)
    depth = np.where(masks > 0, seq["depth"], 0).astype(np.float32)
    assert abs(seq["poses"][3][2, 2]) < 1e-15, "frame 3 looks along +x: rows along z are parallel to the image plane"
    vol, ora = fusion.TSDFVolume(synthetic.room_bounds(), 0.01, ctx=gpu_ctx), oracle_lib.TSDFVolume(synthetic.room_bounds(), 0.01)
    n = vol.integrate(seq["color"][3], depth[3], seq["K"], seq["poses"][3], return_n_updated=True)
    ora.integrate(seq["color"][3], depth[3], seq["K"], seq["poses"][3])
    assert n == ora.last_n_updated
    pair = fusion.TSDFVolume(synthetic.room_bounds(), 0.01, ctx=gpu_ctx)
    pair.integrate_batch(torch.from_numpy(seq["color"][2:4]).cuda(), torch.from_numpy(depth[2:4]).cuda(), seq["K"], seq["poses"][2:4])
    assert pair.last_batch_groups() == [2]
    ora_pair = oracle_lib.TSDFVolume(synthetic.room_bounds(), 0.01)
    for i in (2, 3):
        ora_pair.integrate(seq["color"][i], depth[i], seq["K"], seq["poses"][i])
    assert np.array_equal(pair.get_volume(with_weight=True)[2], ora_pair._weight)


@pytest.mark.parametrize("mode", ["0", "1", "2"])
def test_per_row_far_cut_modes(gpu_ctx, oracle_lib, monkeypatch, mode):
    """The work list's per-row far cut (largest depth in the tiles a row's image segment crosses) is chosen per frame from the tile table
    (HIVE_TSDF_ROW_FAR=1, the default); 0 / 2 force it off / on.  All three give the oracle's volume on a scene where it cuts (foreground-masked
    depth: most tiles empty) and on one where it does not (the room), fused sweeps and the single-frame kernel."""
    import torch
    from hive_amd import fusion, synthetic
    monkeypatch.setenv("HIVE_TSDF_ROW_FAR", mode)
    seq = synthetic.make_sequence(num_frames=6, height=240, width=320, yaw_step_deg=6.0, seed=11)
    masks = synthetic.ellipse_masks(6, 240, 320, num_objects=2, seed=3)
    items = {}
    for name, depth in (("room", seq["depth"]), ("masked", np.where(masks > 0, seq["depth"], 0).astype(np.float32))):
        ora = oracle_lib.TSDFVolume(synthetic.room_bounds(), 0.025)
        for i in range(6):
            ora.integrate(seq["color"][i], depth[i], seq["K"], seq["poses"][i])
        fused = fusion.TSDFVolume(synthetic.room_bounds(), 0.025, ctx=gpu_ctx)
        fused.integrate_batch(torch.from_numpy(seq["color"]).cuda(), torch.from_numpy(depth).cuda(), seq["K"], seq["poses"])
        assert fused.last_batch_groups() == [4, 2]
        items[name] = fused.last_sweep_voxels()
        _volumes_equal(fused, ora)
        one = fusion.TSDFVolume(synthetic.room_bounds(), 0.025, ctx=gpu_ctx)
        for i in range(6):
            one.integrate(seq["color"][i], depth[i], seq["K"], seq["poses"][i])
        _volumes_equal(one, ora)
    assert items["masked"] < items["room"]
