"""The driver functions (hive/fusion.py:37-134) and the on-device depth+fusion stream, on the GPU, against
the same control flow executed with the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class FakeDataset:
    """The attributes of HiveDataset that tsdf_fusion / adjust_voxel_size read (hive/fusion.py:51-121)."""

    def __init__(self, seq, masks=None, inpainted=False):
        from hive_amd import synthetic
        from hive_amd.geometric import Trajectory
        self.num_frames = seq["depth"].shape[0]
        self.camera_matrix = seq["K"]
        self.camera_trajectory = Trajectory(synthetic.trajectory_rows_world_to_cam(seq["poses"]))  # world-to-cam rows, float32
        self._seq = seq
        self.mask_dataset = masks if masks is not None else [np.zeros(seq["depth"].shape[1:], np.uint8)] * self.num_frames
        self.has_inpainted_frame_data = inpainted

    @property
    def bg_rgb_dataset(self):
        return [c.copy() for c in self._seq["color"]]

    @property
    def bg_depth_dataset(self):
        return [d.copy() for d in self._seq["depth"]]  # tsdf_fusion mutates depth in place (fusion.py:121)

    rgb_dataset = bg_rgb_dataset      # the captured frames (no inpainted folders in this fake)
    depth_dataset = bg_depth_dataset


def _oracle_fusion(oracle_lib, dataset, options, frame_set):
    """hive/fusion.py:37-134 restated with the oracle (CPU)."""
    vol_bnds = np.zeros((3, 2))
    c2w = dataset.camera_trajectory.inverse().to_homogenous_transforms()
    depths = dataset.bg_depth_dataset
    for i in frame_set:
        f = oracle_lib.view_frustum(depths[i], dataset.camera_matrix, c2w[i])
        vol_bnds[:, 0] = np.minimum(vol_bnds[:, 0], f.min(axis=1))
        vol_bnds[:, 1] = np.maximum(vol_bnds[:, 1], f.max(axis=1))
    voxel_count = np.ceil(np.prod((vol_bnds[:, 1] - vol_bnds[:, 0]) / options.sdf_voxel_size))
    voxel_size = options.sdf_voxel_size
    if options.sdf_max_voxels and voxel_count > options.sdf_max_voxels:
        voxel_size = (np.prod(vol_bnds[:, 1] - vol_bnds[:, 0]) / options.sdf_max_voxels) ** (1 / 3)
    vol = oracle_lib.TSDFVolume(vol_bnds, voxel_size)
    colors, depths = dataset.bg_rgb_dataset, dataset.bg_depth_dataset
    for i in frame_set:
        depth = depths[i]
        if not dataset.has_inpainted_frame_data:
            mask = oracle_lib.dilate_mask(dataset.mask_dataset[i], options.depth_mask_dilation_iterations)
            depth[mask > 0] = 0.0
        vol.integrate(colors[i], depth, dataset.camera_matrix, c2w[i])
    return voxel_size, vol_bnds, vol


def test_tsdf_fusion_driver_matches_oracle(gpu_ctx, oracle_lib):
    from hive_amd import fusion, synthetic
    from hive_amd.options import BackgroundMeshOptions
    seq = synthetic.make_sequence(num_frames=6, height=60, width=80, yaw_step_deg=60.0)
    rng = np.random.default_rng(0)
    masks = []
    for _ in range(6):
        m = np.zeros((60, 80), np.uint8)
        v, u = rng.integers(10, 50), rng.integers(10, 70)
        m[v:v + 4, u:u + 5] = 2  # an instance id
        masks.append(m)
    options = BackgroundMeshOptions(sdf_voxel_size=0.02, sdf_max_voxels=300_000, depth_mask_dilation_iterations=3)
    ds = FakeDataset(seq, masks)
    frame_set = [0, 2, 3, 5]
    voxel, bnds = fusion.adjust_voxel_size(ds, options, frame_set)
    o_voxel, o_bnds, o_vol = _oracle_fusion(oracle_lib, FakeDataset(seq, masks), options, frame_set)
    assert voxel == o_voxel and np.array_equal(bnds, o_bnds)
    assert voxel > options.sdf_voxel_size, "the voxel budget must kick in for this scene"
    assert (bnds[:, 0] <= 0).all() and (bnds[:, 1] >= 0).all(), "bounds always contain the world origin (fusion.py:48)"
    mesh, vol = fusion.tsdf_fusion(ds, options, frame_set=frame_set, return_volume=True)
    tsdf, color, weight = vol.get_volume(with_weight=True)
    assert np.array_equal(tsdf, o_vol._tsdf) and np.array_equal(color, o_vol._color) and np.array_equal(weight, o_vol._weight)
    o_verts, o_faces, o_norms, o_colors = o_vol.get_mesh()
    assert len(o_verts) > 0
    verts, faces, norms, colors = vol.get_mesh()
    assert np.array_equal(faces, o_faces) and np.array_equal(verts, o_verts) and np.array_equal(colors, o_colors)
    from hive_amd.mesh import Mesh
    if isinstance(mesh, Mesh):  # no trimesh installed: the driver's mesh is the extraction itself (trimesh would merge vertices)
        assert np.array_equal(np.asarray(mesh.faces), o_faces) and np.array_equal(np.asarray(mesh.vertices), o_verts)
    else:
        assert len(mesh.faces) <= len(o_faces) and len(mesh.vertices) <= len(o_verts)
    # defaults: num_frames=-1 -> all frames; inpainted data -> masks are not applied
    ds2 = FakeDataset(seq, masks, inpainted=True)
    mesh2 = fusion.tsdf_fusion(ds2, BackgroundMeshOptions(sdf_voxel_size=0.08, sdf_max_voxels=None))
    assert len(mesh2.vertices) > 0 and np.asarray(mesh2.visual.vertex_colors).shape[1] == 4


def test_depth_fusion_stream_equals_manual_steps(gpu_ctx, oracle_lib):
    """preprocess -> DPT -> hand-off -> integrate as one stream == the same steps done by hand, and the
    TSDF it produces == the oracle fed with the stream's own depth maps."""
    import torch
    from hive_amd import depth as depth_mod, fusion, synthetic
    seq = synthetic.make_sequence(num_frames=4, height=96, width=128, yaw_step_deg=30.0)
    torch.manual_seed(0)
    model = depth_mod.build_model(None, dtype=torch.bfloat16)
    # spread the (random-weight) depth over a useful range: random bias on the last conv input features
    with torch.no_grad():
        model.scratch.output_conv[4].bias.fill_(1500.0)
        model.scratch.output_conv[4].weight.mul_(3000.0)
    vol = fusion.TSDFVolume(synthetic.room_bounds(), 0.08, ctx=gpu_ctx)
    stream = depth_mod.DepthFusionStream(model, vol, seq["K"])
    frames = torch.from_numpy(seq["color"]).cuda()
    depth_m = stream.step(frames, seq["poses"])
    torch.cuda.synchronize()
    d = depth_m.cpu().numpy()
    assert d.shape == (4, 96, 128) and d.dtype == np.float32
    assert np.array_equal(d, np.float32(0.001) * np.round(d * 1000).astype(np.float32)) or np.allclose(d * 1000, np.round(d * 1000), atol=1e-3)
    assert (d <= 10.0).all() and d.max() > 0
    # preprocessing == the reference: float64 ((x / 255) - .5) / .5 -> float32 -> the 16-bit network type
    x = depth_mod.preprocess_on_device(frames, torch.bfloat16)
    ref64 = (seq["color"] / 255.0 - 0.5) / 0.5
    ref = torch.from_numpy(ref64.astype(np.float32)).permute(0, 3, 1, 2).bfloat16()
    assert torch.equal(x.cpu(), ref)
    ora = oracle_lib.TSDFVolume(synthetic.room_bounds(), 0.08)
    for i in range(4):
        ora.integrate(seq["color"][i], d[i], seq["K"], seq["poses"][i])
    tsdf, color, weight = vol.get_volume(with_weight=True)
    assert np.array_equal(tsdf, ora._tsdf) and np.array_equal(color, ora._color) and np.array_equal(weight, ora._weight)


def test_overlapped_stream_equals_single_stream(gpu_ctx):
    """DepthFusionStream(overlap=True): the TSDF sweeps of batch i run on a second (lowest-priority) HIP stream under the network of
    batch i + 1.  Same frames, same order -> the volume is bit-identical to the single-stream run; frame and depth buffers are freed by
    the caller right after `step` (the caching allocator must not recycle them under the sweeps), six batches back to back."""
    import torch
    from hive_amd import depth as depth_mod, fusion, synthetic
    seq = synthetic.make_sequence(num_frames=24, height=96, width=128, yaw_step_deg=3.0)
    model = depth_mod.build_model(None, dtype=torch.float16, init_seed=5)
    vols = []
    for overlap in (False, True):
        ctx = depth_mod.DepthFusionStream.side_stream_context(0) if overlap else gpu_ctx
        vol = fusion.TSDFVolume(synthetic.room_bounds(), 0.04, ctx=ctx)
        stream = depth_mod.DepthFusionStream(model, vol, seq["K"], overlap=overlap)
        for b in range(6):
            frames = torch.from_numpy(seq["color"][4 * b:4 * b + 4]).cuda()
            depth = stream.step(frames, seq["poses"][4 * b:4 * b + 4])
            assert (stream.last_done is not None) == overlap
            del frames, depth  # freed while (overlap) the sweeps may still be reading them
            torch.empty(1 << 20, device="cuda").fill_(1.0)  # main-stream allocations in between
        stream.join()
        assert vol.last_batch_groups() == [4]
        vols.append(vol.device_tensors())
    torch.cuda.synchronize()
    assert float(vols[0][1].max()) >= 8.0, "consecutive frames must overlap in the volume"
    for a, b in zip(*vols):
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        depth_mod.DepthFusionStream(model, fusion.TSDFVolume(synthetic.room_bounds(), 0.16, ctx=gpu_ctx), seq["K"], overlap=True)


def test_accumulate_stream_then_fuse_single_rank(gpu_ctx, oracle_lib):
    import torch
    from hive_amd import depth as depth_mod, distributed as hdist, fusion, synthetic
    seq = synthetic.make_sequence(num_frames=3, height=96, width=128, yaw_step_deg=30.0)
    torch.manual_seed(0)
    model = depth_mod.build_model(None, dtype=torch.bfloat16)
    vol = fusion.TSDFVolume(synthetic.room_bounds(), 0.16, ctx=gpu_ctx)
    stream = depth_mod.DepthFusionStream(model, vol, seq["K"], accumulate=True)
    frames = torch.from_numpy(seq["color"]).cuda()
    d = stream.step(frames, seq["poses"]).cpu().numpy()
    hdist.fuse_sharded(vol, stream)
    ora = oracle_lib.AccumVolume(synthetic.room_bounds(), 0.16)
    for i in range(3):
        ora.integrate(seq["color"][i], d[i], seq["K"], seq["poses"][i])
    ref = ora.finalize()
    tsdf, color, weight = vol.get_volume(with_weight=True)
    assert np.array_equal(tsdf, ref._tsdf) and np.array_equal(color, ref._color) and np.array_equal(weight, ref._weight)


def test_view_frusta_batch_equals_single_calls(gpu_ctx, oracle_lib, small_sequence):
    """The bounds pass for a whole frame set in one launch == n calls of get_view_frustum == the oracle, bit for bit,
    from host arrays and from depth maps already in HBM."""
    import torch
    from hive_amd import fusion
    seq = small_sequence
    batch = fusion.view_frusta(seq["depth"], seq["K"], seq["poses"], ctx=gpu_ctx)
    batch_dev = fusion.view_frusta(torch.from_numpy(seq["depth"]).cuda(), seq["K"], seq["poses"], ctx=gpu_ctx)
    assert batch.shape == (8, 3, 5) and np.array_equal(batch, batch_dev)
    for i in range(8):
        single = fusion.get_view_frustum(seq["depth"][i], seq["K"], seq["poses"][i], ctx=gpu_ctx)
        assert np.array_equal(batch[i], single)
        assert np.array_equal(batch[i], oracle_lib.view_frustum(seq["depth"][i], seq["K"], seq["poses"][i]))
    frames = fusion.DeviceFrames(torch.from_numpy(seq["color"]).cuda(), torch.from_numpy(seq["depth"]).cuda(), seq["poses"])
    bnds = fusion.scene_bounds(frames, seq["K"], ctx=gpu_ctx)
    assert np.array_equal(bnds[:, 0], np.minimum(0, batch.min(axis=(0, 2)))) and np.array_equal(bnds[:, 1], np.maximum(0, batch.max(axis=(0, 2))))


@pytest.mark.parametrize("iterations", [0, 3, 10])
def test_depth_apply_mask_matches_oracle(gpu_ctx, oracle_lib, iterations):
    """On-device `depth[dilate(mask) > 0] = 0` (hive/fusion.py:118-121) and its complement, for a frame set, vs the oracle's
    literal iterated 3 x 3 dilation per frame; masks touching the image border, instance ids, frame borders."""
    import torch
    from hive_amd import fusion, synthetic
    rng = np.random.default_rng(iterations)
    n, H, W = 5, 60, 80
    depth = rng.uniform(0.5, 4.0, (n, H, W)).astype(np.float32)
    masks = synthetic.ellipse_masks(n, H, W, num_objects=3, seed=3)
    masks[0, :2, :] = 2      # touches the top border of frame 0: must not bleed into frame 1 / wrap around
    masks[2, -1, -5:] = 1    # bottom-right corner of frame 2
    frames = fusion.DeviceFrames(torch.zeros((n, H, W, 3), dtype=torch.uint8, device="cuda"), torch.from_numpy(depth).cuda(),
                                 np.tile(np.eye(4), (n, 1, 1)), torch.from_numpy(masks).cuda())
    bg = frames.masked_depth(iterations, fusion.MASK_BACKGROUND, ctx=gpu_ctx).cpu().numpy()
    fg = frames.masked_depth(0, fusion.MASK_FOREGROUND, ctx=gpu_ctx).cpu().numpy()
    fg2 = frames.masked_depth(0, fusion.MASK_FOREGROUND, instance_id=2, ctx=gpu_ctx).cpu().numpy()
    for i in range(n):
        expect = depth[i].copy()
        expect[oracle_lib.dilate_mask(masks[i], iterations) > 0] = 0.0
        assert np.array_equal(bg[i], expect), f"frame {i}"
        assert np.array_equal(fg[i], np.where(masks[i] > 0, depth[i], 0).astype(np.float32))
        assert np.array_equal(fg2[i], np.where(masks[i] == 2, depth[i], 0).astype(np.float32))
    assert (bg == 0).any() and (fg2 > 0).any()
    # the same with structuring elements other than the 3 x 3 box (MaskDilationOptions.filter, hive/options.py:245-268), per instance too
    cross = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], np.uint8)
    for se in (cross, np.ones((5, 5), np.uint8), np.ones((2, 3), np.uint8), np.array([[1, 0, 0], [0, 0, 0], [0, 0, 1]], np.uint8)):
        bg_se = frames.masked_depth(iterations, fusion.MASK_BACKGROUND, ctx=gpu_ctx, dilation_filter=se).cpu().numpy()
        bg_se2 = frames.masked_depth(iterations, fusion.MASK_BACKGROUND, instance_id=2, ctx=gpu_ctx, dilation_filter=se).cpu().numpy()
        for i in range(n):
            expect = depth[i].copy()
            expect[oracle_lib.dilate_mask_se(masks[i], se, iterations)] = 0.0
            assert np.array_equal(bg_se[i], expect), f"frame {i}, element {se.shape}"
            expect = depth[i].copy()
            expect[oracle_lib.dilate_mask_se(masks[i] == 2, se, iterations)] = 0.0
            assert np.array_equal(bg_se2[i], expect), f"frame {i}, element {se.shape}, instance 2"


def test_fg_bg_volumes_match_oracle(gpu_ctx, oracle_lib):
    """BASELINE config 5 in miniature: instance masks (moving ellipses), background volume = masked depth, foreground volume =
    the complement, both from one resident frame set; each volume bit-exact against the oracle fed with numpy-masked depth."""
    from hive_amd import fusion, synthetic
    from hive_amd.options import BackgroundMeshOptions
    n = 6
    seq = synthetic.make_sequence(num_frames=n, height=60, width=80, yaw_step_deg=25.0)
    masks = synthetic.ellipse_masks(n, 60, 80, num_objects=2, seed=5)
    options = BackgroundMeshOptions(sdf_voxel_size=0.05, sdf_max_voxels=400_000, depth_mask_dilation_iterations=2)
    ds = FakeDataset(seq, list(masks))
    vols = fusion.tsdf_fusion_fg_bg(ds, options)
    c2w = ds.camera_trajectory.inverse().to_homogenous_transforms()
    o_voxel, o_bnds, o_bg = _oracle_fusion(oracle_lib, FakeDataset(seq, list(masks)), options, range(n))
    o_fg = oracle_lib.TSDFVolume(o_bnds, o_voxel)
    for i in range(n):
        o_fg.integrate(seq["color"][i], np.where(masks[i] > 0, seq["depth"][i], 0).astype(np.float32), seq["K"], c2w[i])
    for name, ora in (("bg", o_bg), ("fg", o_fg)):
        tsdf, color, weight = vols[name].get_volume(with_weight=True)
        assert np.array_equal(weight, ora._weight) and np.array_equal(tsdf, ora._tsdf) and np.array_equal(color, ora._color), name
    assert o_fg._weight.max() > 0 and (o_bg._weight > 0).sum() > (o_fg._weight > 0).sum()
    # the static-scene driver computes the same background volume
    mesh, vol = fusion.tsdf_fusion(ds, options, return_volume=True)
    assert np.array_equal(vol.get_volume()[0], o_bg._tsdf)


def test_chunked_frame_streaming_equals_resident_set(gpu_ctx, oracle_lib):
    """A frame set longer than the resident chunk (``chunk_frames``) streams through one reused staging set in two passes --
    bounds (depth only), then masking + integration chunk by chunk -- and gives the same bounds, volumes and mesh, bit for bit, as
    the resident path and as the oracle's frame-at-a-time loop (hive/fusion.py:113-124); 11 frames in chunks of 4 (4 + 4 + 3,
    the chunk boundary cuts a fused sweep)."""
    from hive_amd import fusion, synthetic
    from hive_amd.options import BackgroundMeshOptions
    n = 11
    seq = synthetic.make_sequence(num_frames=n, height=60, width=80, yaw_step_deg=7.0)
    masks = list(synthetic.ellipse_masks(n, 60, 80, num_objects=2, seed=3))
    options = BackgroundMeshOptions(sdf_voxel_size=0.05, sdf_max_voxels=400_000, depth_mask_dilation_iterations=2)
    _, whole = fusion.tsdf_fusion(FakeDataset(seq, masks), options, return_volume=True)
    assert whole.last_batch_groups() == [4, 4, 3]
    _, chunked = fusion.tsdf_fusion(FakeDataset(seq, masks), options, return_volume=True, chunk_frames=4)
    assert chunked.last_batch_groups() == [3], "the last chunk holds frames 8..10"
    o_voxel, o_bnds, o_vol = _oracle_fusion(oracle_lib, FakeDataset(seq, masks), options, range(n))
    assert np.array_equal(chunked._vol_bnds[:, 0], whole._vol_bnds[:, 0]) and np.array_equal(chunked._vol_dim, whole._vol_dim)
    for vol in (whole, chunked):
        tsdf, color, weight = vol.get_volume(with_weight=True)
        assert np.array_equal(tsdf, o_vol._tsdf) and np.array_equal(color, o_vol._color) and np.array_equal(weight, o_vol._weight)
    vols = fusion.tsdf_fusion_fg_bg(FakeDataset(seq, masks), options, chunk_frames=5)
    assert np.array_equal(vols["bg"].get_volume()[0], o_vol._tsdf) and float(vols["fg"].get_volume(with_weight=True)[2].max()) > 0
    # more frames than one batched launch of the bounds / masking kernels takes: the wrappers split the set
    old = fusion.MAX_BATCH_FRAMES
    fusion.MAX_BATCH_FRAMES = 4
    try:
        frames = fusion.DeviceFrames.from_dataset(FakeDataset(seq, masks), list(range(n)), with_masks=True)
        assert np.array_equal(fusion.scene_bounds(frames, seq["K"]), o_bnds)
        import torch
        a = frames.masked_depth(2, fusion.MASK_BACKGROUND)
    finally:
        fusion.MAX_BATCH_FRAMES = old
    assert torch.equal(a, frames.masked_depth(2, fusion.MASK_BACKGROUND))


def test_fg_bg_partition_property_full_size(gpu_ctx):
    """Size-independent property at the benchmark's size (640 x 480 frames, 512^3): with no dilation the background and
    foreground depth maps partition every frame's valid pixels, and a voxel's update depends on its own pixel only, so the
    observation counts add up exactly: weight_bg + weight_fg == weight of the unmasked fusion, voxel for voxel."""
    import torch
    from hive_amd import fusion, synthetic
    n = 6
    seq = synthetic.make_sequence(num_frames=n, yaw_step_deg=30.0)
    masks = synthetic.ellipse_masks(n, 480, 640, num_objects=3, seed=9)
    frames = fusion.DeviceFrames(torch.from_numpy(seq["color"]).cuda(), torch.from_numpy(seq["depth"]).cuda(), seq["poses"],
                                 torch.from_numpy(masks).cuda())
    weights = {}
    for name, depth in (("full", frames.depth), ("bg", frames.masked_depth(0, fusion.MASK_BACKGROUND, ctx=gpu_ctx)),
                        ("fg", frames.masked_depth(0, fusion.MASK_FOREGROUND, ctx=gpu_ctx))):
        vol = fusion.TSDFVolume(synthetic.room_bounds(), 0.01, ctx=gpu_ctx)
        assert tuple(vol.vol_dim) == (512, 512, 512)
        vol.integrate_batch(frames.color, depth, seq["K"], frames.poses)
        weights[name] = torch.from_numpy(vol.get_volume(with_weight=True)[2])
        vol.close()
    assert float(weights["fg"].max()) > 0
    assert torch.equal(weights["bg"] + weights["fg"], weights["full"])
