"""select_key_frames (hive/io.py:1117-1189) on the GPU against its numpy restatement on the oracle, and the
dataset -> key frames -> TSDF -> mesh pipeline on a synthetic TUM-layout sequence (BASELINE.json config 1 shape:
ground-truth pose + sensor depth)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tum_fixture import write_tum_sequence  # noqa: E402


def _select_key_frames_oracle(oracle_lib, dataset, threshold, frame_step):
    """io.py:1117-1189 restated with the CPU oracle's unproject / project."""
    from hive_amd.geometric import get_pose_components, pose_vec2mat
    width, height, K = dataset.metadata.width, dataset.metadata.height, dataset.camera_matrix
    Kinv = np.linalg.inv(K).astype(np.float64)
    key_frames = [0]
    for frame in range(1, dataset.num_frames, frame_step):
        depth = dataset.bg_depth_dataset[frame]
        mask = np.asarray(dataset.mask_dataset[frame]) == 0
        R, t = get_pose_components(pose_vec2mat(dataset.camera_trajectory[frame]))
        pts, _ = oracle_lib.unproject(depth, mask, Kinv, R, t)
        for kf in key_frames:
            Rk, tk = get_pose_components(pose_vec2mat(dataset.camera_trajectory[kf]))
            uv, _ = oracle_lib.project(pts, K, Rk, tk)
            vis = uv[(uv[:, 0] >= 0) & (uv[:, 0] < width) & (uv[:, 1] >= 0) & (uv[:, 1] < height)]
            if len(vis) == 0:
                continue
            area = np.prod(vis.max(axis=0) - vis.min(axis=0))
            if area / (width * height) >= threshold:
                break
        else:
            key_frames.append(frame)
    return key_frames


def test_project_bbox_matches_oracle(gpu_ctx, oracle_lib):
    from hive_amd._lib import MEM_HOST, ptr
    rng = np.random.default_rng(0)
    pts = rng.uniform(-2, 2, (5000, 3)) + np.array([0, 0, 3.0])
    K = np.array([[580.0, 0, 319.5], [0, 580.0, 239.5], [0, 0, 1]])
    R, t = np.eye(3), np.array([0.1, -0.2, 0.3])
    box = np.zeros(5, np.int32)
    gpu_ctx.check(gpu_ctx.lib.hive_project_bbox(gpu_ctx.handle, ptr(pts), len(pts), ptr(K), ptr(R), ptr(t), 640, 480, MEM_HOST, ptr(box)))
    uv, _ = oracle_lib.project(pts, K, R, t)
    vis = uv[(uv[:, 0] >= 0) & (uv[:, 0] < 640) & (uv[:, 1] >= 0) & (uv[:, 1] < 480)]
    assert list(box) == [vis[:, 0].min(), vis[:, 0].max(), vis[:, 1].min(), vis[:, 1].max(), len(vis)]
    # nothing visible / empty input
    far = pts + np.array([1000.0, 0, 0])
    gpu_ctx.check(gpu_ctx.lib.hive_project_bbox(gpu_ctx.handle, ptr(far), len(far), ptr(K), ptr(R), ptr(t), 640, 480, MEM_HOST, ptr(box)))
    assert box[4] == 0
    gpu_ctx.check(gpu_ctx.lib.hive_project_bbox(gpu_ctx.handle, None, 0, ptr(K), ptr(R), ptr(t), 640, 480, MEM_HOST, ptr(box)))
    assert box[4] == 0


def test_select_key_frames_and_pipeline_on_tum_sequence(gpu_ctx, oracle_lib, tmp_path):
    from hive_amd.dataset_adaptors import get_dataset
    from hive_amd.options import BackgroundMeshOptions
    from hive_amd.pipeline import Pipeline
    tum, out = str(tmp_path / "tum"), str(tmp_path / "hive")
    write_tum_sequence(tum, num_frames=8, yaw_step_deg=25.0)
    ds = get_dataset(tum, out)
    for threshold, step in ((0.3, 1), (0.6, 2), (0.95, 1)):
        got = ds.select_key_frames(threshold=threshold, frame_step=step)
        assert got == _select_key_frames_oracle(oracle_lib, ds, threshold, step), (threshold, step)
        assert got[0] == 0 and got == sorted(set(got))
    assert len(ds.select_key_frames(threshold=0.95, frame_step=1)) > len(ds.select_key_frames(threshold=0.3, frame_step=1))
    # whole background path: dataset folder -> key frames -> TSDF fusion (voxel budget kicks in) -> mesh -> PLY
    options = BackgroundMeshOptions(sdf_voxel_size=0.02, sdf_max_voxels=2_000_000, key_frame_threshold=0.9, key_frame_step=2)
    pipe = Pipeline(background_mesh_options=options)
    mesh = pipe.run(tum, str(tmp_path / "run"))
    assert len(mesh.vertices) > 1000 and len(mesh.faces) > 1000
    assert os.path.getsize(str(tmp_path / "run" / "mesh" / "bg.ply")) > 10000
    assert pipe.profiling["timing"]["background_reconstruction"]["total"] > 0
    # the reconstructed room walls lie near the analytic room of the generator, expressed in the adaptor's frame:
    # all vertices within the 5.12 m cube's diagonal of the first camera
    assert np.linalg.norm(np.asarray(mesh.vertices), axis=1).max() < 9.0


def test_config1_tum_layout_50_frames_256_cubed_numpy_rounding(gpu_ctx, oracle_lib, tmp_path):
    """BASELINE.json configs[0] at its stated shape, on a synthetic TUM-layout folder (the TUM sequence itself is not in the container): 50 frames with
    ground-truth poses and sensor depth (PNG scale 5000, /root/reference/hive/dataset_adaptors.py:574-766) -> `get_dataset` in the reference's call form ->
    a FIXED 256^3 volume (2 cm voxels over a 5.12 m cube around the scene) with `TSDFVolume(..., use_gpu=False)`, i.e. the rounding of the reference
    library's numpy path (np.round, ties to even) -> the fused sweeps.  Volume bits == the C oracle with round_mode 0, all three planes."""
    import torch
    from hive_amd import fusion
    from hive_amd._lib import ROUND_HALF_EVEN
    from hive_amd.dataset_adaptors import get_dataset
    from hive_amd.options import COLMAPOptions, PipelineOptions, StorageOptions
    tum, out = str(tmp_path / "tum"), str(tmp_path / "hive")
    n = 50
    write_tum_sequence(tum, num_frames=n, yaw_step_deg=360.0 / n)
    ds = get_dataset(StorageOptions(dataset_path=tum, output_path=out), COLMAPOptions(), PipelineOptions(num_frames=n, frame_step=1))
    assert ds.num_frames == n
    frames = fusion.DeviceFrames.from_dataset(ds, list(range(n)), with_masks=False)
    K = ds.camera_matrix
    # the scene's bounds (union of the view frusta, as fusion.py:48-61) fix the centre of the cube; its side and the voxel size are the configuration's
    bnds = fusion.scene_bounds(frames, K)
    centre = np.round(bnds.mean(axis=1), 2)
    lo = centre - 2.56
    hi = lo + 5.12
    for a in range(3):  # (hi - lo) / 0.02 must not round above 256 (the library takes the ceiling, as the reference's does)
        while np.ceil((hi[a] - lo[a]) / 0.02) > 256:
            hi[a] = np.nextafter(hi[a], -np.inf)
    vol_bnds = np.stack([lo, hi], axis=1)
    vol = fusion.TSDFVolume(vol_bnds, 0.02, use_gpu=False, ctx=gpu_ctx)
    assert vol.round_mode == ROUND_HALF_EVEN and tuple(int(d) for d in vol.vol_dim) == (256, 256, 256)
    vol.integrate_batch(frames.color, frames.depth, K, frames.poses)
    assert max(vol.last_batch_groups()) > 1, "consecutive frames 7.2 degrees apart share sweeps"
    ora = oracle_lib.TSDFVolume(vol_bnds, 0.02, round_mode=0)
    color, depth = frames.color.cpu().numpy(), frames.depth.cpu().numpy()
    for i in range(n):
        ora.integrate(color[i], depth[i], K, frames.poses[i])
    tsdf, col, weight = vol.get_volume(with_weight=True)
    assert float(weight.max()) >= 10 and (tsdf < 0).any(), "the fusion saw surfaces"
    assert np.array_equal(weight, ora._weight) and np.array_equal(tsdf, ora._tsdf) and np.array_equal(col, ora._color)
    verts, faces, _, _ = vol.get_mesh()
    o_verts, o_faces, _, _ = ora.get_mesh()
    assert np.array_equal(faces, o_faces) and np.array_equal(verts, o_verts)
