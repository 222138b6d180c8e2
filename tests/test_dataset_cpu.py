"""CPU-only plumbing of the data formats on either side of the hot path (BASELINE.json config 1: a TUM
sequence with ground-truth pose + depth): TUM -> HIVE conversion, the HIVE loader's depth contract, errors."""
import os
import sys

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tum_fixture import write_tum_sequence  # noqa: E402


def test_tum_to_hive_conversion(tmp_path):
    from hive_amd.dataset_adaptors import TUMAdaptor, get_dataset
    from hive_amd.geometric import Trajectory
    from hive_amd.io import HiveDataset
    tum, out = str(tmp_path / "tum"), str(tmp_path / "hive")
    seq = write_tum_sequence(tum, num_frames=4)
    adaptor = TUMAdaptor(tum, out, num_frames=3)
    assert adaptor.num_frames == 3 and len(adaptor.image_filenames) == 4
    ds = adaptor.convert()
    assert isinstance(ds, HiveDataset) and ds.num_frames == 3 and len(ds.rgb_dataset) == 3
    assert ds.camera_matrix.dtype == np.float32 and np.allclose(ds.camera_matrix, TUMAdaptor.intrinsic_matrix)
    assert ds.camera_trajectory.values.dtype == np.float32 and ds.camera_trajectory.shape == (3, 7)
    # frames and depth survive the round trip: mm quantisation of the 1/5000 m TUM depth, > 10 m -> 0
    assert np.array_equal(ds.rgb_dataset[1], seq["color"][1])
    d = ds.depth_dataset[1]
    assert d.dtype == np.float32 and d.shape == (480, 640)
    raw = np.round(seq["depth"][1].astype(np.float64) * 5000.0).astype(np.uint16)
    expect = np.float32(1. / 1000.) * (1000 * (raw * (1.0 / 5000.0))).astype(np.uint16).astype(np.float32)  # dataset_adaptors.py:762-764
    assert np.array_equal(d, expect)
    assert (ds.mask_dataset[0] == 0).all() and not ds.has_inpainted_frame_data
    assert ds.bg_depth_dataset is ds.depth_dataset
    # trajectory: cam-to-world ground truth -> re-based, inverted (world-to-cam), rotated -90 deg about x
    gt = Trajectory(np.array([np.hstack([Rotation.from_matrix(p[:3, :3]).as_quat(), p[:3, 3]]) for p in seq["poses"][:3]]))
    rot = np.eye(4)
    rot[:3, :3] = Rotation.from_euler('xyz', [-90, 0, 0], degrees=True).as_matrix()
    expect_traj = gt.normalise_position().inverse().apply(rot).to_homogenous_transforms()
    got = Trajectory(ds.camera_trajectory.values.astype(np.float64)).to_homogenous_transforms()
    np.testing.assert_allclose(got, expect_traj, atol=2e-5)
    # dispatch: an existing HIVE folder is opened as is; a TUM folder is converted
    assert get_dataset(out, out).num_frames == 3
    assert get_dataset(tum, str(tmp_path / "hive2"), num_frames=2).num_frames == 2
    with pytest.raises(RuntimeError):
        get_dataset(str(tmp_path), str(tmp_path / "x"))


def test_tum_depth_to_mm_all_uint16_values():
    """The TUM adaptor's raw -> millimetre conversion over every uint16 value, in the reference's operation order
    (/root/reference/hive/dataset_adaptors.py:762-764): `(1000 * (raw * (1 / 5000))).astype(uint16)`.  The one-step form
    `raw / 5000 * 1000` differs for 41 raw values (e.g. 10005 -> 2000 instead of 2001)."""
    from hive_amd.dataset_adaptors import TUMAdaptor
    adaptor = TUMAdaptor.__new__(TUMAdaptor)
    adaptor.depth_scale_factor = 1.0 / 5000.0
    raw = np.arange(65536, dtype=np.uint16).reshape(256, 256)
    got = adaptor.depth_to_mm(raw)
    scale = 1.0 / 5000.0
    expect = np.array([int(1000 * (float(r) * scale)) for r in range(65536)], np.uint16).reshape(256, 256)  # scalar restatement
    assert got.dtype == np.uint16 and np.array_equal(got, expect)
    assert got.reshape(-1)[10005] == 2001
    one_step = (raw.astype(np.float64) * scale / (1.0 / 1000.0)).astype(np.uint16)
    assert 0 < int((one_step != got).sum()) < 100, "the sweep must cover the values where the operation order matters"


def test_hive_dataset_validation_and_depth_contract(tmp_path):
    from PIL import Image
    from hive_amd.io import DatasetMetadata, HiveDataset
    root = tmp_path / "ds"
    with pytest.raises(RuntimeError):
        HiveDataset(str(root))
    for f in ("rgb", "depth", "mask"):
        os.makedirs(root / f)
    with pytest.raises(RuntimeError, match="missing the file"):
        HiveDataset(str(root))
    DatasetMetadata(num_frames=1, fps=30.0, width=4, height=2, max_depth=10.0).save(str(root / "metadata.json"))
    np.savetxt(root / "camera_matrix.txt", np.eye(3))
    np.savetxt(root / "camera_trajectory.txt", np.array([[0, 0, 0, 1, 0, 0, 0.0]]))
    Image.fromarray(np.zeros((2, 4, 3), np.uint8)).save(root / "rgb" / "000000.png")
    Image.fromarray(np.zeros((2, 4), np.uint8)).save(root / "mask" / "000000.png")
    Image.fromarray(np.array([[0, 1, 999, 1000], [9999, 10000, 10001, 65535]], np.uint16)).save(root / "depth" / "000000.png")
    ds = HiveDataset(str(root))
    d = ds.depth_dataset[0]
    s = np.float32(1. / 1000.)
    assert np.array_equal(d, np.array([[0, s * 1, s * 999, s * 1000], [s * 9999, s * 10000, 0, 0]], np.float32))  # > max_depth -> 0
    assert ds.camera_trajectory.shape == (1, 7) and ds.metadata == DatasetMetadata.load(str(root / "metadata.json"))
    with pytest.raises(ValueError):
        DatasetMetadata(num_frames=0, fps=1.0, width=1, height=1)
    # threshold shortcuts of select_key_frames need no GPU
    assert ds.select_key_frames(threshold=0.0) == [0] and ds.select_key_frames(threshold=1.0) == [0]
    with pytest.raises(ValueError):
        ds.select_key_frames(threshold=1.5)


def test_ply_writer(tmp_path):
    from hive_amd.pipeline import write_ply
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    f = np.array([[0, 1, 2]], np.int32)
    path = str(tmp_path / "m.ply")
    write_ply(path, v, f, vertex_colors=np.array([[255, 0, 0]] * 3, np.uint8), vertex_normals=np.array([[0, 0, 1.0]] * 3))
    blob = open(path, "rb").read()
    head, body = blob.split(b"end_header\n")
    assert b"element vertex 3" in head and b"element face 1" in head and b"property uchar red" in head
    assert len(body) == 3 * (24 + 3) + 1 * (1 + 12)
