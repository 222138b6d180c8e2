"""HIP geometry kernels through the C ABI, against the golden vectors of the real reference module
and against the CPU oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
F64_TOL = dict(rtol=1e-12, atol=1e-12)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "geometric_48x64.npz"))


@pytest.mark.parametrize("kname", ["k32", "k64"])
def test_point_cloud_from_depth_matches_reference(gpu_ctx, gold, kname):
    from hive_amd.geometric import point_cloud_from_depth, point_cloud_from_rgbd
    K = gold["K32"] if kname == "k32" else gold["K64"]
    pts = point_cloud_from_depth(gold["depth"], gold["mask"], K, gold["R"], gold["t"])
    assert pts.dtype == np.float64 and pts.shape == gold[f"pc_{kname}"].shape
    np.testing.assert_allclose(pts, gold[f"pc_{kname}"], **F64_TOL)
    pts2, col = point_cloud_from_rgbd(gold["rgb"], gold["depth"], gold["mask"], K, gold["R"], gold["t"])
    np.testing.assert_allclose(pts2, gold[f"pcrgbd_pts_{kname}"], **F64_TOL)
    assert col.dtype == np.uint8 and np.array_equal(col, gold[f"pcrgbd_col_{kname}"])


def test_point_cloud_defaults(gpu_ctx, gold):
    from hive_amd.geometric import point_cloud_from_depth
    np.testing.assert_allclose(point_cloud_from_depth(gold["depth"], gold["mask"], gold["K32"]), gold["pc_identity"], **F64_TOL)
    pts = point_cloud_from_depth(gold["depth"], np.ones_like(gold["mask"]), gold["K32"], gold["R"], gold["t"])
    np.testing.assert_allclose(pts, gold["pc_allmask"], **F64_TOL)
    empty = point_cloud_from_depth(np.zeros((48, 64), np.float32), gold["mask"], gold["K32"])
    assert empty.shape == (0, 3)
    with pytest.raises(AssertionError):
        point_cloud_from_depth(gold["depth"], gold["mask"], gold["K32"], gold["R"], gold["t"].reshape(3))


def test_image2world_matches_reference(gpu_ctx, gold):
    from hive_amd.geometric import image2world
    out = image2world(gold["i2w_uv"], gold["i2w_d"], gold["K32"], gold["R"], gold["t"])
    np.testing.assert_allclose(out, gold["i2w"], **F64_TOL)
    out = image2world(gold["i2w_uv"], gold["i2w_d"], gold["K32"], gold["R"], gold["t"], scale_factor=2.0)
    np.testing.assert_allclose(out, gold["i2w_scale"], **F64_TOL)
    with pytest.raises(AssertionError):
        image2world(gold["i2w_uv"], gold["i2w_d"][:-1], gold["K32"], gold["R"], gold["t"])


def test_world2image_matches_reference(gpu_ctx, gold):
    from hive_amd.geometric import world2image
    uv, dep = world2image(gold["pc_k32"], gold["K32"], gold["R2"], gold["t2"])
    assert uv.dtype == np.int32 and np.array_equal(uv, gold["w2i_uv_i32"])
    np.testing.assert_allclose(dep, gold["w2i_depth"], **F64_TOL)
    uvf, _ = world2image(gold["pc_k32"], gold["K32"], gold["R2"], gold["t2"], dtype=np.float64)
    np.testing.assert_allclose(uvf, gold["w2i_uv_f64"], rtol=1e-11, atol=1e-10)
    uvs, _ = world2image(gold["pc_k32"], gold["K32"], gold["R2"], gold["t2"], scale_factor=2.0)
    assert np.array_equal(uvs, gold["w2i_uv_scaled"])
    tie, _ = world2image(gold["tie_pts"], gold["tie_K"])
    assert np.array_equal(tie, gold["tie_uv"]), "half-pixel ties must round half-to-even like np.round"
    e_uv, e_d = world2image(np.zeros((0, 3)), gold["K32"])
    assert e_uv.shape == (0, 2) and e_d.shape == (0,)


def test_full_resolution_checksums_and_oracle(gpu_ctx, oracle_lib):
    from hive_amd.geometric import point_cloud_from_depth, world2image
    chk = np.load(os.path.join(GOLDEN, "geometric_full_checksums.npz"))
    rng = np.random.default_rng(1)
    H, W = 480, 640
    K = np.array([[580.0, 0, 319.5], [0, 580.0, 239.5], [0, 0, 1]], np.float32)
    depth = rng.uniform(0.5, 5.0, size=(H, W)).astype(np.float32)
    depth[rng.random((H, W)) < 0.10] = 0.0
    mask = rng.random((H, W)) < 0.9
    pc = point_cloud_from_depth(depth, mask, K, chk["R"], chk["t"])
    assert len(pc) == int(chk["n"])
    np.testing.assert_allclose(pc[chk["idx"]], chk["pc_rows"], **F64_TOL)
    np.testing.assert_allclose(pc.sum(axis=0), chk["pc_sum"], rtol=1e-9)
    o_pc, _ = oracle_lib.unproject(depth, mask, np.linalg.inv(K).astype(np.float64), chk["R"], chk["t"])
    np.testing.assert_allclose(pc, o_pc, rtol=1e-14, atol=1e-14)
    uv, dep = world2image(pc, K, chk["R"], chk["t"])
    assert np.array_equal(uv[chk["idx"]], chk["uv_rows"])
    assert np.array_equal(uv.astype(np.int64).sum(axis=0), chk["uv_sum"])
    o_uv, o_dep = oracle_lib.project(pc, K, chk["R"], chk["t"])
    assert np.array_equal(uv, o_uv)
    np.testing.assert_allclose(dep, o_dep, rtol=1e-14)


def test_unproject_is_reentrant_from_threads(gpu_ctx, gold):
    """The reference calls these from a ThreadPool (pipeline.py:491): each thread gets its own context."""
    from multiprocessing.pool import ThreadPool
    from hive_amd.geometric import point_cloud_from_depth

    def work(_):
        return point_cloud_from_depth(gold["depth"], gold["mask"], gold["K32"], gold["R"], gold["t"])

    with ThreadPool(4) as pool:
        results = pool.map(work, range(16))
    for r in results:
        np.testing.assert_allclose(r, gold["pc_k32"], **F64_TOL)


def test_view_frustum_matches_oracle(gpu_ctx, oracle_lib, small_sequence):
    import torch
    from hive_amd.fusion import get_view_frustum
    seq = small_sequence
    for i in range(3):
        f = get_view_frustum(seq["depth"][i], seq["K"], seq["poses"][i])
        o = oracle_lib.view_frustum(seq["depth"][i], seq["K"], seq["poses"][i])
        assert f.shape == (3, 5) and np.array_equal(f, o)
    d = torch.from_numpy(seq["depth"][0]).cuda()
    assert np.array_equal(get_view_frustum(d, seq["K"], seq["poses"][0]), oracle_lib.view_frustum(seq["depth"][0], seq["K"], seq["poses"][0]))


def test_dilate_mask_matches_oracle(gpu_ctx, oracle_lib):
    from hive_amd.image_processing import dilate_mask
    from hive_amd.options import MaskDilationOptions
    rng = np.random.default_rng(11)
    for shape in ((48, 64), (37, 53)):
        m = rng.random(shape) < 0.01
        for it in (0, 1, 10):
            out = dilate_mask(m, MaskDilationOptions(num_iterations=it))
            assert out.dtype == bool and np.array_equal(out, oracle_lib.dilate_mask(m, it))
    ids = (rng.random((48, 64)) < 0.01).astype(np.uint8) * 3  # instance-id masks are uint8 ids
    assert np.array_equal(dilate_mask(ids, MaskDilationOptions(num_iterations=2)), oracle_lib.dilate_mask(ids, 2))
    with pytest.raises(AssertionError):
        dilate_mask(np.zeros((2, 3, 4)), MaskDilationOptions(1))


def test_dilate_mask_any_structuring_element_matches_oracle(gpu_ctx, oracle_lib):
    """`MaskDilationOptions(num_iterations, dilation_filter)` with filters other than the default 3x3 box
    (/root/reference/hive/options.py:245-268): cross, ellipse-like, asymmetric, even-sized, a non-square rectangle (separable path)
    and an element without its own centre -- bit-exact vs the oracle's literal iterated cv2-style dilation."""
    from hive_amd.image_processing import dilate_mask
    from hive_amd.options import MaskDilationOptions
    rng = np.random.default_rng(12)
    cross = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], np.uint8)
    ellipse5 = np.array([[0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [0, 0, 1, 0, 0]], np.uint8)  # cv2.MORPH_ELLIPSE (5, 5)
    lopsided = np.array([[1, 0, 0], [0, 0, 0], [0, 0, 1]], np.uint8)  # no centre: the mask itself is not part of its dilation
    even = np.ones((2, 4), np.uint8)
    rect = np.ones((3, 5), np.uint8)
    ring = np.ones((7, 7), np.uint8)
    ring[1:6, 1:6] = 0
    for shape in ((48, 64), (37, 53)):
        m = rng.random(shape) < 0.01
        m[0, 0] = m[-1, -1] = m[0, shape[1] // 2] = True  # the image border
        for se in (cross, ellipse5, lopsided, even, rect, ring, np.ones((1, 1), np.uint8)):
            for it in (0, 1, 2, 5):
                out = dilate_mask(m, MaskDilationOptions(num_iterations=it, dilation_filter=se))
                assert out.dtype == bool and np.array_equal(out, oracle_lib.dilate_mask_se(m, se, it)), (se.shape, it)
    box = np.ones((3, 3), np.uint8)
    assert np.array_equal(dilate_mask(m, MaskDilationOptions(4, box)), oracle_lib.dilate_mask(m, 4))
    with pytest.raises(ValueError):
        dilate_mask(m, MaskDilationOptions(1, np.zeros((3, 3), np.uint8)))
    with pytest.raises(ValueError):
        dilate_mask(m, MaskDilationOptions(1, np.ones((33, 3), np.uint8)))


def test_depth_quantize_matches_oracle(gpu_ctx, oracle_lib):
    import torch
    from hive_amd import _lib
    rng = np.random.default_rng(2)
    H, W = 48, 64
    d = rng.uniform(0.0, 12.0, (H, W)).astype(np.float32)
    mask = rng.random((H, W)) < 0.2
    o_mm, o_m = oracle_lib.depth_quantize(d, mask=mask)
    for dtype, code in ((torch.float32, _lib.F32), (torch.float16, _lib.F16), (torch.bfloat16, _lib.BF16)):
        dd = torch.from_numpy(d).cuda().to(dtype)
        mm = torch.empty((H, W), dtype=torch.int16, device="cuda")
        m = torch.empty((H, W), dtype=torch.float32, device="cuda")
        mk = torch.from_numpy(mask.astype(np.uint8)).cuda()
        gpu_ctx.check(gpu_ctx.lib.hive_depth_quantize(gpu_ctx.handle, dd.data_ptr(), code, H, W, 1.0 / 1000.0, 10.0, mk.data_ptr(),
                                                      mm.data_ptr(), m.data_ptr()))
        torch.cuda.synchronize()
        e_mm, e_m = oracle_lib.depth_quantize(dd.float().cpu().numpy(), mask=mask)
        assert np.array_equal(mm.cpu().numpy().view(np.uint16), e_mm)
        assert np.array_equal(m.cpu().numpy(), e_m)
    assert np.array_equal(o_mm, oracle_lib.depth_quantize(d, mask=mask)[0]) and o_m.dtype == np.float32
