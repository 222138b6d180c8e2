"""Foreground per-frame meshing kernels (csrc/fgmesh.hip, through the C ABI) against numpy restatements of
/root/reference/hive/pipeline.py:651-694, 782-808 -- and against scipy's Delaunay (the reference's triangulator, available in
this image), to quantify what the implicit pixel-grid triangulation does differently from Qhull."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _numpy_filter_faces(points2d, depth, faces, max_px, max_depth):
    """pipeline.py:684-692, literally."""
    pixel_distances = np.linalg.norm(points2d[faces[:, [0, 2, 0]]] - points2d[faces[:, [1, 1, 2]]], axis=-1)
    depth_proj = depth.reshape((*depth.shape, 1))
    depth_distances = np.linalg.norm(depth_proj[faces[:, [0, 2, 0]]] - depth_proj[faces[:, [1, 1, 2]]], axis=-1)
    valid = np.all((pixel_distances <= max_px) & (depth_distances <= max_depth), axis=1)
    return faces[valid]


def _numpy_grid_faces(depth, mask):
    """The implicit triangulation's rule, restated with loops: 2 x 2 blocks in row-major order; all four corners valid -> the
    two halves across the b-c diagonal, exactly three -> their triangle; wound with a negative (u, v) cross product."""
    valid = mask & (depth > 0)
    vid = -np.ones(depth.shape, np.int64)
    vid[valid] = np.arange(int(valid.sum()))
    H, W = depth.shape
    faces = []
    for v in range(H - 1):
        for u in range(W - 1):
            a, b, c, d = (v, u), (v, u + 1), (v + 1, u), (v + 1, u + 1)
            ok = [valid[p] for p in (a, b, c, d)]
            if sum(ok) == 4:
                faces += [(a, c, b), (b, c, d)]
            elif sum(ok) == 3:
                faces.append({0: (b, c, d), 1: (a, c, d), 2: (a, d, b), 3: (a, c, b)}[ok.index(False)])
    return np.array([[vid[p] for p in f] for f in faces], np.int64).reshape(-1, 3), valid


def _scene(seed, H=40, W=56):
    rng = np.random.default_rng(seed)
    v, u = np.mgrid[0:H, 0:W]
    mask = ((v - H / 2) / (H * 0.42)) ** 2 + ((u - W / 2) / (W * 0.4)) ** 2 <= 1
    mask[rng.integers(5, H - 5, 6), rng.integers(5, W - 5, 6)] = False          # one-pixel holes
    mask[H // 3:H // 3 + 2, W // 2:W // 2 + 7] = False                           # a slit
    depth = (2.0 + 0.01 * u + 0.004 * v + 0.6 * (u > 0.6 * W) + rng.normal(0, 0.004, (H, W))).astype(np.float32)  # a depth step
    depth[rng.random((H, W)) < 0.02] = 0.0
    return depth, mask


@pytest.mark.parametrize("seed", [0, 1])
@pytest.mark.parametrize("limits", [(2, 0.1), (1.2, 0.05), (5, 10.0)])
def test_grid_faces_equal_rule_plus_reference_filter(gpu_ctx, seed, limits):
    """Face LIST (order included) == implicit rule restated in numpy, then the reference's _filter_faces verbatim."""
    import torch
    from hive_amd import foreground
    from hive_amd.options import MeshFilteringOptions
    depth, mask = _scene(seed)
    opts = MeshFilteringOptions(max_pixel_distance=limits[0], max_depth_distance=limits[1])
    all_faces, valid = _numpy_grid_faces(depth, mask)
    vv, uu = valid.nonzero()
    points2d = np.vstack((uu, vv)).T
    expect = _numpy_filter_faces(points2d, depth[valid], all_faces, *limits)
    got, n_vert = foreground.grid_faces(depth, mask, opts, ctx=gpu_ctx, return_vertex_count=True)
    assert n_vert == int(valid.sum()) and got.dtype == np.int32
    assert np.array_equal(got, expect), f"{len(got)} faces vs {len(expect)}"
    got_dev = foreground.grid_faces(torch.from_numpy(depth).cuda(), torch.from_numpy(mask).cuda(), opts, ctx=gpu_ctx)
    assert np.array_equal(got_dev.cpu().numpy(), expect)
    if limits[0] >= 2:
        assert len(expect) > 1000
        p = points2d[got].astype(np.float64)
        cross = (p[:, 1, 0] - p[:, 0, 0]) * (p[:, 2, 1] - p[:, 0, 1]) - (p[:, 1, 1] - p[:, 0, 1]) * (p[:, 2, 0] - p[:, 0, 0])
        assert (cross < 0).all(), "winding of the reference's reversed Delaunay simplices"
    else:
        assert len(expect) == 0, "max_pixel_distance below sqrt(2) removes every face (each has a diagonal)"


def test_filter_faces_on_scipy_delaunay_equals_reference(gpu_ctx):
    """hive_filter_faces on the reference's OWN triangulation (scipy Delaunay, reversed simplices, pipeline.py:661-667) == the
    reference's numpy filter, face for face; and the implicit grid triangulation covers the same area except where Qhull
    bridges one-pixel holes."""
    from scipy.spatial import Delaunay
    from hive_amd import foreground
    from hive_amd.options import MeshFilteringOptions
    depth, mask = _scene(3)
    valid = mask & (depth > 0)
    vv, uu = valid.nonzero()
    points2d = np.vstack((uu, vv)).T
    faces = np.asarray(Delaunay(points2d).simplices)[:, ::-1]
    opts = MeshFilteringOptions()
    expect = _numpy_filter_faces(points2d, depth[valid], faces, opts.max_pixel_distance, opts.max_depth_distance)
    got = foreground.filter_faces(points2d, depth[valid], faces, opts, ctx=gpu_ctx)
    assert np.array_equal(got, expect) and 0 < len(expect) < len(faces)

    def area2(f):  # twice the area, integer
        p = points2d[f].astype(np.int64)
        return np.abs((p[:, 1, 0] - p[:, 0, 0]) * (p[:, 2, 1] - p[:, 0, 1]) - (p[:, 1, 1] - p[:, 0, 1]) * (p[:, 2, 0] - p[:, 0, 0]))

    grid = foreground.grid_faces(depth, mask, opts, ctx=gpu_ctx)
    unit = area2(expect) == 1                      # Qhull's faces inside fully valid / three-corner blocks
    bridging = (~unit).sum()                       # (sqrt 2, sqrt 2, 2) triangles across a missing pixel: area 1 each, twice = 2
    assert (area2(grid) == 1).all()
    assert bridging <= 0.05 * len(expect), "hole-bridging faces are a small share of the reference's mesh (53 of 2071 on this mask: 6 holes and a slit)"
    # same covered cells: every unit face of either triangulation lies in one 2 x 2 block; compare the per-block face counts
    def per_block(f):
        p = points2d[f]
        key = p[:, :, 1].min(axis=1) * 100000 + p[:, :, 0].min(axis=1)
        return np.unique(key, return_counts=True)
    kq, cq = per_block(expect[unit])
    kg, cg = per_block(grid)
    common, iq, ig = np.intersect1d(kq, kg, return_indices=True)
    assert len(common) >= 0.97 * max(len(kq), len(kg)), "the two triangulations fill the same blocks (bar the depth-filtered diagonals)"
    assert (cq[iq] == cg[ig]).mean() > 0.97


def test_triangulate_faces_matches_delaunay_face_count_on_solid_region(gpu_ctx):
    from scipy.spatial import Delaunay
    from hive_amd import foreground
    v, u = np.mgrid[0:12, 0:17]
    points = np.vstack((u.reshape(-1) + 5, v.reshape(-1) + 3)).T  # a solid rectangle, offset from the origin
    faces = foreground.triangulate_faces(points, ctx=gpu_ctx)
    assert len(faces) == len(Delaunay(points).simplices) == 2 * 11 * 16
    assert faces.min() == 0 and faces.max() == len(points) - 1


def test_texture_window_and_uv_match_numpy(gpu_ctx):
    """_get_mesh_texture_and_uv (pipeline.py:797-808) restated with hive_amd.geometric.world2image (pinned by golden vectors)."""
    from scipy.spatial.transform import Rotation
    from hive_amd import foreground, synthetic
    from hive_amd.geometric import point_cloud_from_depth, world2image
    rng = np.random.default_rng(4)
    H, W = 120, 160
    K = synthetic.scaled_intrinsics(H, W)
    depth = rng.uniform(1.0, 3.0, (H, W)).astype(np.float32)
    mask = np.zeros((H, W), bool)
    mask[30:75, 50:131] = True
    R = Rotation.from_euler("xyz", [3, -8, 2], degrees=True).as_matrix()
    t = np.array([[0.1], [-0.05], [0.2]])
    verts = point_cloud_from_depth(depth, mask, K, R, t)
    image = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    uv, _ = world2image(verts, K, R, t, 1.0)  # default dtype: int32 pixels, as the reference calls it (pipeline.py:797)
    mn = np.min(np.round(uv), axis=0).astype(int)
    mx = np.max(np.round(uv), axis=0).astype(int) + 1
    exp_tex = image[mn[1]:mx[1], mn[0]:mx[0], :]
    exp_uv = uv.copy()
    exp_uv -= np.min(np.round(uv), axis=0)
    tex, got_uv = foreground.get_mesh_texture_and_uv(verts, image, K, R, t, ctx=gpu_ctx)
    assert tex.shape == exp_tex.shape and np.array_equal(tex, exp_tex)
    assert tex.shape[:2] == (45, 81), "projecting the unprojected pixels gives back the mask's bounding box"
    assert got_uv.dtype == np.int32 and np.array_equal(got_uv, exp_uv)
