"""Foreground per-frame meshing kernels (csrc/fgmesh.hip, through the C ABI) against numpy restatements of
/root/reference/hive/pipeline.py:651-694, 782-808 -- and against scipy's Delaunay (the reference's triangulator, available in
this image): the implicit pixel-grid triangulation gives the reference's face set after its filter, up to the diagonal Qhull picks
inside four co-circular lattice points."""
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
    """The implicit triangulation's rule, restated with loops.  Per pixel (v, u) in row-major order: first the 2 x 2 block whose top-left
    corner it is -- all four corners valid -> the two halves across the b-c diagonal, exactly three -> their triangle -- then, where the
    pixel itself is invalid, the triangles that bridge it (a 4-neighbour outside the image counts as invalid): all four 4-neighbours valid -> the diamond
    split west - east, exactly three -> their triangle.  Wound with a negative (u, v) cross product."""
    valid = mask & (depth > 0)
    vid = -np.ones(depth.shape, np.int64)
    vid[valid] = np.arange(int(valid.sum()))
    H, W = depth.shape
    faces = []
    for v in range(H):
        for u in range(W):
            if v + 1 < H and u + 1 < W:
                a, b, c, d = (v, u), (v, u + 1), (v + 1, u), (v + 1, u + 1)
                ok = [valid[p] for p in (a, b, c, d)]
                if sum(ok) == 4:
                    faces += [(a, c, b), (b, c, d)]
                elif sum(ok) == 3:
                    faces.append({0: (b, c, d), 1: (a, c, d), 2: (a, d, b), 3: (a, c, b)}[ok.index(False)])
            if not valid[v, u]:
                n, s, w, e = (v - 1, u), (v + 1, u), (v, u - 1), (v, u + 1)
                ok = {k: (0 <= p[0] < H and 0 <= p[1] < W and bool(valid[p])) for k, p in zip("nswe", (n, s, w, e))}
                if sum(ok.values()) == 4:
                    faces += [(w, e, n), (w, s, e)]
                elif sum(ok.values()) == 3:
                    faces.append({"n": (w, s, e), "s": (w, e, n), "w": (n, s, e), "e": (n, w, s)}[[k for k in "nswe" if not ok[k]][0]])
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


def _cells(points2d, faces):
    """Faces keyed by the co-circular cell they tile: ('sq', u, v) = the unit square with that top-left corner (sides 1, 1, sqrt 2),
    ('dm', u, v) = the diamond around pixel (u, v) (sides sqrt 2, sqrt 2, 2: the pixel is the midpoint of the side of length 2).  Value: the
    number of faces in the cell and the union of their vertices -- what does not depend on which diagonal splits a four-point cell."""
    cells = {}
    for f in faces:
        p = points2d[f].astype(np.int64)
        d2 = sorted(int(((p[i] - p[j]) ** 2).sum()) for i, j in ((0, 1), (1, 2), (0, 2)))
        if d2 == [1, 1, 2]:
            key = ("sq", int(p[:, 0].min()), int(p[:, 1].min()))
        else:
            assert d2 == [2, 2, 4], d2
            i, j = [(i, j) for i, j in ((0, 1), (1, 2), (0, 2)) if ((p[i] - p[j]) ** 2).sum() == 4][0]
            mid = (p[i] + p[j]) // 2
            key = ("dm", int(mid[0]), int(mid[1]))
        n, verts = cells.get(key, (0, frozenset()))
        cells[key] = (n + 1, verts | frozenset(int(x) for x in f))
    return cells


def test_border_holes_are_bridged_like_delaunay(gpu_ctx):
    """One-pixel gaps ON the image border (all four border rows / columns): Delaunay bridges such a gap with the (sqrt 2, sqrt 2, 2) triangle of its three
    in-image neighbours -- its long side lies on the convex hull -- and that triangle passes the default 2-pixel filter.  The implicit triangulation treats a
    neighbour outside the image as invalid and emits the same triangle: face sets compared per cell against scipy + the reference's filter."""
    from scipy.spatial import Delaunay
    from hive_amd import foreground
    from hive_amd.options import MeshFilteringOptions
    H, W = 24, 32
    mask = np.ones((H, W), bool)
    for u in (3, 9, 20, 27):
        mask[0, u] = mask[H - 1, u + 1] = False
    for v in (4, 11, 17):
        mask[v, 0] = mask[v + 2, W - 1] = False
    mask[10, 15] = False  # (and an interior hole)
    depth = np.full((H, W), 2.0, np.float32)
    valid = mask & (depth > 0)
    vv, uu = valid.nonzero()
    points2d = np.vstack((uu, vv)).T
    opts = MeshFilteringOptions()
    faces = np.asarray(Delaunay(points2d).simplices)[:, ::-1]
    expect = _numpy_filter_faces(points2d, depth[valid], faces, opts.max_pixel_distance, opts.max_depth_distance)
    grid = foreground.grid_faces(depth, mask, opts, ctx=gpu_ctx)
    ref_cells, grid_cells = _cells(points2d, expect), _cells(points2d, grid)
    border = [k for k in ref_cells if k[0] == "dm" and (k[1] in (0, W - 1) or k[2] in (0, H - 1))]
    assert len(border) == 14, "scipy bridges every border gap"
    assert ref_cells == grid_cells
    rule, _ = _numpy_grid_faces(depth, mask)
    assert np.array_equal(grid, _numpy_filter_faces(points2d, depth[valid], rule, opts.max_pixel_distance, opts.max_depth_distance))


@pytest.mark.parametrize("seed", [3, 5])
def test_face_set_equals_scipy_delaunay_plus_reference_filter(gpu_ctx, seed):
    """"face set after filtering".  hive_filter_faces on the reference's OWN triangulation (scipy Delaunay, reversed simplices,
    pipeline.py:661-667) == the reference's numpy filter, face for face; and the implicit triangulation of hive_grid_mesh produces the SAME
    face set as that, up to the one freedom Delaunay itself has on a lattice: which diagonal splits four co-circular points (a fully valid
    unit square, or the diamond around a one-pixel hole whose four neighbours are valid).  Both sets are therefore compared per cell
    (face count + vertex set); a four-point cell whose two possible splits are filtered differently (a depth step along one diagonal
    only) depends on Qhull's arbitrary choice and is compared as "either split"."""
    from scipy.spatial import Delaunay
    from hive_amd import foreground
    from hive_amd.options import MeshFilteringOptions
    depth, mask = _scene(seed)
    valid = mask & (depth > 0)
    vv, uu = valid.nonzero()
    points2d = np.vstack((uu, vv)).T
    faces = np.asarray(Delaunay(points2d).simplices)[:, ::-1]
    opts = MeshFilteringOptions()
    expect = _numpy_filter_faces(points2d, depth[valid], faces, opts.max_pixel_distance, opts.max_depth_distance)
    got = foreground.filter_faces(points2d, depth[valid], faces, opts, ctx=gpu_ctx)
    assert np.array_equal(got, expect) and 0 < len(expect) < len(faces)

    grid = foreground.grid_faces(depth, mask, opts, ctx=gpu_ctx)
    ref_cells, grid_cells = _cells(points2d, expect), _cells(points2d, grid)
    n_bridging = sum(n for (kind, _, _), (n, _) in ref_cells.items() if kind == "dm")
    assert n_bridging >= 6, "the mask's one-pixel holes are bridged by the reference's triangulation"
    # cells that differ can only be four-point cells where the split decides what the depth filter keeps
    vid = -np.ones(depth.shape, np.int64)
    vid[valid] = np.arange(int(valid.sum()))

    def splits(key):  # both ways of splitting a four-point cell, each filtered like the reference does
        kind, u, v = key
        if kind == "sq":
            a, b, c, d = (v, u), (v, u + 1), (v + 1, u), (v + 1, u + 1)
            options = [[(a, c, b), (b, c, d)], [(a, c, d), (a, d, b)]]
        else:
            n, s, w, e = (v - 1, u), (v + 1, u), (v, u - 1), (v, u + 1)
            options = [[(w, e, n), (w, s, e)], [(n, s, e), (n, w, s)]]
        out = []
        for tris in options:
            if any(not (0 <= p[0] < depth.shape[0] and 0 <= p[1] < depth.shape[1]) or vid[p] < 0 for t in tris for p in t):
                return None  # not a four-point cell
            f = np.array([[vid[p] for p in t] for t in tris], np.int64)
            kept = _numpy_filter_faces(points2d, depth[valid], f, opts.max_pixel_distance, opts.max_depth_distance)
            out.append((len(kept), frozenset(int(x) for x in kept.ravel())))
        return out

    ambiguous = 0
    for key in set(ref_cells) | set(grid_cells):
        r, g = ref_cells.get(key, (0, frozenset())), grid_cells.get(key, (0, frozenset()))
        if r == g:
            continue
        both = splits(key)
        assert both is not None and r in both and g in both, f"cell {key}: reference {r}, grid {g}"
        ambiguous += 1
    assert ambiguous <= 0.01 * len(ref_cells), "splits that the depth filter treats differently are rare (cells on a depth step)"
    # and as plain numbers: the same face count up to those cells
    assert abs(len(grid) - len(expect)) <= 2 * ambiguous


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


@pytest.mark.parametrize("on_device", [False, True])
def test_frame_mesh_in_one_call_equals_the_separate_functions(gpu_ctx, on_device):
    """hive_fg_frame_mesh (one call, device-resident, one read-back) == point_cloud_from_depth + grid_faces + get_mesh_texture_and_uv on the same object
    mask, bit for bit: vertices (float64), faces, uv, crop box and texture -- /root/reference/hive/pipeline.py:383-461 without its CPU-library stages."""
    import torch
    from hive_amd import foreground, geometric, synthetic
    from hive_amd.options import MeshFilteringOptions
    seq = synthetic.make_sequence(num_frames=2, height=240, width=320, yaw_step_deg=20.0)
    masks = synthetic.ellipse_masks(2, 240, 320, num_objects=2, seed=7)
    K = seq["K"]  # float32, as loaded from disk: inverted in its own dtype
    opts = MeshFilteringOptions()
    buffers = foreground.FrameMeshBuffers(240, 320)
    for f in range(2):
        w2c = np.linalg.inv(seq["poses"][f])
        R, t = w2c[:3, :3], w2c[:3, 3:4]
        depth, rgb = seq["depth"][f], seq["color"][f]
        for obj in (1, 2, 0):  # (0: no pixel of the mask is set below)
            mask = (masks[f] == obj) if obj else np.zeros_like(masks[f], bool)
            want_v = geometric.point_cloud_from_depth(depth, mask, K, R, t)
            if on_device:
                got = foreground.frame_mesh(torch.from_numpy(depth).cuda(), torch.from_numpy(mask).cuda(), torch.from_numpy(rgb).cuda(), K, R, t, opts, ctx=gpu_ctx,
                                            buffers=buffers)
            else:
                got = foreground.frame_mesh(depth, mask, rgb, K, R, t, opts, ctx=gpu_ctx)
            assert np.array_equal(got["vertices"].cpu().numpy(), want_v)
            if len(want_v) == 0:
                assert got["faces"].shape[0] == 0 and got["texture"] is None
                continue
            want_f = foreground.grid_faces(depth, mask, opts, ctx=gpu_ctx)
            want_tex, want_uv = foreground.get_mesh_texture_and_uv(want_v, rgb, K, R, t, ctx=gpu_ctx)
            assert len(want_f) > 100
            assert np.array_equal(got["faces"].cpu().numpy(), want_f)
            assert np.array_equal(got["uv"].cpu().numpy(), want_uv)
            assert np.array_equal(got["texture"].cpu().numpy(), want_tex)
            vv, uu = (mask & (depth > 0)).nonzero()
            assert got["bbox"] == (uu.min(), vv.min(), uu.max() + 1, vv.max() + 1), "an unprojected pixel projects back onto itself"


def _reference_pack_textures(textures_atlas, uvs_atlas, n_rows=1):
    """What Pipeline._pack_textures computes for n_rows = 1 (pipeline.py:811-866), restated directly: one row, u offsets by the running width, normalisation."""
    heights, widths = [t.shape[0] for t in textures_atlas], [t.shape[1] for t in textures_atlas]
    atlas = np.zeros((max(heights), sum(widths), 3), np.uint8)
    out, x = [], 0
    for t, uv in zip(textures_atlas, uvs_atlas):
        atlas[:t.shape[0], x:x + t.shape[1]] = t
        uv = uv.astype(np.float64).copy()
        uv[:, 0] += x
        out.append(uv)
        x += t.shape[1]
    final = np.vstack(out)
    final[:, 0] /= atlas.shape[1]
    final[:, 1] = 1.0 - final[:, 1] / atlas.shape[0]
    return atlas, final


def test_process_frame_stacks_the_objects_like_the_reference_loop(gpu_ctx):
    """`foreground.process_frame` == the reference's loop over a frame's object ids (pipeline.py:357-468) restated with the per-function entry points: dilated binary
    masks, the 1 % coverage rule, vertices stacked, faces offset by the vertex count, textures packed in one row with normalised uv; an object too small to cover
    1 % of the frame is skipped, and with the coverage constraint disabled it comes back."""
    import torch
    from hive_amd import foreground, geometric, synthetic
    from hive_amd.image_processing import dilate_mask
    from hive_amd.options import MaskDilationOptions, MeshFilteringOptions
    H, W = 240, 320
    seq = synthetic.make_sequence(num_frames=1, height=H, width=W)
    ids = synthetic.ellipse_masks(1, H, W, num_objects=2, seed=3)[0].copy()
    ids[5:12, 5:14] = 3  # a third object of 63 pixels: 0.08 % of the frame
    depth, rgb = seq["depth"][0], seq["color"][0]
    pose = np.linalg.inv(seq["poses"][0])
    R, t = pose[:3, :3], pose[:3, 3:4]
    dil, flt = MaskDilationOptions(num_iterations=2), MeshFilteringOptions()
    K = seq["K"]

    def reference(disable_coverage):
        verts, faces, texs, uvs, count, kept = [], [], [], [], 0, []
        for oid in range(1, int(ids.max()) + 1):
            mask = dilate_mask(ids == oid, dil)
            if mask.mean() < 0.01 and not disable_coverage:
                continue
            v = geometric.point_cloud_from_depth(depth, mask, K, R, t)
            if len(v) < 9:
                continue
            f = foreground.grid_faces(depth, mask, flt, ctx=gpu_ctx)
            if len(f) < 1:
                continue
            tex, uv = foreground.get_mesh_texture_and_uv(v, rgb, K, R, t, ctx=gpu_ctx)
            verts.append(v), faces.append(f.astype(np.int64) + count), texs.append(tex), uvs.append(uv), kept.append(oid)
            count += len(v)
        atlas, uv = _reference_pack_textures(texs, uvs)
        return np.vstack(verts), np.vstack(faces), uv, atlas, kept

    for disable in (False, True):
        want_v, want_f, want_uv, want_atlas, want_kept = reference(disable)
        got = foreground.process_frame(torch.from_numpy(rgb).cuda(), torch.from_numpy(depth).cuda(), torch.from_numpy(ids).cuda(), K, pose, dil, flt,
                                       disable_coverage_constraint=disable, ctx=gpu_ctx)
        assert got["objects"] == want_kept == ([1, 2, 3] if disable else [1, 2])
        assert np.array_equal(got["vertices"].cpu().numpy(), want_v) and np.array_equal(got["faces"].cpu().numpy(), want_f)
        assert np.array_equal(got["texture"].cpu().numpy(), want_atlas) and np.array_equal(got["uv"].cpu().numpy(), want_uv)
        assert 0.0 <= float(got["uv"].min()) and float(got["uv"].max()) <= 1.0
    empty = foreground.process_frame(rgb, depth, np.zeros_like(ids), K, pose, dil, flt, ctx=gpu_ctx)
    assert empty is None
