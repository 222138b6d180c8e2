"""CPU-only checks: the oracle against the golden vectors generated from the real reference module,
against its own numpy restatement, and against analytic known answers."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "geometric_48x64.npz"))


FULL_SWEEPS = bool(os.environ.get("HIVE_TEST_FULL"))  # exhaustive parameter sweeps (minutes) instead of the strided ones


def kinv(K):
    return np.linalg.inv(K).astype(np.float64)


# float64 tolerance between BLAS-evaluated numpy expressions and the plain-C evaluation order
F64_TOL = dict(rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("kname", ["k32", "k64"])
def test_unproject_matches_reference(oracle_lib, gold, kname):
    K = gold["K32"] if kname == "k32" else gold["K64"]
    pts, rgba = oracle_lib.unproject(gold["depth"], gold["mask"], kinv(K), gold["R"], gold["t"], rgb=gold["rgb"])
    ref = gold[f"pc_{kname}"]
    assert pts.shape == ref.shape, "compaction count / order (integer work) must be exact"
    np.testing.assert_allclose(pts, ref, **F64_TOL)
    np.testing.assert_allclose(pts, gold[f"pcrgbd_pts_{kname}"], **F64_TOL)
    assert np.array_equal(rgba, gold[f"pcrgbd_col_{kname}"])


def test_unproject_defaults_and_full_mask(oracle_lib, gold):
    pts, _ = oracle_lib.unproject(gold["depth"], gold["mask"], kinv(gold["K32"]), np.eye(3), np.zeros(3))
    np.testing.assert_allclose(pts, gold["pc_identity"], **F64_TOL)
    pts, _ = oracle_lib.unproject(gold["depth"], None, kinv(gold["K32"]), gold["R"], gold["t"])
    np.testing.assert_allclose(pts, gold["pc_allmask"], **F64_TOL)


def test_project_matches_reference(oracle_lib, gold):
    pts = gold["pc_k32"]
    uv, dep = oracle_lib.project(pts, gold["K32"], gold["R2"], gold["t2"])
    assert np.array_equal(uv, gold["w2i_uv_i32"]), "pixel indices must be bit-exact"
    np.testing.assert_allclose(dep, gold["w2i_depth"], **F64_TOL)
    uvf, _ = oracle_lib.project(pts, gold["K32"], gold["R2"], gold["t2"], integer=False)
    np.testing.assert_allclose(uvf, gold["w2i_uv_f64"], rtol=1e-11, atol=1e-10)
    uvs, _ = oracle_lib.project(pts, gold["K32"], gold["R2"], gold["t2"], scale_factor=2.0)
    assert np.array_equal(uvs, gold["w2i_uv_scaled"])


def test_project_half_pixel_ties(oracle_lib, gold):
    uv, _ = oracle_lib.project(gold["tie_pts"], gold["tie_K"], np.eye(3), np.zeros(3))
    assert np.array_equal(uv, gold["tie_uv"]), "np.round is half-to-even"
    uv_away, _ = oracle_lib.project(gold["tie_pts"], gold["tie_K"], np.eye(3), np.zeros(3), round_mode=oracle_lib.ROUND_HALF_AWAY)
    assert not np.array_equal(uv_away, gold["tie_uv"]), "tie inputs must separate the two rounding modes"


def test_full_resolution_checksums(oracle_lib):
    chk = np.load(os.path.join(GOLDEN, "geometric_full_checksums.npz"))
    rng = np.random.default_rng(1)
    H, W = 480, 640
    K = np.array([[580.0, 0, 319.5], [0, 580.0, 239.5], [0, 0, 1]], np.float32)
    depth = rng.uniform(0.5, 5.0, size=(H, W)).astype(np.float32)
    depth[rng.random((H, W)) < 0.10] = 0.0
    mask = rng.random((H, W)) < 0.9
    pc, _ = oracle_lib.unproject(depth, mask, kinv(K), chk["R"], chk["t"])
    assert len(pc) == int(chk["n"])
    np.testing.assert_allclose(pc[chk["idx"]], chk["pc_rows"], **F64_TOL)
    np.testing.assert_allclose(pc.sum(axis=0), chk["pc_sum"], rtol=1e-9)
    np.testing.assert_allclose(np.abs(pc).sum(axis=0), chk["pc_abs_sum"], rtol=1e-9)
    uv, dep = oracle_lib.project(pc, K, chk["R"], chk["t"])
    assert np.array_equal(uv[chk["idx"]], chk["uv_rows"])
    assert np.array_equal(uv.astype(np.int64).sum(axis=0), chk["uv_sum"])
    np.testing.assert_allclose(dep.sum(), chk["dep_sum"], rtol=1e-9)


def test_unproject_project_round_trip(oracle_lib):
    """identity: project(unproject(pixels)) returns the integer pixels."""
    rng = np.random.default_rng(3)
    H, W = 60, 80
    K = np.array([[72.5, 0, 39.5], [0, 72.5, 29.5], [0, 0, 1]], np.float32)
    depth = rng.uniform(0.5, 4.0, (H, W)).astype(np.float32)
    R = np.eye(3)
    t = np.zeros(3)
    pc, _ = oracle_lib.unproject(depth, None, kinv(K), R, t)
    uv, dep = oracle_lib.project(pc, K, R, t)
    v, u = np.mgrid[0:H, 0:W]
    assert np.array_equal(uv[:, 0], u.ravel()) and np.array_equal(uv[:, 1], v.ravel())
    np.testing.assert_allclose(dep, depth.ravel().astype(np.float64), rtol=1e-6)


def test_numpy_restatement_is_bit_exact_with_c(oracle_lib, small_sequence):
    from hive_amd import synthetic
    seq = small_sequence
    for rm in (0, 1):
        vol = oracle_lib.TSDFVolume(synthetic.room_bounds(), 0.08, round_mode=rm)
        t = np.ones_like(vol._tsdf)
        w = np.zeros_like(t)
        c = np.zeros_like(t)
        for i in range(4):
            vol.integrate(seq["color"][i], seq["depth"][i], seq["K"], seq["poses"][i])
            n = oracle_lib.integrate_numpy(t, w, c, vol._vol_origin, vol._voxel_size, np.float32(vol._trunc_margin), seq["color"][i],
                                           seq["depth"][i], seq["K"], seq["poses"][i], round_mode=rm)
            assert n == vol.last_n_updated
        assert np.array_equal(t, vol._tsdf) and np.array_equal(w, vol._weight) and np.array_equal(c, vol._color)


def test_plane_known_answer(oracle_lib):
    """A fronto-parallel wall at depth d seen by an axis-aligned camera: the tsdf of voxel z is
    min(1, (d - z_cam)/trunc), it crosses zero at the analytic voxel, weights count frames."""
    H, W = 48, 64
    K = np.array([[60.0, 0, 31.5], [0, 60.0, 23.5], [0, 0, 1]], np.float32)
    pose = np.eye(4)
    pose[:3, 3] = [0.5, 0.4, -0.5]
    d = 1.2
    depth = np.full((H, W), d, np.float32)
    color = np.full((H, W, 3), (200, 100, 50), np.uint8)
    vol = oracle_lib.TSDFVolume(np.array([[0, 1.0], [0, 0.8], [0, 1.0]]), 0.02)
    for _ in range(3):
        vol.integrate(color, depth, K, pose)
    tsdf, _ = vol.get_volume()
    x, y = 25, 20  # a column through the middle of the frustum
    z_cam = (np.float32(0) + np.arange(50, dtype=np.float32) * np.float32(0.02)) + np.float32(0.5)
    expect = np.minimum(np.float32(1), (np.float32(d) - z_cam) / np.float32(5 * 0.02))
    seen = (np.float32(d) - z_cam) >= -np.float32(5 * 0.02)
    col = tsdf[x, y, :]
    np.testing.assert_allclose(col[seen], expect[seen], rtol=0, atol=1e-6)
    assert (col[~seen] == 1.0).all()
    assert (vol._weight[x, y, seen] == 3).all() and (vol._weight[x, y, ~seen] == 0).all()
    zc = np.where((col[:-1] >= 0) & (col[1:] < 0))[0]
    assert len(zc) == 1 and abs((zc[0] + col[zc[0]] / (col[zc[0]] - col[zc[0] + 1])) * 0.02 - (d - 0.5)) < 1e-4
    # packed colour: b*65536 + g*256 + r
    assert vol._color[x, y, 30] == 50 * 65536 + 100 * 256 + 200
    verts, faces, norms, colors = vol.get_mesh()
    # two sheets, as in the reference library: the wall itself, and the jump back to the unobserved
    # value 1 one truncation distance behind it
    front = verts[:, 2] < d - 0.5 + 0.05
    assert front.sum() * 2 == len(verts)
    np.testing.assert_allclose(verts[front, 2], d - 0.5, atol=1e-4)
    assert (colors[front] == np.array([200, 100, 50], np.uint8)).all()
    np.testing.assert_allclose(norms[front, 2], -1.0, atol=1e-5)  # towards the camera = increasing tsdf


def _sphere_volume(oracle_lib, n=40, r=0.31):
    vol = oracle_lib.TSDFVolume(np.array([[0, 1.0]] * 3), 1.0 / n)
    g = (np.arange(n, dtype=np.float32) + 0) / n
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    sd = np.sqrt((X - 0.5) ** 2 + (Y - 0.47) ** 2 + (Z - 0.52) ** 2) - r
    vol._tsdf[...] = np.clip(sd / vol._trunc_margin, -1, 1).astype(np.float32)
    vol._color[...] = 255.0
    return vol


def test_sphere_mesh_is_closed_manifold(oracle_lib):
    vol = _sphere_volume(oracle_lib)
    verts, faces, norms, colors, vvox = vol.get_mesh(return_voxel_coords=True)
    edges = np.sort(np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]]), axis=1)
    uniq, counts = np.unique(edges, axis=0, return_counts=True)
    assert (counts == 2).all(), "every edge of a closed surface is shared by exactly two triangles"
    assert len(verts) - len(uniq) + len(faces) == 2, "Euler characteristic of a sphere"
    # consistent outward orientation: face normal . (centroid - centre) > 0
    tri = verts[faces]
    fn = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    out = tri.mean(axis=1) - np.array([0.5, 0.47, 0.52], np.float32)
    assert ((fn * out).sum(axis=1) > 0).all()
    assert ((norms * (verts - np.array([0.5, 0.47, 0.52], np.float32))).sum(axis=1) > 0).all()
    np.testing.assert_allclose(np.linalg.norm(verts - np.array([0.5, 0.47, 0.52]), axis=1), 0.31, atol=2e-3)
    # the vertex set is exactly the set of sign-changing grid edges
    t = vol._tsdf
    n_edges = sum(int(((np.take(t, range(0, t.shape[a] - 1), axis=a) < 0) != (np.take(t, range(1, t.shape[a]), axis=a) < 0)).sum())
                  for a in range(3))
    assert len(verts) == n_edges


def test_mc_tables_are_watertight():
    """Every case: triangles use exactly the crossed edges; for every pair of face-adjacent cells the
    segments drawn on the shared face agree (checked by exhaustive enumeration of 12-corner pairs)."""
    import re
    hdr = open(os.path.join(os.path.dirname(GOLDEN), "..", "include", "hive_mc_tables.h")).read()
    rows = re.findall(r"\{\s*((?:\d+,\s*){14}\d+)\}", hdr)
    assert len(rows) == 256
    tri = [[int(v) for v in r.replace(" ", "").split(",")] for r in rows]
    EC = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
    for cs in range(256):
        used = {e for e in tri[cs] if e != 255}
        crossed = {e for e, (a, b) in enumerate(EC) if ((cs >> a) & 1) != ((cs >> b) & 1)}
        assert used == crossed
    # +x face of cell A (corners 1,2,6,5) is the -x face of cell B (corners 0,3,7,4); edges 1,10,5,9 <-> 3,11,7,8
    a_edges, b_edges = [1, 10, 5, 9], [3, 11, 7, 8]
    def face_segments(cs, face_edges):
        segs = set()
        t = tri[cs]
        for k in range(0, 15, 3):
            if t[k] == 255:
                break
            for i in range(3):
                e0, e1 = t[k + i], t[k + (i + 1) % 3]
                if e0 in face_edges and e1 in face_edges:
                    segs.add(frozenset((face_edges.index(e0), face_edges.index(e1))))
        return segs
    for shared in range(16):  # signs of the 4 shared corners
        s = [(shared >> i) & 1 for i in range(4)]
        for rest_a in range(16):
            for rest_b in range(16):
                ra = [(rest_a >> i) & 1 for i in range(4)]
                rb = [(rest_b >> i) & 1 for i in range(4)]
                cs_a = (ra[0] << 0) | (s[0] << 1) | (s[1] << 2) | (ra[1] << 3) | (ra[2] << 4) | (s[3] << 5) | (s[2] << 6) | (ra[3] << 7)
                cs_b = (s[0] << 0) | (rb[0] << 1) | (rb[1] << 2) | (s[1] << 3) | (s[3] << 4) | (rb[2] << 5) | (rb[3] << 6) | (s[2] << 7)
                sa, sb = face_segments(cs_a, a_edges), face_segments(cs_b, b_edges)
                # interior fan diagonals may also join two face edges; the face's own segments are those
                # both cells must share, so compare only when neither cell adds a diagonal on this face
                n_cross = sum(1 for i in range(4) if s[i] != s[(i + 1) % 4])
                if n_cross == 0:
                    assert not sa and not sb
                else:
                    assert sa & sb, (cs_a, cs_b)
                    assert len(sa & sb) >= n_cross // 2


def test_view_frustum_known_answer(oracle_lib):
    depth = np.zeros((480, 640), np.float32)
    depth[100, 200] = 3.0
    K = np.array([[580.0, 0, 319.5], [0, 580.0, 239.5], [0, 0, 1]], np.float32)
    pose = np.eye(4)
    pose[:3, 3] = [1.0, 2.0, 3.0]
    f = oracle_lib.view_frustum(depth, K, pose)
    assert f.shape == (3, 5)
    np.testing.assert_allclose(f[:, 0], [1, 2, 3])
    np.testing.assert_allclose(f[:, 1], [1 + (0 - 319.5) * 3 / 580, 2 + (0 - 239.5) * 3 / 580, 6])
    np.testing.assert_allclose(f[:, 4], [1 + (640 - 319.5) * 3 / 580, 2 + (480 - 239.5) * 3 / 580, 6])


def test_dilate_equals_box_max(oracle_lib):
    rng = np.random.default_rng(5)
    m = rng.random((40, 50)) < 0.01
    for it in (0, 1, 3, 10):
        out = oracle_lib.dilate_mask(m, it)
        ref = np.zeros_like(m)
        for v, u in zip(*np.nonzero(m)):
            ref[max(0, v - it):v + it + 1, max(0, u - it):u + it + 1] = True
        assert np.array_equal(out, ref)


def test_fast_colour_update_identity():
    """The sweep's division-free colour update (hive_amd/csrc/tsdf.hip, update_voxels FASTC) against the contract's expression
    c' = min(255, roundf((c w + c_new) / (w + 1))) evaluated in float32 operation by operation, on the CPU:
      A. roundf(fl(n / d)) == floor((2 n + d) / (2 d)) for integers n <= 255 d, d < 65536 -- every (c, c_new) for a set of weights, and for
         EVERY d the numerators next to each half-integer quotient (where a double rounding would show);
      B. floor(fma(delta, y, 1/2 + y / 4)) == floor((2 delta + d) / (2 d)) for every delta in [-255, 255] and every d in [1, 65534], with
         y = the float32 reciprocal of d and y moved by up to two ulps either way (the kernel's refined reciprocal is within one)."""
    f32, f64 = np.float32, np.float64

    def quotient_rounded(n, d):  # roundf of the correctly rounded float32 quotient (n, d: float32 integers; n / d >= 0)
        q = (n.astype(f64) / d.astype(f64)).astype(f32)  # float64 quotient of float32 operands rounds to the correctly rounded float32 quotient
        return np.minimum(np.floor(q.astype(f64) + 0.5), 255.0)

    def fast(c, cn, d, ulps):
        y = (1.0 / d.astype(f64)).astype(f32)
        for _ in range(abs(ulps)):
            y = np.nextafter(y, f32(np.inf) if ulps > 0 else f32(-np.inf))
        bias = (0.25 * y.astype(f64) + 0.5).astype(f32)                                       # fma(0.25, y, 0.5): exact in float64, one rounding
        t = ((cn - c).astype(f64) * y.astype(f64) + bias.astype(f64)).astype(f32)              # fma(delta, y, bias)
        return c.astype(f64) + np.floor(t.astype(f64))

    d = np.arange(1, 65535, dtype=np.int64)
    for delta in range(-255, 256, 1 if FULL_SWEEPS else 3):
        exact = np.floor_divide(2 * delta + d, 2 * d)
        for ulps in (-2, 0, 2):
            assert np.array_equal(fast(np.zeros(len(d), f32), np.full(len(d), delta, f32), d.astype(f32), ulps), exact), (delta, ulps)
    for k in range(0, 255):  # numerators around (k + 1/2) d, every d
        n_lo = ((2 * k + 1) * d) // 2
        for n in (n_lo - 1, n_lo, n_lo + 1):
            n = np.clip(n, 0, 255 * d)
            assert np.array_equal(quotient_rounded(n.astype(f32), d.astype(f32)), np.floor_divide(2 * n + d, 2 * d)), k
    cc, cn = (a.ravel().astype(f32) for a in np.meshgrid(np.arange(256), np.arange(256), indexing='ij'))
    for w in list(range(0, 40)) + [255, 256, 1000, 4095, 4096, 32767, 32768, 65532, 65533]:
        n = (cc * f32(w)).astype(f32) + cn  # the contract's numerator, exact below 2^24
        ref = quotient_rounded(n.astype(f32), np.full(len(cc), w + 1, f32))
        assert np.array_equal(fast(cc, cn, np.full(len(cc), w + 1, f32), 0), ref), w
        assert np.array_equal(fast(cc, cn, np.full(len(cc), w + 1, f32), 1), ref), w


def test_dilate_with_a_structuring_element(oracle_lib):
    """Known answers of cv2.dilate's definition (anchor at the centre, outside ignored, iterated): a single pixel dilated once is the
    element mirrored about its anchor; the 3x3 box gives the default path; an element without its centre moves the mask."""
    m = np.zeros((9, 11), bool)
    m[4, 5] = True
    cross = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], np.uint8)
    out = oracle_lib.dilate_mask_se(m, cross, 1)
    assert out.sum() == 5 and out[3, 5] and out[5, 5] and out[4, 4] and out[4, 6] and out[4, 5]
    out2 = oracle_lib.dilate_mask_se(m, cross, 2)  # the diamond of radius 2
    vv, uu = np.mgrid[0:9, 0:11]
    assert np.array_equal(out2, np.abs(vv - 4) + np.abs(uu - 5) <= 2)
    corner = np.array([[1, 0, 0], [0, 0, 0], [0, 0, 0]], np.uint8)  # dst(v, u) = src(v - 1, u - 1): the mask moves down-right
    moved = oracle_lib.dilate_mask_se(m, corner, 3)
    assert moved.sum() == 1 and moved[7, 8]
    rng = np.random.default_rng(6)
    r = rng.random((40, 50)) < 0.02
    for it in (0, 1, 4):
        assert np.array_equal(oracle_lib.dilate_mask_se(r, np.ones((3, 3), np.uint8), it), oracle_lib.dilate_mask(r, it))
    assert np.array_equal(oracle_lib.dilate_mask_se(r, cross, 0), r)


def test_depth_quantize(oracle_lib):
    d = np.array([[0.0, 0.0004, 1.23456, 7.2569, 9.9999, 10.0004, 12.0]], np.float32)
    mm, m = oracle_lib.depth_quantize(d)
    assert list(mm[0]) == [0, 0, 1234, 7256, 9999, 10000, 12000]
    np.testing.assert_array_equal(m[0], np.array([0, 0, np.float32(0.001) * np.float32(1234), np.float32(0.001) * np.float32(7256),
                                                  np.float32(0.001) * np.float32(9999), np.float32(0.001) * np.float32(10000), 0],
                                                 np.float32))


def test_cubic_resize_restatement_against_torch_bicubic():
    """`resize_bicubic_cv2` restates cv2.resize(INTER_CUBIC) from its published source (cv2 is not installed here and the reference holds no fixture of it:
    parity with cv2 itself is UNPINNED).  What pins the restatement: torch's bicubic with align_corners=False is an independent implementation of the same
    a = -0.75 kernel with the same index clamping, evaluated in float64; the only stated difference is that cv2 rounds the source coordinate and the four
    weights to float32 -- a coordinate of ~1000 carries 6e-5 of rounding, so the agreement is ~1e-5 when shrinking and exact-ish when the scale is a
    power of two.  Also: torch's nearest rule (exact) and the sizing rule of the reference's Resize for the frame sizes BASELINE names."""
    import torch
    import oracle
    from hive_amd import depth as depth_mod
    rng = np.random.default_rng(0)
    for (H, W, oh, ow), tol in (((108, 192, 48, 86), 5e-5), ((48, 64, 96, 128), 1e-12), ((31, 45, 32, 64), 1e-5), ((270, 480, 96, 160), 1e-4)):
        img = rng.random((H, W, 3))
        mine = oracle.resize_bicubic_cv2(img, ow, oh)
        ref = torch.nn.functional.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None], size=(oh, ow), mode="bicubic", align_corners=False)[0].permute(1, 2, 0).numpy()
        assert mine.shape == (oh, ow, 3) and np.abs(mine - ref).max() <= tol, (H, W, oh, ow, np.abs(mine - ref).max())
    # a constant image stays constant (the weights sum to 1 by construction), overshoot is not clipped (float images)
    assert np.allclose(oracle.resize_bicubic_cv2(np.full((20, 30, 3), 0.25), 13, 7), 0.25, atol=1e-7)
    step = np.zeros((16, 16, 1))
    step[:, 8:] = 1.0
    r = oracle.resize_bicubic_cv2(step, 40, 16)
    assert r.min() < -0.01 and r.max() > 1.01
    d = rng.random((2, 48, 86)).astype(np.float32)
    for size in ((108, 192), (1080, 1920), (48, 86), (31, 57)):
        ref = torch.nn.functional.interpolate(torch.from_numpy(d)[:, None], size=size, mode="nearest")[:, 0].numpy()
        assert np.array_equal(oracle.resize_nearest(d, *size), ref)
    assert depth_mod.network_size(480, 640) == (480, 640) and depth_mod.network_size(1080, 1920) == (480, 864)
    assert depth_mod.network_size(240, 320) == (480, 640) and depth_mod.network_size(720, 1280) == (480, 864) and depth_mod.network_size(1920, 1080) == (1152, 640)
