"""CPU oracle for the HIVE depth->TSDF hot path: ctypes bindings over ``hive_oracle.c`` plus a
vectorised numpy restatement of the integrate step.

TEST INFRASTRUCTURE ONLY.  Importable from ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; never from ``hive_amd/``.  See the header of
``hive_oracle.c`` for what is pinned by golden vectors (hive.geometric) and what is
**parity unpinned** (everything restated from the absent third_party/tsdf_fusion_python).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libhive_oracle.so")
_lib = None

ROUND_HALF_EVEN = 0
ROUND_HALF_AWAY = 1


def build(force=False):
    """Compile hive_oracle.c with gcc (seconds)."""
    src = os.path.join(_HERE, "hive_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_tsdf_integrate.restype = ctypes.c_uint64
        _lib.oracle_tsdf_accum_integrate.restype = ctypes.c_uint64
        _lib.oracle_unproject.restype = ctypes.c_int64
        _lib.oracle_marching_cubes.restype = ctypes.c_int
    return _lib


def set_threads(n=0):
    """Threads of the integrate loops (OpenMP over the x planes of the volume; the results do not depend on it: voxels are
    independent).  ``n = 0`` keeps the OpenMP default (OMP_NUM_THREADS, else all cores).  Returns the count in effect."""
    return int(lib().oracle_set_threads(int(n)))


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def tsdf_dims(vol_bnds, voxel_size):
    b = _c(vol_bnds, np.float64).reshape(6)
    out = np.zeros(3, np.int64)
    lib().oracle_tsdf_dims(_p(b), ctypes.c_double(voxel_size), _p(out))
    return out


def view_frustum(depth_im, cam_intr, cam_pose):
    d = _c(depth_im, np.float32)
    K = _c(cam_intr, np.float32).reshape(9)
    P = _c(cam_pose, np.float64).reshape(16)
    out = np.zeros(15, np.float64)
    lib().oracle_view_frustum(_p(d), d.shape[0], d.shape[1], _p(K), _p(P), _p(out))
    return out.reshape(3, 5)


def unproject(depth, mask, Kinv, R, t, rgb=None):
    d = _c(depth, np.float32)
    H, W = d.shape
    m = None if mask is None else _c(np.asarray(mask) != 0, np.uint8)
    c = None if rgb is None else _c(rgb, np.uint8)
    out = np.zeros((H * W, 3), np.float64)
    rgba = None if rgb is None else np.zeros((H * W, 4), np.uint8)
    n = lib().oracle_unproject(_p(d), _p(m), _p(c), H, W, _p(_c(Kinv, np.float64).reshape(9)),
                               _p(_c(R, np.float64).reshape(9)), _p(_c(t, np.float64).reshape(3)), _p(out), _p(rgba))
    return (out[:n].copy(), None if rgba is None else rgba[:n].copy())


def project(points, K, R, t, scale_factor=1.0, integer=True, round_mode=ROUND_HALF_EVEN):
    pts = _c(points, np.float64)
    n = pts.shape[0]
    uv_i = np.zeros((n, 2), np.int32) if integer else None
    uv_f = None if integer else np.zeros((n, 2), np.float64)
    depth = np.zeros(n, np.float64)
    lib().oracle_project(_p(pts), ctypes.c_int64(n), _p(_c(K, np.float64).reshape(9)), _p(_c(R, np.float64).reshape(9)),
                         _p(_c(t, np.float64).reshape(3)), ctypes.c_double(scale_factor), round_mode, _p(uv_i), _p(uv_f),
                         _p(depth))
    return (uv_i if integer else uv_f), depth


def dilate_mask(mask, iterations):
    m = _c(np.asarray(mask) != 0, np.uint8)
    out = np.zeros_like(m)
    lib().oracle_dilate_mask(_p(m), m.shape[0], m.shape[1], int(iterations), _p(out))
    return out.astype(bool)


def dilate_mask_se(mask, se, iterations):
    """Literal iterated cv2-style dilation with an arbitrary structuring element (centre anchor, outside ignored)."""
    m = _c(np.asarray(mask) != 0, np.uint8)
    se = _c(np.asarray(se) != 0, np.uint8)
    out = np.zeros_like(m)
    lib().oracle_dilate_mask_se(_p(m), m.shape[0], m.shape[1], _p(se), se.shape[0], se.shape[1], int(iterations), _p(out))
    return out.astype(bool)


def depth_quantize(depth_m, depth_scale=1.0 / 1000.0, max_depth=10.0, mask=None):
    d = _c(depth_m, np.float32)
    mm = np.zeros(d.shape, np.uint16)
    m = np.zeros(d.shape, np.float32)
    mk = None if mask is None else _c(np.asarray(mask) != 0, np.uint8)
    lib().oracle_depth_quantize(_p(d), ctypes.c_int64(d.size), ctypes.c_float(depth_scale), ctypes.c_float(max_depth),
                                _p(mk), _p(mm), _p(m))
    return mm, m


def _cubic_taps_cv2(dst, src):
    """Tap indices [dst, 4] and float32 weights [dst, 4] of cv2.resize(..., interpolation=cv2.INTER_CUBIC) along one axis (published algorithm,
    modules/imgproc/src/resize.cpp: `fx = (float)((dx + 0.5) * scale - 0.5); sx = cvFloor(fx); fx -= sx; interpolateCubic(fx, cbuf)` with
    scale = 1 / (dst / src) in double and A = -0.75; taps sx - 1 .. sx + 2 clamped to the image).  PARITY UNPINNED against cv2 itself (not in this image,
    no fixture in the reference); pinned against torch's bicubic by tests/test_oracle_cpu.py."""
    scale = 1.0 / (np.float64(dst) / np.float64(src))
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    A = np.float32(-0.75)
    one = np.float32(1.0)
    w = np.empty((dst, 4), np.float32)
    w[:, 0] = ((A * (f + one) - np.float32(5.0) * A) * (f + one) + np.float32(8.0) * A) * (f + one) - np.float32(4.0) * A
    w[:, 1] = ((A + np.float32(2.0)) * f - (A + np.float32(3.0))) * f * f + one
    w[:, 2] = ((A + np.float32(2.0)) * (one - f) - (A + np.float32(3.0))) * (one - f) * (one - f) + one
    w[:, 3] = one - w[:, 0] - w[:, 1] - w[:, 2]
    idx = np.clip(s[:, None] + np.arange(-1, 3)[None, :], 0, src - 1)
    return idx, w


def resize_bicubic_cv2(image, out_w, out_h):
    """cv2.resize(image, (out_w, out_h), interpolation=cv2.INTER_CUBIC) for a float64 image [H, W, C] -- what the reference's
    `dpt.transforms.Resize` runs on `image / 255.0` (/root/reference/hive/dataset_adaptors.py:1376-1389, 1407): float32 weights, float64 sums, the
    horizontal pass first, each sum left to right (HResizeCubic / VResizeCubic)."""
    image = np.asarray(image, np.float64)
    H, W = image.shape[:2]
    ix, wx = _cubic_taps_cv2(int(out_w), W)
    iy, wy = _cubic_taps_cv2(int(out_h), H)
    wx, wy = wx.astype(np.float64), wy.astype(np.float64)
    rows = image[:, ix[:, 0]] * wx[None, :, 0, None]
    for k in range(1, 4):
        rows = rows + image[:, ix[:, k]] * wx[None, :, k, None]  # [H, out_w, C]
    out = rows[iy[:, 0]] * wy[:, 0, None, None]
    for k in range(1, 4):
        out = out + rows[iy[:, k]] * wy[:, k, None, None]
    return out


def dpt_resize_preprocess(frames_u8, net_h, net_w):
    """The reference's network input for frames [B, H, W, 3] uint8 that are not the network's size: Resize(cubic) of `image / 255.0`, NormalizeImage(0.5, 0.5),
    PrepareForNet's float32 (dataset_adaptors.py:1376-1392, 1407) -> float32 [B, net_h, net_w, 3] (channels-last; the 16-bit cast is the caller's)."""
    out = [((resize_bicubic_cv2(f / 255.0, net_w, net_h) - 0.5) / 0.5).astype(np.float32) for f in np.asarray(frames_u8)]
    return np.stack(out)


def nearest_index(dst, src):
    """Source indices of torch.nn.functional.interpolate(mode="nearest") along one axis (ATen nearest_neighbor_compute_source_index:
    min((int)floorf(dst_index * scale), src - 1), scale = (float)src / dst) -- dataset_adaptors.py:1421-1426."""
    scale = np.float32(src) / np.float32(dst)
    return np.minimum(np.floor(np.arange(dst, dtype=np.float32) * scale).astype(np.int64), src - 1)


def resize_nearest(depth, out_h, out_w):
    """depth [..., h, w] -> [..., out_h, out_w], torch's nearest rule."""
    depth = np.asarray(depth)
    return depth[..., nearest_index(out_h, depth.shape[-2])[:, None], nearest_index(out_w, depth.shape[-1])[None, :]]


class TSDFVolume:
    """CPU oracle with the call signatures of the reference library's ``fusion.TSDFVolume``
    (call sites /root/reference/hive/fusion.py:104,124,127)."""

    def __init__(self, vol_bnds, voxel_size, round_mode=None, use_gpu=True):
        # use_gpu selects which arithmetic path of the reference library is restated: True (its default) = the
        # CUDA kernel (roundf), False = the numpy path (np.round); an explicit round_mode overrides it
        if round_mode is None:
            round_mode = ROUND_HALF_AWAY if use_gpu else ROUND_HALF_EVEN
        vol_bnds = np.asarray(vol_bnds, dtype=np.float64)
        assert vol_bnds.shape == (3, 2), "[!] `vol_bnds` should be of shape (3, 2)."
        self._voxel_size = float(voxel_size)
        self._trunc_margin = 5 * self._voxel_size
        self._vol_dim = tsdf_dims(vol_bnds, self._voxel_size)
        self._vol_bnds = vol_bnds.copy()
        self._vol_bnds[:, 1] = self._vol_bnds[:, 0] + self._vol_dim * self._voxel_size
        self._vol_origin = self._vol_bnds[:, 0].astype(np.float32)
        self.round_mode = round_mode
        shape = tuple(int(v) for v in self._vol_dim)
        self._tsdf = np.ones(shape, np.float32)
        self._weight = np.zeros(shape, np.float32)
        self._color = np.zeros(shape, np.float32)
        self.last_n_updated = 0

    def integrate(self, color_im, depth_im, cam_intr, cam_pose, obs_weight=1.):
        d = _c(depth_im, np.float32)
        c = _c(color_im, np.uint8)
        H, W = d.shape
        self.last_n_updated = lib().oracle_tsdf_integrate(
            _p(self._tsdf), _p(self._weight), _p(self._color), _p(self._vol_dim), _p(self._vol_origin),
            ctypes.c_float(self._voxel_size), ctypes.c_float(np.float32(self._trunc_margin)), _p(c), _p(d), H, W,
            _p(_c(cam_intr, np.float32).reshape(9)), _p(_c(cam_pose, np.float64).reshape(16)),
            ctypes.c_float(obs_weight), self.round_mode)

    def get_volume(self):
        return self._tsdf, self._color

    def get_mesh(self, return_voxel_coords=False):
        nv = ctypes.c_int64(0)
        nf = ctypes.c_int64(0)
        args = (_p(self._tsdf), _p(self._color), _p(self._vol_dim), _p(self._vol_origin),
                ctypes.c_float(self._voxel_size), ctypes.byref(nv), ctypes.byref(nf))
        lib().oracle_marching_cubes(*args, None, None, None, None, None)
        if nv.value == 0:
            raise ValueError("Surface level must be within volume data range.")
        verts = np.zeros((nv.value, 3), np.float32)
        faces = np.zeros((nf.value, 3), np.int32)
        norms = np.zeros((nv.value, 3), np.float32)
        colors = np.zeros((nv.value, 3), np.uint8)
        vvox = np.zeros((nv.value, 3), np.float32)
        lib().oracle_marching_cubes(*args, _p(verts), _p(faces), _p(norms), _p(colors), _p(vvox))
        if return_voxel_coords:
            return verts, faces, norms, colors, vvox
        return verts, faces, norms, colors


class AccumVolume:
    """Oracle for the frame-sharded accumulate -> (sum over ranks) -> finalize path."""

    def __init__(self, vol_bnds, voxel_size, round_mode=None, use_gpu=True):
        self.vol = TSDFVolume(vol_bnds, voxel_size, round_mode, use_gpu)
        self.accum = np.zeros((5,) + self.vol._tsdf.shape, np.float32)

    def integrate(self, color_im, depth_im, cam_intr, cam_pose, obs_weight=1.):
        v = self.vol
        d = _c(depth_im, np.float32)
        c = _c(color_im, np.uint8)
        return lib().oracle_tsdf_accum_integrate(
            _p(self.accum), _p(v._vol_dim), _p(v._vol_origin), ctypes.c_float(v._voxel_size),
            ctypes.c_float(np.float32(v._trunc_margin)), _p(c), _p(d), d.shape[0], d.shape[1],
            _p(_c(cam_intr, np.float32).reshape(9)), _p(_c(cam_pose, np.float64).reshape(16)),
            ctypes.c_float(obs_weight), v.round_mode)

    @staticmethod
    def planes_from_volume(vol):
        """[tsdf * w, w, r * w, g * w, b * w] of a running-average oracle volume, float32 products (what
        ``hive_tsdf_accum_from_volume`` computes): a rank's contribution when it fused its frames the ordinary way."""
        w = vol._weight.astype(np.float32)
        c = vol._color.astype(np.uint32)
        return np.stack([vol._tsdf * w, w, (c & 255).astype(np.float32) * w, ((c >> 8) & 255).astype(np.float32) * w,
                         (c >> 16).astype(np.float32) * w]).astype(np.float32)

    def finalize(self, accum=None):
        v = self.vol
        a = _c(self.accum if accum is None else accum, np.float32)
        lib().oracle_tsdf_accum_finalize(_p(a), ctypes.c_int64(v._tsdf.size), _p(v._tsdf), _p(v._weight), _p(v._color),
                                         v.round_mode)
        return v


def integrate_numpy(tsdf, weight, color, origin, voxel_size, trunc_margin, color_im, depth_im, cam_intr, cam_pose,
                    obs_weight=1.0, round_mode=ROUND_HALF_EVEN, chunk_x=16):
    """Vectorised numpy restatement of the integrate step, in the style of the reference library's
    CPU path (whole-volume temporaries, boolean masks, fancy-index scatter) but with the float32
    arithmetic contract of ``oracle_tsdf_integrate`` so the two agree bit for bit.
    Processes ``chunk_x`` x-slabs at a time to bound the temporaries.  In place; returns n_updated.
    This is the ``cpu_baseline`` ("port") that bench.py times on the host cores."""
    f32 = np.float32
    rnd = np.rint if round_mode == ROUND_HALF_EVEN else (lambda a: np.copysign(np.floor(np.abs(a) + f32(0.5)), a))
    X, Y, Z = tsdf.shape
    H, W = depth_im.shape
    P = np.asarray(cam_pose, np.float64).astype(f32)
    K = np.asarray(cam_intr, f32)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    vs, tm, ow = f32(voxel_size), f32(trunc_margin), f32(obs_weight)
    origin = np.asarray(origin, f32)
    cim = color_im.reshape(-1, 3)
    dflat = np.ascontiguousarray(depth_im, f32).reshape(-1)
    ys = (origin[1] + np.arange(Y, dtype=f32) * vs) - P[1, 3]
    zs = (origin[2] + np.arange(Z, dtype=f32) * vs) - P[2, 3]
    n_upd = 0
    for x0 in range(0, X, chunk_x):
        x1 = min(X, x0 + chunk_x)
        xs = (origin[0] + np.arange(x0, x1, dtype=f32) * vs) - P[0, 3]
        tx, ty, tz = xs[:, None, None], ys[None, :, None], zs[None, None, :]
        cam_x = (P[0, 0] * tx + P[1, 0] * ty) + P[2, 0] * tz
        cam_y = (P[0, 1] * tx + P[1, 1] * ty) + P[2, 1] * tz
        cam_z = (P[0, 2] * tx + P[1, 2] * ty) + P[2, 2] * tz
        with np.errstate(divide="ignore", invalid="ignore"):
            px = rnd(fx * (cam_x / cam_z) + cx)
            py = rnd(fy * (cam_y / cam_z) + cy)
        valid = (cam_z > 0) & (px >= 0) & (px < W) & (py >= 0) & (py < H)
        pix = np.zeros(valid.shape, np.int64)
        pix[valid] = py[valid].astype(np.int64) * W + px[valid].astype(np.int64)
        depth_val = np.zeros(valid.shape, f32)
        depth_val[valid] = dflat[pix[valid]]
        depth_diff = depth_val - cam_z
        upd = valid & (depth_val != 0) & (depth_diff >= -tm)
        if not upd.any():
            continue
        sl = (slice(x0, x1),)
        dist = np.minimum(f32(1.0), depth_diff[upd] / tm)
        w_old = weight[sl][upd]
        w_new = w_old + ow
        t_old = tsdf[sl][upd]
        c_old = color[sl][upd]
        tsdf[sl][upd] = (t_old * w_old + ow * dist) / w_new
        weight[sl][upd] = w_new
        old_b = np.floor(c_old / f32(65536.0))
        old_g = np.floor((c_old - old_b * f32(65536.0)) / f32(256.0))
        old_r = c_old - old_b * f32(65536.0) - old_g * f32(256.0)
        new = cim[pix[upd]].astype(f32)
        b = np.minimum(rnd((old_b * w_old + ow * new[:, 2]) / w_new), f32(255.0))
        g = np.minimum(rnd((old_g * w_old + ow * new[:, 1]) / w_new), f32(255.0))
        r = np.minimum(rnd((old_r * w_old + ow * new[:, 0]) / w_new), f32(255.0))
        color[sl][upd] = b * f32(65536.0) + g * f32(256.0) + r
        n_upd += int(upd.sum())
    return n_upd
