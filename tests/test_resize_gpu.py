"""Frames that are not the network's size (BASELINE configs[3]: 1920 x 1080 -> 864 x 480) on the MI355X: the reference's two resampling steps
(/root/reference/hive/dataset_adaptors.py:1376-1389 cv2.INTER_CUBIC in, :1421-1426 nearest out) as HIP kernels behind the C ABI
(`hive_dpt_resize_preprocess`, `hive_depth_resize_nearest`, `hive_dpt_forward_frames`) against the CPU oracle's restatement.

Stated tolerances: the kernel evaluates the cubic in float32 where the restatement (like cv2 on a float64 image) sums in float64 -- the float32 result
agrees to 2e-6 absolute on the normalised [-1, 1] scale (1 / 2000 of a uint8 step), and the 16-bit network input is the correctly rounded value or its
neighbour (<= 1 unit in the last place of the network's type, measured in ordinal distance) wherever such a unit exceeds those 2e-6 (|x| >= 2^-7); nearer
zero the bound is the absolute one.  The nearest resize and the hand-off are exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ordinal(t):
    """16-bit floats as integers that count representable values in order (sign-magnitude -> two's complement)."""
    bits = t.view(torch.int16).to(torch.int32)
    return torch.where(bits < 0, -(bits & 0x7FFF), bits)


@pytest.mark.parametrize("shape", [((1080, 1920), (480, 864)), ((240, 320), (480, 640)), ((37, 53), (64, 96)), ((480, 640), (480, 640)), ((720, 1280), (480, 864))])
def test_resize_preprocess_matches_the_cubic_restatement(gpu_ctx, shape):
    import oracle
    from hive_amd import depth as depth_mod
    (H, W), (nh, nw) = shape
    rng = np.random.default_rng(H * 7 + W)
    frames = rng.integers(0, 256, (2, H, W, 3), dtype=np.uint8)
    frames[1, : H // 2] = (np.linspace(0, 255, W)[None, :, None] * np.ones((H // 2, 1, 3))).astype(np.uint8)  # a smooth ramp next to the noise
    want = oracle.dpt_resize_preprocess(frames, nh, nw)  # float32 [B, nh, nw, 3]
    dev = torch.from_numpy(frames).cuda()
    got32 = depth_mod.resize_preprocess_on_device(dev, (nh, nw), torch.float32).permute(0, 2, 3, 1).cpu().numpy()
    assert got32.shape == want.shape
    assert np.abs(got32 - want).max() <= 2e-6, np.abs(got32 - want).max()
    for dtype in (torch.float16, torch.bfloat16):
        got = depth_mod.resize_preprocess_on_device(dev, (nh, nw), dtype)
        assert got.shape == (2, 3, nh, nw) and got.is_contiguous(memory_format=torch.channels_last)
        ref = torch.from_numpy(want).to(dtype)
        got_nhwc = got.permute(0, 2, 3, 1).cpu().contiguous()
        dist = (_ordinal(got_nhwc) - _ordinal(ref)).abs()
        big = torch.from_numpy(np.abs(want) >= 2.0 ** -7)  # where one unit in the last place is well above the 2e-6 of the float32 sums (fp16: 7.6e-6 at 2^-7)
        assert int(dist[big].max()) <= 1, int(dist[big].max())
        assert float((dist[big] > 0).float().mean()) < 0.01  # (a neighbour only where the float32 / float64 sums straddle a rounding boundary)
        # near zero the same 2e-6 spans several (tiny) units: the bound there is absolute -- the rounding of values that agree to 2e-6
        assert float((got_nhwc.float() - ref.float()).abs()[~big].max()) <= 2e-6 + 2.0 ** -7 * 2.0 ** (-10 if dtype == torch.float16 else -7)
    if (H, W) == (nh, nw):  # the identity resize: the taps are exactly (0, 1, 0, 0) -- the table-driven kernel's values (its table is made in float64: a neighbour at most)
        same = depth_mod.preprocess_on_device(dev, torch.float16)
        assert int((_ordinal(same.contiguous()) - _ordinal(depth_mod.resize_preprocess_on_device(dev, (nh, nw), torch.float16).contiguous())).abs().max()) <= 1


def test_nearest_resize_and_hand_off_match_torch_and_the_oracle(gpu_ctx):
    import oracle
    from hive_amd import depth as depth_mod
    rng = np.random.default_rng(3)
    depth = (rng.random((3, 48, 86)) * 12.0).astype(np.float32)  # some beyond max_depth
    # (stays below 65.535 m: beyond it `astype(np.uint16)` of the reference is implementation-defined; the oracle and the head's tail agree the range is DPT's <= 7.3 m)
    for (H, W) in ((108, 192), (1080, 1920), (48, 86), (31, 57)):
        out, mm, m = depth_mod.resize_depth_nearest(torch.from_numpy(depth).cuda(), (H, W), max_depth=10.0)
        want = oracle.resize_nearest(depth, H, W)
        assert np.array_equal(out.cpu().numpy(), want)
        assert np.array_equal(out.cpu().numpy(), torch.nn.functional.interpolate(torch.from_numpy(depth)[:, None], size=(H, W), mode="nearest")[:, 0].numpy())
        for b in range(3):
            o_mm, o_m = oracle.depth_quantize(want[b], 1.0 / 1000.0, 10.0)
            assert np.array_equal(mm[b].cpu().numpy().view(np.uint16), o_mm) and np.array_equal(m[b].cpu().numpy(), o_m)


@pytest.mark.parametrize("backbone", ["vitb_rn50_384", "vitl16_384"])
def test_network_object_on_frames_of_another_size(gpu_ctx, backbone):
    """`hive_dpt_forward_frames` with net != frame size == resize kernel -> the network object at the network's size -> nearest kernel, bit for bit;
    and DepthFusionStream takes such frames (the network size from the reference's rule unless given)."""
    from dpt_weights import seeded_init
    from hive_amd import depth as depth_mod, fusion, synthetic
    from hive_amd.dpt.models import DPTDepthModel
    model = DPTDepthModel(path=None, scale=depth_mod.DPT_SCALE, shift=depth_mod.DPT_SHIFT, invert=True, backbone=backbone, engine="hip").eval()
    seeded_init(model, seed=5)
    model = model.to(memory_format=torch.channels_last).to(torch.float16).cuda()
    seq = synthetic.make_sequence(num_frames=3, height=135, width=240, yaw_step_deg=10.0)
    frames = torch.from_numpy(seq["color"]).cuda()
    net = (96, 128)
    with torch.no_grad():
        depth, mm, m = model.forward_frames(frames, max_depth=10.0, net_size=net)
        assert depth.shape == (3, 135, 240) and mm.shape == (3, 135, 240) and m.shape == (3, 135, 240)
        x = depth_mod.resize_preprocess_on_device(frames, net, torch.float16)
        small = x.permute(0, 2, 3, 1)  # the network object on the resized frames: feed it as uint8-free input through the Python orchestration
        d_net = model(x)
        want, want_mm, want_m = depth_mod.resize_depth_nearest(d_net, (135, 240), max_depth=10.0)
    assert torch.equal(depth, want) and torch.equal(mm, want_mm) and torch.equal(m, want_m)
    assert float(depth.max() - depth.min()) > 0.5 and small.shape == (3, 96, 128, 3)
    ctx = gpu_ctx
    vol = fusion.TSDFVolume(synthetic.room_bounds(), 0.16, ctx=ctx)
    stream = depth_mod.DepthFusionStream(model, vol, seq["K"], net_size=net)
    got = stream.step(frames, seq["poses"])
    assert torch.equal(got, want_m)
    assert vol.stats()[0] == 3
    import oracle
    ora = oracle.TSDFVolume(synthetic.room_bounds(), 0.16)
    d_np = got.cpu().numpy()
    for i in range(3):
        ora.integrate(seq["color"][i], d_np[i], seq["K"], seq["poses"][i])
    assert np.array_equal(vol.get_volume()[0], ora._tsdf)
