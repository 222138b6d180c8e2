#!/usr/bin/env python3
"""The body of tests/test_distributed_gpu.py::test_rccl_merge_of_a_volume_on_the_overlap_side_stream with progress lines."""
import faulthandler
import os
import sys

faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from hive_amd import _lib, depth as depth_mod, distributed as hdist, fusion, synthetic  # noqa: E402

say = lambda *a: print(*a, flush=True)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29534")
torch.cuda.set_device(0)
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for cyc in range(cycles):  # (the test file creates and destroys the group per test)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    say("group", cyc)
    gpu_ctx = _lib.default_context(0)
    seq = synthetic.make_sequence(num_frames=8, height=120, width=160, yaw_step_deg=4.0)
    color_d, depth_d = torch.from_numpy(seq["color"]).cuda(), torch.from_numpy(seq["depth"]).cuda()
    bounds = synthetic.room_bounds()
    ref = fusion.TSDFVolume(bounds, 0.04, ctx=gpu_ctx)
    ref.integrate_batch(color_d, depth_d, seq["K"], seq["poses"])
    want = [t.clone() for t in ref.device_tensors()]
    say("reference volume")
    vctx = depth_mod.DepthFusionStream.side_stream_context(0)
    side = vctx.torch_stream()
    vol = fusion.TSDFVolume(bounds, 0.04, ctx=vctx)
    main = torch.cuda.current_stream()
    for rep in range(3):
        vol.reset()
        say(" reset", rep)
        scaled = depth_d * 1.0
        side.wait_stream(main)
        with torch.cuda.stream(side):
            vol.integrate_batch(color_d, scaled, seq["K"], seq["poses"])
        say(" sweeps queued")
        scaled.record_stream(side)
        say(" record_stream")
        hdist.fuse_sharded(vol)
        say(" merge issued")
        got = vol.device_tensors()
        say(" device_tensors")
        torch.cuda.synchronize()
        ok = torch.equal(got[1], want[1]) and torch.equal(got[2], want[2]) and float((got[0] - want[0]).abs().max()) <= 1e-6
        say(" rep", rep, "ok", ok)
    del vol, vctx, side
    say("context dropped")
    dist.destroy_process_group()
    say("group destroyed")
say("done")
