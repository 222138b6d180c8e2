#!/usr/bin/env python3
"""Where does a single-rank RCCL merge of a side-stream volume crash?  Progress lines, flushed."""
import faulthandler
import os
import sys

faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from hive_amd import _lib, depth as depth_mod, distributed as hdist, fusion, synthetic  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
say = lambda *a: print(*a, flush=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "own_low"
say("group up; mode", mode)
a = torch.ones(1 << 20, device="cuda")
b = torch.empty(1 << 20, device="cuda")
dist.reduce_scatter_tensor(b, a)
torch.cuda.synchronize()
say("collective on the default stream ok")
if mode == "torch_stream":
    side = torch.cuda.Stream()
else:
    vctx = _lib.Context(0, stream=mode)
    side = vctx.torch_stream()
say("side stream", side)
with torch.cuda.stream(side):
    c = torch.ones(1 << 20, device="cuda")
    d = torch.empty(1 << 20, device="cuda")
    say("tensors on the side stream")
    dist.reduce_scatter_tensor(d, c)
    say("collective issued on the side stream")
torch.cuda.synchronize()
say("collective on the side stream ok")
if mode != "torch_stream":
    seq = synthetic.make_sequence(num_frames=4, height=120, width=160, yaw_step_deg=4.0)
    color_d, depth_d = torch.from_numpy(seq["color"]).cuda(), torch.from_numpy(seq["depth"]).cuda()
    vol = fusion.TSDFVolume(synthetic.room_bounds(), 0.04, ctx=vctx)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        vol.integrate_batch(color_d, depth_d, seq["K"], seq["poses"])
    say("sweeps queued")
    hdist.fuse_sharded(vol)
    say("merge issued")
    torch.cuda.synchronize()
    say("merge ok", vol.stats())
dist.destroy_process_group()
say("done")
