#!/usr/bin/env python3
"""Headline benchmark: frames/sec of the hot path (DPT-Hybrid depth + TSDF integrate) at 640 x 480 into a
512^3 volume (BASELINE.json `metric`, configs[1]; configs[2] for --gpus > 1).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one batch of `--batch` synthetic frames, already resident in HBM as uint8, through
preprocess -> DPT-Hybrid (random-init weights of the real architecture, bf16, HIP ViT engine) -> f32 head
tail + uint16-mm hand-off -> TSDF integrate.  N > 1: frames are sharded over the ranks (weak scaling:
fixed work per GPU), every rank fuses its shard into its own volume, and ONE all-reduce of the 5 accumulator planes
+ finalize merges the shared static-scene volume inside the timed region.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=24, help="frames per step and GPU (the sequence wraps around; 20-28 measured 3-5 %% above 16)")
    ap.add_argument("--frames", type=int, default=150, help="length of the synthetic sequence (per GPU)")
    ap.add_argument("--voxel", type=float, default=0.01, help="0.01 -> 512^3 over the 5.12 m volume")
    ap.add_argument("--engine", default="hip", choices=["hip", "torch"], help="'torch' = PyTorch-op ViT blocks (comparison only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(seq, voxel, K):
    """The CPU path timed on this box's host cores, on a bounded sample of the same workload: the numpy
    port of the integrate step (bit-identical arithmetic to the C oracle) on one 640 x 480 frame into the
    same 512^3 volume, and the fp32 torch-CPU DPT-Hybrid on two frames (after one warm-up)."""
    import oracle
    from hive_amd import synthetic
    from hive_amd.dpt.models import DPTDepthModel
    threads = torch.get_num_threads()
    model = DPTDepthModel(path=None, scale=0.000305, shift=0.1378, invert=True, engine="torch").eval()
    x = torch.from_numpy(seq["color"][:1].astype(np.float32) / 255.0 * 2.0 - 1.0).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        model(x)
        t0 = time.time()
        for _ in range(2):
            depth = model(x)
        t_dpt = (time.time() - t0) / 2
    depth_np = depth[0].numpy().astype(np.float32)
    ora = oracle.TSDFVolume(synthetic.room_bounds(), voxel)
    tsdf, weight, color = ora._tsdf, ora._weight, ora._color
    t0 = time.time()
    n_upd = oracle.integrate_numpy(tsdf, weight, color, ora._vol_origin, ora._voxel_size, np.float32(ora._trunc_margin), seq["color"][0],
                                   depth_np, K, seq["poses"][0])
    t_tsdf = time.time() - t0
    return {"value": 1.0 / (t_dpt + t_tsdf), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"2 frames DPT-Hybrid fp32 torch-CPU ({threads} threads, {t_dpt:.2f} s/frame) + 1 frame numpy TSDF integrate "
                      f"into {'x'.join(str(int(d)) for d in ora._vol_dim)} (1 thread, {t_tsdf:.2f} s/frame, N_upd {n_upd})"}


def main():
    args = parse_args()
    from hive_amd import _lib, depth as depth_mod, distributed as hdist, fusion, synthetic

    rank, world, local_rank = hdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback); the CPU baseline is only the comparison leg")
    dev_index = local_rank % torch.cuda.device_count()  # one GPU per rank on a node; wraps only in 1-GPU rehearsals
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    torch.backends.cudnn.benchmark = True

    H, W = 480, 640
    B = args.batch
    # synthetic sequence: every rank owns `frames` frames of the 150-pose room trajectory (its shard of
    # world * frames), generated from the seed
    T = args.frames
    seq = synthetic.make_sequence(num_frames=T, height=H, width=W, seed=1234 + rank, yaw_step_deg=360.0 / T)
    K = seq["K"]
    frames_dev = torch.from_numpy(seq["color"]).to(device)  # uint8 [T, H, W, 3] resident in HBM
    poses = seq["poses"]

    ctx = _lib.default_context(dev_index)
    model = depth_mod.build_model(None, device=device, dtype=torch.bfloat16, engine=args.engine)
    volume = fusion.TSDFVolume(synthetic.room_bounds(), args.voxel, ctx=ctx)
    stream = depth_mod.DepthFusionStream(model, volume, K)  # every rank fuses its shard with the same kernels as one GPU

    def batch_indices(step):
        return [(step * B + j) % T for j in range(B)]

    def run_step(step):
        idx = batch_indices(step)
        if idx[-1] == idx[0] + B - 1:
            fr = frames_dev[idx[0]:idx[0] + B]
        else:
            fr = frames_dev[torch.tensor(idx, device=device)]
        return stream.step(fr, poses[idx])

    for s in range(args.warmup):
        run_step(s)
    if world > 1 and args.warmup > 0:
        hdist.fuse_sharded(volume)  # warm-up of the collective too (RCCL sets up its channels on the first large all-reduce)
    torch.cuda.synchronize()
    # reset the volume so that the timed job starts from an empty scene
    volume.reset()
    ctx.set_timing(True)
    hdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        run_step(args.warmup + s)
    if world > 1:
        hdist.fuse_sharded(volume)  # volumes -> sums, one all-reduce of the 5 planes, finalize
    torch.cuda.synchronize()
    hdist.barrier()
    elapsed = time.perf_counter() - t0
    n_launch, kernel_ms = ctx.kernel_time_total()
    ctx.set_timing(False)
    elapsed = hdist.max_over_ranks(elapsed, device=device if world > 1 else "cpu")

    # untimed: N_upd of the timed frames (depends on depth + pose only, not on the volume state)
    n_upd = []
    with torch.no_grad():
        for s in range(min(args.steps, 4)):
            idx = batch_indices(args.warmup + s)
            depth_m, _ = stream.depth(frames_dev[torch.tensor(idx, device=device)])
            for j, i in enumerate(idx):
                n_upd.append(volume.integrate(frames_dev[i], depth_m[j], K, poses[i], return_n_updated=True))
    n_upd_mean = float(np.mean(n_upd))
    bytes_per_voxel = 24  # 3 volumes read + written
    alg_bytes = bytes_per_voxel * n_upd_mean + 8.0 * H * W
    avg_kernel_s = kernel_ms / max(n_launch, 1) * 1e-3
    achieved = alg_bytes / avg_kernel_s / 1e9
    traffic = None
    traffic_file = os.path.join(ROOT, "profiles", "integrate_traffic.json")
    if os.path.exists(traffic_file):
        try:
            traffic = json.load(open(traffic_file)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    if rank != 0:
        return
    total_frames = args.steps * B * world
    dims = "x".join(str(int(d)) for d in volume.vol_dim)
    out = {
        "metric": "frames/sec (depth+TSDF integrate) @640x480, 512^3 vol",
        "value": total_frames / elapsed,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {
            "workload": f"synthetic {W}x{H}x{T} RGB (seeded room trajectory), DPT-Hybrid depth (random-init weights, bf16, "
                        f"{args.engine} ViT engine) + {dims} TSDF integrate, {B} frames/step/GPU"
                        + (", frame-sharded, one RCCL all-reduce of the 5 accumulator planes" if world > 1 else ""),
            "frames_per_step_per_gpu": B, "image": [H, W], "volume": dims, "voxel_m": args.voxel,
            "n_upd_mean": n_upd_mean, "n_upd_fraction": n_upd_mean / volume.num_voxels,
        },
        "roofline": {
            "kernel": "integrate_kernel", "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
            "frac": achieved / 8000.0, "traffic": traffic,
            "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_us": avg_kernel_s * 1e6, "launches": n_launch,
        },
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(seq, args.voxel, K)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
