#!/usr/bin/env python3
"""Headline benchmark: frames/sec of the hot path (DPT-Hybrid depth + TSDF integrate) at 640 x 480 into a
512^3 volume (BASELINE.json `metric`; configs[1] at N = 1, configs[2] at N > 1).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

The job: K * B frames of the seeded 150-frame synthetic sequence (wrapping), B = --batch (96).  A step = one batch of B
frames: uint8 frames start in PINNED HOST memory and are uploaded inside the timed region (double-buffered on a copy stream,
SURVEY.md 8d) -> preprocess -> DPT-Hybrid (random-init weights of the real architecture, bf16, HIP engine) -> f32 depth tail
+ uint16-mm hand-off -> TSDF integrate.

N > 1 (BASELINE configs[2]: the SAME sequence frame-sharded; `--scaling strong`, default): the K * B frames are split in
contiguous blocks over the ranks, identical per-frame work to N = 1, then the shared static-scene volume is merged INSIDE the
timed region:
  --merge sum   (default; north_star's design): every rank fuses its block into its own volume; reduce-scatter of the 5
                accumulator planes -> every rank folds its 1 / N of the voxels -> all-gather of the 3 result planes.
  --merge exact (bit-identical to one GPU): depth + colour frames are all-gathered, every rank integrates all frames in
                sequence order into its x-slab of the volume, the slabs are all-gathered.
`--scaling weak` keeps the round-1 mode (every rank runs K full steps on its own sequence, one merge at the end).
Rank 0 prints one JSON line.
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
    ap.add_argument("--batch", type=int, default=96, help="frames per step (final code: 64 -> 1074, 96 -> 1104, 128 -> 1102 frames/s on one box)")
    ap.add_argument("--frames", type=int, default=150, help="length of the synthetic sequence")
    ap.add_argument("--voxel", type=float, default=0.01, help="0.01 -> 512^3 over the 5.12 m volume")
    ap.add_argument("--engine", default="hip", choices=["hip", "torch"], help="'torch' = PyTorch-op ViT blocks (comparison only)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"], help="N > 1: shard the same job (strong) or repeat it per rank (weak)")
    ap.add_argument("--merge", default="sum", choices=["sum", "exact"], help="N > 1: how the shared volume is merged")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-only", action="store_true", help="profiling runs: skip the untimed roofline measurements behind the timed region")
    return ap.parse_args()


def cpu_baseline(seq, voxel, K):
    """The CPU path timed on this box's host cores, on a bounded sample of the same workload: the numpy
    port of the integrate step (bit-identical arithmetic to the C oracle; numpy runs it on ONE thread) on one 640 x 480
    frame into the same 512^3 volume, and the fp32 torch-CPU DPT-Hybrid on two frames (after one warm-up, all cores)."""
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
                                   depth_np, K, seq["poses"][0], round_mode=ora.round_mode)
    t_tsdf = time.time() - t0
    return {"value": 1.0 / (t_dpt + t_tsdf), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"2 frames DPT-Hybrid fp32 torch-CPU ({threads} threads, {t_dpt:.2f} s/frame) + 1 frame numpy TSDF integrate "
                      f"into {'x'.join(str(int(d)) for d in ora._vol_dim)} (numpy: 1 thread, {t_tsdf:.2f} s/frame, N_upd {n_upd})"}


class FrameFeeder:
    """uint8 frames in pinned host memory -> device, one batch ahead of the compute stream (two device buffers, a copy stream)."""

    def __init__(self, frames_host, batch, device):
        self.host = frames_host  # pinned [T, H, W, 3]
        self.T = frames_host.shape[0]
        self.B = batch
        self.bufs = [torch.empty((batch,) + tuple(frames_host.shape[1:]), dtype=torch.uint8, device=device) for _ in range(2)]
        self.ready = [torch.cuda.Event(), torch.cuda.Event()]  # copy into buffer i has landed
        self.free = [torch.cuda.Event(), torch.cuda.Event()]   # compute that read buffer i is done
        self.copy_stream = torch.cuda.Stream(device=device)
        self.slot = 0
        for e in self.free:
            e.record(torch.cuda.current_stream())

    def prefetch(self, frame_ids):
        i = self.slot
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(self.free[i])
            n = len(frame_ids)
            j = 0
            while j < n:  # one copy per contiguous run of the sequence (a batch wraps around its end at most a few times)
                k = j + 1
                while k < n and frame_ids[k] == frame_ids[k - 1] + 1:
                    k += 1
                self.bufs[i][j:k].copy_(self.host[frame_ids[j]:frame_ids[j] + k - j], non_blocking=True)
                j = k
            self.ready[i].record(self.copy_stream)
        self.slot ^= 1
        return i, n

    def acquire(self, token):
        i, n = token
        torch.cuda.current_stream().wait_event(self.ready[i])
        return self.bufs[i][:n]

    def release(self, token):
        self.free[token[0]].record(torch.cuda.current_stream())


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

    H, W, B, T = 480, 640, args.batch, args.frames
    strong = world > 1 and args.scaling == "strong"
    exact = world > 1 and args.merge == "exact"
    # the synthetic sequence (every rank generates the same one in strong mode; its own in weak mode)
    seq = synthetic.make_sequence(num_frames=T, height=H, width=W, seed=1234 + (0 if strong or world == 1 else rank), yaw_step_deg=360.0 / T)
    K = seq["K"]
    poses = seq["poses"]
    frames_host = torch.from_numpy(seq["color"]).pin_memory()  # uint8 [T, H, W, 3]

    ctx = _lib.default_context(dev_index)
    model = depth_mod.build_model(None, device=device, dtype=torch.bfloat16, engine=args.engine)
    if exact:
        merger = hdist.ExactSlabFusion(synthetic.room_bounds(), args.voxel, ctx=ctx)
        volume = merger.slab
    else:
        merger = None
        volume = fusion.TSDFVolume(synthetic.room_bounds(), args.voxel, ctx=ctx)
    stream = depth_mod.DepthFusionStream(model, volume, K)
    feeder = FrameFeeder(frames_host, B, device)

    # the job: frames job[0 .. K * B) of the wrapping sequence; this rank's contiguous block of it
    def job_frames(first_step, n_steps):
        ids = [(first_step * B + j) % T for j in range(n_steps * B)]
        if strong:
            lo, hi = hdist.shard_range(len(ids), rank, world)
            return ids[lo:hi], [b - a for a, b in (hdist.shard_range(len(ids), r, world) for r in range(world))]
        return ids, [len(ids)] * world

    def batches(ids):
        """At most B frames per batch, in equal parts (60 frames -> 30 + 30, not 48 + 12: small batches fill the chip worse)."""
        n_b = max(1, -(-len(ids) // B))
        cuts = [len(ids) * i // n_b for i in range(n_b + 1)]
        return [ids[a:b] for a, b in zip(cuts[:-1], cuts[1:]) if b > a]

    def run_job(first_step, n_steps, keep_depth=False):
        """All of this rank's batches: upload (one batch ahead) -> depth -> integrate (or, exact mode, keep the depth maps)."""
        ids, counts = job_frames(first_step, n_steps)
        todo = batches(ids)
        kept = []
        token = feeder.prefetch(todo[0]) if todo else None
        for bi, batch_ids in enumerate(todo):
            fr = feeder.acquire(token)
            nxt = feeder.prefetch(todo[bi + 1]) if bi + 1 < len(todo) else None
            if exact:
                depth_m, _ = stream.depth(fr)
                kept.append((fr.clone(), depth_m))
            else:
                depth_m = stream.step(fr, poses[batch_ids])
                if keep_depth:
                    kept.append((batch_ids, depth_m))
            feeder.release(token)
            token = nxt
        if exact:  # all-gather the frames, integrate every frame of the job in sequence order into this rank's x-slab
            color = torch.cat([c for c, _ in kept]) if kept else torch.empty((0, H, W, 3), dtype=torch.uint8, device=device)
            depth = torch.cat([d for _, d in kept]) if kept else torch.empty((0, H, W), dtype=torch.float32, device=device)
            all_ids = [(first_step * B + j) % T for j in range(n_steps * B)]
            merger.integrate(color, depth, K, poses[all_ids], counts)
        return kept

    for s in range(0, args.warmup):
        run_job(s, 1)
    if strong and args.warmup > 0:  # the timed job's own batch sizes (this rank's share of K * B frames) also run once untimed:
        warmed = {len(b) for s in range(args.warmup) for b in batches(job_frames(s, 1)[0])}  # first use sizes the activation arena
        for size in sorted({len(b) for b in batches(job_frames(args.warmup, args.steps)[0])} - warmed):
            ids = list(range(size))
            tok = feeder.prefetch(ids)
            stream.depth(feeder.acquire(tok))
            feeder.release(tok)
    if world > 1 and args.warmup > 0:  # warm-up of the collectives too (RCCL sets up its channels on the first large transfer)
        if exact:
            merger.gather()
        else:
            hdist.fuse_sharded(volume)
    torch.cuda.synchronize()
    volume.reset()  # the timed job starts from an empty scene
    ctx.set_timing(True)
    hdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if strong:
        run_job(args.warmup, args.steps)
    else:
        for s in range(args.steps):
            run_job(args.warmup + s, 1)
    if world > 1:
        merged = merger.gather() if exact else hdist.fuse_sharded(volume)
    torch.cuda.synchronize()
    hdist.barrier()
    elapsed = time.perf_counter() - t0
    n_launch, kernel_ms = ctx.kernel_time_total()
    ctx.set_timing(False)
    elapsed = hdist.max_over_ranks(elapsed, device=device if world > 1 else "cpu")
    if world > 1:
        del merged

    if args.timed_only:
        if rank == 0:
            print(json.dumps({"value": args.steps * B * (1 if strong or world == 1 else world) / elapsed, "unit": "frames/s", "ms_per_step": elapsed / args.steps * 1e3,
                              "avg_integrate_us": kernel_ms / max(n_launch, 1) * 1e3, "note": "--timed-only: no roofline / cpu_baseline legs"}))
        return

    # ---- untimed: what the timed launches processed --------------------------------------------------------------------
    # N_upd of EVERY frame this rank integrated in the timed region (depends on depth + pose only, not on the volume state)
    def measure(frame_sets, depth_of, time_kernel=False):
        """(mean N_upd, ms per frame of the TSDF leg = pack + work list + sweep, [mean sweep-kernel us, launches])."""
        n_upd, leg_ms, k_ms, k_n = [], [], 0.0, 0
        for ids in frame_sets:
            fr = torch.from_numpy(seq["color"][ids]).to(device)
            depth_m = depth_of(fr, ids)
            for j, i in enumerate(ids):  # the counting variant of the kernel (one atomic per wave): never timed
                n_upd.append(volume.integrate(fr[j], depth_m[j], K, poses[i], return_n_updated=True))
            if time_kernel:
                ctx.set_timing(True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            volume.integrate_batch(fr, depth_m, K, poses[ids])  # per frame, back to back
            e1.record()
            e1.synchronize()
            leg_ms.append(e0.elapsed_time(e1) / len(ids))
            if time_kernel:
                n, ms = ctx.kernel_time_total()
                ctx.set_timing(False)
                k_ms, k_n = k_ms + ms, k_n + n
        return float(np.mean(n_upd)), float(np.mean(leg_ms)), (k_ms / max(k_n, 1) * 1e3, k_n)

    def roofline(n_upd_mean, launch_us, traffic_key, frames_per_launch):
        # SURVEY 8(d)'s per-frame figure: 3 volumes read + written for every updated voxel + one read of depth / colour -- times the
        # frames one launch integrates (hive_tsdf_integrate_batch sweeps up to 4 consecutive frames at once: the volume is loaded and
        # stored once for all of them: a quarter of the writes; `traffic` is the PMC figure of the same launch)
        alg = (24.0 * n_upd_mean + 8.0 * H * W) * frames_per_launch
        achieved = alg / (launch_us * 1e-6) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "integrate_traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get(traffic_key)
            except Exception:
                traffic = None
        return {"kernel": "integrate_multi_kernel" if frames_per_launch > 1 else "integrate_kernel", "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                "traffic": traffic, "algorithmic_bytes_per_launch": alg, "frames_per_launch": frames_per_launch, "avg_launch_us": launch_us, "n_upd_mean": n_upd_mean,
                "n_upd_fraction": n_upd_mean / (volume.num_voxels * (world if exact else 1)),
                "note": ("up to 4 consecutive frames per launch share ONE load / store of the volume (bit-identical to one sweep per frame): achieved = "
                         "SURVEY 8(d)'s per-frame bytes x frames_per_launch / avg_launch_us, i.e. useful bytes per second; traffic = 2 x FETCH_SIZE + "
                         "WRITE_SIZE per launch (PMC; the fetches include texel gathers served by the Infinity Cache)") if frames_per_launch > 1 else "one frame per launch"}

    with torch.no_grad():
        if exact:
            timed_sets = batches([(args.warmup * B + j) % T for j in range(args.steps * B)])
        elif strong:
            timed_sets = batches(job_frames(args.warmup, args.steps)[0])
        else:
            timed_sets = [job_frames(args.warmup + s, 1)[0] for s in range(args.steps)]
        n_upd_dpt, leg_dpt, _ = measure(timed_sets, lambda fr, ids: stream.depth(fr)[0])
        frames_timed = sum(len(ids) for ids in timed_sets)  # frames this rank integrated in the timed region
        main_roof = roofline(n_upd_dpt, kernel_ms / max(n_launch, 1) * 1e3, "hbm_bytes_per_launch", frames_timed / max(n_launch, 1))
        main_roof["launches"] = n_launch
        main_roof["tsdf_leg_us_per_frame"] = leg_dpt * 1e3
        # the room scene: the same integrate on the ray-cast (analytic) depth of the sequence -- real depth variation, a
        # surface inside the volume; the DPT-fed scene above has random-weight depth (nearly constant, free space only)
        room_ids = list(range(0, T, max(1, T // 30)))[:30]
        volume.reset()
        n_upd_room, leg_room, (us_room, n_room) = measure([room_ids], lambda fr, ids: torch.from_numpy(seq["depth"][ids]).to(device), time_kernel=True)
        room_roof = roofline(n_upd_room, us_room, "hbm_bytes_per_launch_room", len(room_ids) / max(n_room, 1))
        room_roof["launches"] = n_room
        room_roof["tsdf_leg_us_per_frame"] = leg_room * 1e3
        room_roof["scene"] = "analytic ray-cast depth of the same trajectory (surface inside the volume), 30 frames"

    if rank != 0:
        return
    total_frames = args.steps * B * (1 if strong or world == 1 else world)
    dims = "x".join(str(int(d)) for d in (merger.dims if exact else volume.vol_dim))
    merge_note = ""
    if world > 1:
        merge_note = (", frames all-gathered, every rank integrates all frames into its x-slab (bit-identical to 1 GPU), slabs all-gathered"
                      if exact else ", frame-sharded, shared volume merged by reduce-scatter of the 5 accumulator planes + all-gather of the 3 volumes (RCCL)")
    out = {
        "metric": "frames/sec (depth+TSDF integrate) @640x480, 512^3 vol",
        "value": total_frames / elapsed,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {
            "workload": f"synthetic {W}x{H}x{T} RGB (seeded room trajectory), host-resident uint8 frames uploaded in the timed region, DPT-Hybrid depth "
                        f"(random-init weights, bf16, {args.engine} engine) + {dims} TSDF integrate, {B} frames/step" + merge_note,
            "frames_per_step": B, "frames_total": total_frames, "image": [H, W], "volume": dims, "voxel_m": args.voxel,
            "n_upd_mean": n_upd_dpt, "n_upd_fraction": main_roof["n_upd_fraction"], "merge": (args.merge if world > 1 else None),
        },
        "roofline": main_roof,
        "roofline_room": room_roof,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(seq, args.voxel, K)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
