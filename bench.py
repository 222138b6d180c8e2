#!/usr/bin/env python3
"""Headline benchmark: frames/sec of the hot path (DPT-Hybrid depth + TSDF integrate) at 640 x 480 into a
512^3 volume (BASELINE.json `metric`; configs[1] at N = 1, configs[2] at N > 1).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

The job: K * B frames of the seeded 150-frame synthetic sequence (wrapping), B = --batch (107: fills the GEMMs' tile rounds).  A step = one batch of B
frames: uint8 frames start in PINNED HOST memory and are uploaded inside the timed region (double-buffered on a copy stream,
SURVEY.md 8d) -> preprocess -> DPT-Hybrid (seeded random weights of the real architecture: depth maps of 1-7 m, so the TSDF scene
has surfaces inside the volume; bf16 or --dtype fp16, HIP engine) -> f32 depth tail
+ uint16-mm hand-off -> TSDF integrate.

One run prints ONE JSON line that covers the BASELINE configurations this command can reach:
  N = 1: the headline (configs[1], `value`), the same job in float16 -- the reference's own type -- as `value_fp16` (with `small_batch`: the network object at
         batches of 1 and 8 in that type, the reference's literal one-frame loop), and a bounded `config4` leg
         (configs[3]: 1920 x 1080 frames, DPT-Large at the reference's 864 x 480 network size, 1024^3 volume).
  N > 1: frames are independent units.  The headline `value` is the WEAK job ("scaling": "weak": N * K * B frames, a contiguous block of K
         steps per rank, per-GPU work = the N = 1 job, no collective on the data path, the shared volume merged ONCE inside the timed
         region), and the same run also times BASELINE configs[2] literally under the key `strong`: the SAME K * B frames split in
         contiguous blocks over the ranks, one merge (fixed total work; the merge weighs more the shorter the per-rank share).
         `--scaling strong` swaps the two (headline = strong, the weak job under `weak`).
The merge:
  --merge sum   (default; north_star's design): every rank fuses its block into its own volume; reduce-scatter of the 5
                accumulator planes -> every rank folds its 1 / N of the voxels -> all-gather of the 3 result planes.
  --merge exact (bit-identical to one GPU): depth + colour frames are all-gathered, every rank integrates all frames in
                sequence order into its x-slab of the volume, the slabs are all-gathered.
(`--merge exact` is a strong-scaling mode by construction: every rank integrates every frame.)
Rank 0 prints the line.
"""
import argparse
import hashlib
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
    ap.add_argument("--batch", type=int, default=107,
                    help="frames per step.  107 frames x 1216 padded tokens = 508.25 row tiles of 256: the N = 768 / 1536 / 3072 GEMMs then have 1527 / 3054 / "
                         "6108 tiles = 5.96 / 11.93 / 23.86 rounds of the 256 CUs (96 frames: 5.34 rounds, the sixth a third full).  Same box: 96 -> 1175, "
                         "107 -> 1207, 125 -> 1194, 143 -> 1208 frames/s")
    ap.add_argument("--frames", type=int, default=150, help="length of the synthetic sequence")
    ap.add_argument("--voxel", type=float, default=0.01, help="0.01 -> 512^3 over the 5.12 m volume")
    ap.add_argument("--engine", default="hip", choices=["hip", "torch"], help="'torch' = PyTorch-op ViT blocks (comparison only)")
    ap.add_argument("--scaling", default="weak", choices=["strong", "weak"],
                    help="N > 1: which job is the headline `value` (the other one is timed too and reported under its own key).  weak = N x K x B frames, a "
                         "contiguous block of K steps per rank; strong = BASELINE config 3 literally: the same K x B frames split over the ranks")
    ap.add_argument("--merge", default="sum", choices=["sum", "exact"], help="N > 1: how the shared volume is merged")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"], help="16-bit type of the headline network (north_star: bf16; the reference runs fp16)")
    ap.add_argument("--no-overlap", action="store_true", help="TSDF sweeps on the network's stream (default: on a second stream, under the next batch's network)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the legs beside the headline job (fp16 job, config4, the other scaling mode at N > 1)")
    ap.add_argument("--notes", action="store_true", help="keep the explanatory note strings in the JSON line (default: numbers only -- the notes are DESIGN.md section 6; "
                                                         "the full line is ~10 KB, the compact one ~4 KB, which a log tail does not cut)")
    ap.add_argument("--timed-only", action="store_true", help="profiling runs: only the headline timed job (no roofline / cpu_baseline / extra legs)")
    return ap.parse_args()


def compact(obj, keep_notes=False):
    """The JSON line without its prose: floats to 6 significant digits, the explanatory strings (`note`, `*_note`, `overlap`, `scene`, per-kernel counter tables) dropped
    unless --notes.  Every number stays."""
    drop = ("note", "expected_note", "algorithmic_bytes_note", "traffic_note", "overlap", "traffic_detail")
    if isinstance(obj, dict):
        return {k: compact(v, keep_notes) for k, v in obj.items() if keep_notes or not (k in drop or k.endswith("_note"))}
    if isinstance(obj, (list, tuple)):
        return [compact(v, keep_notes) for v in obj]
    if isinstance(obj, float):
        return float(f"{obj:.6g}")
    return obj


def host_cores():
    """Cores this process may really use: the scheduler affinity, cut down to the cgroup's CPU quota where one is set (a GPU box
    hands a 1-GPU job a share of the host: 256 threads on a 16-core share ran the torch-CPU DPT at 87 s per frame)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    env = os.environ.get("HIVE_BENCH_CPU_THREADS")
    return int(env) if env else min(cores, 64)


def cpu_baseline(seq, voxel, K, budget_s=20.0):
    """The CPU path timed on this box's host cores, on a bounded sample of the same workload (about `budget_s` seconds of CPU work):
    the fp32 torch-CPU DPT-Hybrid on consecutive frames of the sequence (batches of 4, after a warm-up; as many batches as fit half the budget, at
    most 16 frames), then the C oracle's integrate -- the restatement of the reference library's loop, its x planes spread over the same cores with
    OpenMP, as the library's own CPU fallback is `numba @njit(parallel=True)` -- of exactly those frames, each with ITS OWN CPU depth map (after the
    uint16-mm hand-off) and pose, into the same 512^3 volume.  Both legs use every core the process may run on; the count is in `cores`."""
    import oracle
    from hive_amd import synthetic
    from hive_amd.dpt.init import seeded_init
    from hive_amd.dpt.models import DPTDepthModel
    cores = host_cores()
    torch.set_num_threads(cores)
    model = seeded_init(DPTDepthModel(path=None, scale=0.000305, shift=0.1378, invert=True, engine="torch"), seed=1234).eval()
    to_x = lambda ids: torch.from_numpy(seq["color"][ids].astype(np.float32) / 255.0 * 2.0 - 1.0).permute(0, 3, 1, 2).contiguous()
    depths, t_dpt, n_dpt = [], 0.0, 0
    with torch.no_grad():
        model(to_x([0]))  # warm-up
        while n_dpt < 16 and (n_dpt == 0 or t_dpt < budget_s / 2):
            x = to_x(list(range(n_dpt, n_dpt + 4)))
            t0 = time.time()
            d = model(x)
            t_dpt += time.time() - t0
            depths.append(d.numpy().astype(np.float32))
            n_dpt += 4
    depth_np = np.concatenate(depths)
    depth_np = np.where(depth_np > 10.0, 0.0, np.trunc(depth_np * 1000.0) / 1000.0).astype(np.float32)  # the uint16-mm hand-off
    threads = oracle.set_threads(cores)
    ora = oracle.TSDFVolume(synthetic.room_bounds(), voxel)
    n_upd = 0
    t0 = time.time()
    for i in range(n_dpt):
        ora.integrate(seq["color"][i], depth_np[i], K, seq["poses"][i])
        n_upd += ora.last_n_updated
    t_tsdf = time.time() - t0
    per_frame = (t_dpt + t_tsdf) / n_dpt
    return {"value": 1.0 / per_frame, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"frames 0..{n_dpt - 1} of the sequence: DPT-Hybrid fp32 torch-CPU in batches of 4 ({torch.get_num_threads()} threads, {t_dpt / n_dpt:.2f} s/frame) + "
                      f"C-oracle TSDF integrate of each frame's own depth map into {'x'.join(str(int(d)) for d in ora._vol_dim)} (OpenMP, {threads} threads, "
                      f"{t_tsdf / n_dpt:.3f} s/frame, mean N_upd {n_upd // n_dpt}); {t_dpt + t_tsdf:.1f} s of CPU work in all"}


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

    def release(self, token, stream=None):
        self.free[token[0]].record(stream or torch.cuda.current_stream())


def source_stamp():
    """sha256 of the TSDF kernel source: profiles/*_integrate_pmc.json carry the stamp of the source they were measured on, and the counter figures
    are only merged into the line when it matches (a kernel change without a profile refresh must not mix generations)."""
    h = hashlib.sha256()
    for name in ("tsdf.hip",):  # (round 5: the kernel file alone -- hive_internal.hpp holds host-side structs of every translation unit)
        with open(os.path.join(ROOT, "hive_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def load_profile(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def config4_leg(device, ctx, steps=2, batch=48, overlap=True, prefetch=True, unique_frames=48):
    """BASELINE configs[3], bounded, through the PRODUCT path: 1920 x 1080 frames (host-resident, uploaded in the timed region) -> `DepthFusionStream.step`:
    the reference's resize rule (640 x 480 target, keep aspect ratio, "minimal", multiple of 32: 864 x 480 -- hive_amd.depth.network_size) and its cv2.INTER_CUBIC
    resize + normalisation as one HIP kernel -> DPT-Large depth (backbone vitl16_384, bf16, seeded weights) -> nearest back to 1080p + uint16-mm hand-off as
    one HIP kernel (all of it ONE C-ABI call, hive_dpt_forward_frames) -> integrate into a 1024^3 volume (5 mm voxels), the sweeps of a batch on the side
    stream under the next batch's network as in the headline job.  No torch operator in the timed region but the upload.  `steps` timed steps of `batch` frames
    after one warm-up step, wrapping around `unique_frames` distinct frames (ray-casting a 1080p frame on the host takes about a second).  Batches of 48 (round 5: 389 frames/s at
    16, 396 at 32, 409 at 48 -- tools/probe_config4_batch.py; the network's per-frame time still falls with the batch, 1.94 -> 1.81 ms).  The sweep's roofline (SURVEY 8d bytes and must-move bytes) on two scenes, as for the headline: the DPT depth of the last step's
    frames (`roofline`: seeded random weights give a noise-like depth map -- two thirds of the voxels the sweep must test cannot update) and the analytic
    ray-cast depth of the same frames (`roofline_room`: the room's walls)."""
    from hive_amd import depth as depth_mod, fusion, synthetic
    from hive_amd.dpt.init import seeded_init
    from hive_amd.dpt.models import DPTDepthModel
    H, W, T = 1080, 1920, min(unique_frames, batch * (steps + 1))  # (the job wraps around a short sequence, as the headline job does: ray-casting a 1080p frame on the host takes a second)
    frame_ids = lambda i: [(i * batch + j) % T for j in range(batch)]
    net_h, net_w = depth_mod.network_size(H, W)
    seq = synthetic.make_sequence(num_frames=T, height=H, width=W, yaw_step_deg=2.4)
    model = DPTDepthModel(path=None, scale=depth_mod.DPT_SCALE, shift=depth_mod.DPT_SHIFT, invert=True, backbone="vitl16_384", engine="hip")
    seeded_init(model, seed=1234)
    model = model.eval().to(memory_format=torch.channels_last).to(torch.bfloat16).to(device)
    storage = tuple(torch.empty(1024 ** 3, dtype=torch.float32, device=device) for _ in range(3))  # caller-owned planes: the weight plane is read below
    vctx = depth_mod.DepthFusionStream.side_stream_context(device.index or 0) if overlap else ctx
    vol = fusion.TSDFVolume(synthetic.room_bounds(), 0.005, ctx=vctx, storage=storage)
    assert tuple(int(d) for d in vol.vol_dim) == (1024, 1024, 1024)
    host = torch.from_numpy(seq["color"]).pin_memory()
    stream = depth_mod.DepthFusionStream(model, vol, seq["K"], overlap=overlap)
    w_plane = storage[1]

    feeder = FrameFeeder(host, batch, device)  # uploads one batch ahead on a copy stream, as the headline job

    def run_steps(first, count):
        if not prefetch:  # (A/B: the upload in line on the compute stream)
            for i in range(first, first + count):
                fr = host[frame_ids(i)].to(device, non_blocking=True)
                dm = stream.step(fr, seq["poses"][frame_ids(i)])
            return fr, dm
        token = feeder.prefetch(frame_ids(first))
        for i in range(first, first + count):
            fr = feeder.acquire(token)
            nxt = feeder.prefetch(frame_ids(i + 1)) if i + 1 < first + count else None
            dm = stream.step(fr, seq["poses"][frame_ids(i)])
            feeder.release(token, stream.side)  # the buffer is free once the sweeps that read its colours are done
            token = nxt
        return fr, dm  # (the feeder buffer holding the last batch is not written again)

    def sweep_roofline(fr, dm, ids):
        """The sweep alone on these frames: launch duration (HIP events inside the library), N_upd per frame (counting kernel), N_union of the last sweep."""
        torch.cuda.synchronize()  # (the frames / depth maps were made on torch's stream, the volume's kernels run on its own)
        n_upd = [vol.integrate(fr[j], dm[j], seq["K"], seq["poses"][i], return_n_updated=True) for j, i in enumerate(ids)]
        vol.reset()
        torch.cuda.synchronize()
        vctx.set_timing(True)
        vol.integrate_batch(fr, dm, seq["K"], seq["poses"][ids])
        vctx.synchronize()
        n_launch, k_ms = vctx.kernel_time_total()
        vctx.set_timing(False)
        groups = vol.last_batch_groups()
        wl = vol.last_sweep_voxels()
        w_before = w_plane.clone()
        torch.cuda.synchronize()  # (the copy runs on torch's stream, the sweep below on the volume's)
        nf = groups[-1]
        vol.integrate_batch(fr[-nf:], dm[-nf:], seq["K"], seq["poses"][ids[-nf:]])
        vctx.synchronize()
        n_union = int((w_plane != w_before).sum().item())
        del w_before
        launch_us = k_ms / max(n_launch, 1) * 1e3
        fpl = len(ids) / max(len(groups), 1)
        serial = (24.0 * float(np.mean(n_upd)) + 8.0 * H * W) * fpl
        must = 24.0 * n_union + 8.0 * H * W * nf
        return {"kernel": "integrate_multi_kernel", "bound": "hbm", "achieved": serial / (launch_us * 1e-6) / 1e9, "peak": 8000.0, "unit": "GB/s",
                "frac": serial / (launch_us * 1e-6) / 1e9 / 8000.0, "algorithmic_bytes_per_launch": serial, "frames_per_launch": fpl,
                "avg_launch_us": launch_us, "us_per_frame": launch_us / fpl, "n_upd_mean": float(np.mean(n_upd)),
                "must_move": {"bytes_per_launch": must, "gbs": must / (launch_us * 1e-6) / 1e9, "frac": must / (launch_us * 1e-6) / 1e9 / 8000.0,
                              "n_union_last_sweep": n_union},
                "worklist": {"voxels_last_sweep": wl, "updated_share": n_union / max(wl, 1)}}

    with torch.no_grad():
        run_steps(0, 1)
        stream.join()
        torch.cuda.synchronize()
        vol.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fr, dm = run_steps(1, steps)
        stream.join()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        timed_frames = vol.stats()[0]  # (frames the volume was handed since the reset in front of the timed steps)
        weight_sum = int(w_plane.double().sum().item())
        fr_step = fr
        ids = frame_ids(steps)[:8]  # (eight frames of the last step: two sweeps of four)
        fr, dm = fr[:8].contiguous(), dm[:8].contiguous()
        roof = sweep_roofline(fr, dm, ids)
        roof["scene"] = "DPT-Large depth (seeded weights) of the last step's frames"
        roof_room = sweep_roofline(fr, torch.from_numpy(seq["depth"][ids]).to(device), ids)
        roof_room["scene"] = "analytic ray-cast depth of the same frames (the room's walls inside the volume)"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2):
            stream.depth(fr_step)
        e1.record()
        e1.synchronize()
        dpt_ms = e0.elapsed_time(e1) / 2
    out = {"workload": f"synthetic {W}x{H} RGB (room trajectory, 2.4 degrees per frame), DPT-Large (vitl16_384, bf16, seeded weights) at {net_w}x{net_h} + "
                       f"{'x'.join(str(int(d)) for d in vol.vol_dim)} TSDF integrate, {steps} steps of {batch} frames, uploads in the timed region",
           "value": steps * batch / elapsed, "unit": "frames/s", "ms_per_step": elapsed / steps * 1e3, "frames_per_step": batch, "steps": steps,
           "dpt_ms_per_frame": dpt_ms / batch, "network_size": [net_h, net_w], "resize_method": "minimal", "tsdf_overlap": overlap,
           "frames_integrated": timed_frames, "weight_sum": weight_sum, "unique_frames": T, "roofline": roof, "roofline_room": roof_room}
    vol.close()
    del model, vol, storage, stream, w_plane
    torch.cuda.empty_cache()
    return out


def self_launch(n_ranks):
    """`python bench.py --gpus N` without a launcher: start N FRESH rank processes with torch.distributed.run (rendezvous on 127.0.0.1, a free
    port) and hand their exit status on.  Runs before this process has touched the GPU or the library -- nothing that has initialised HIP is ever
    re-exec'ed; the ranks are ordinary children and rank 0 prints the line on the stdout they inherit."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: RCCL's device-memory exchange needs it on these hosts)
    sys.stdout.flush()
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    from hive_amd import _lib, depth as depth_mod, distributed as hdist, fusion, synthetic

    rank, world, local_rank = hdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} (or with no launcher at all)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback); the CPU baseline is only the comparison leg")
    dev_index = local_rank % torch.cuda.device_count()  # one GPU per rank on a node; wraps only in 1-GPU rehearsals
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    torch.backends.cudnn.benchmark = True

    H, W, B, T = 480, 640, args.batch, args.frames
    exact = world > 1 and args.merge == "exact"
    headline_strong = world > 1 and (args.scaling == "strong" or exact)  # (exact: every rank integrates every frame -- fixed total work by construction)
    extra_legs = not (args.no_extra_legs or args.timed_only)
    # the synthetic sequence: the same one on every rank (the job is a stretch of its wrapping frame index)
    seq = synthetic.make_sequence(num_frames=T, height=H, width=W, seed=1234, yaw_step_deg=360.0 / T)
    K = seq["K"]
    poses = seq["poses"]
    frames_host = torch.from_numpy(seq["color"]).pin_memory()  # uint8 [T, H, W, 3]

    ctx = _lib.default_context(dev_index)
    overlap = not args.no_overlap and args.merge != "exact"  # (exact mode integrates after the all-gather: nothing to overlap)
    vctx = depth_mod.DepthFusionStream.side_stream_context(dev_index) if overlap else ctx  # the timed volume's context (and stream)
    torch_dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}
    if exact:
        merger = hdist.ExactSlabFusion(synthetic.room_bounds(), args.voxel, ctx=ctx)
        volume = merger.slab
    else:
        merger = None
        volume = fusion.TSDFVolume(synthetic.room_bounds(), args.voxel, ctx=vctx)
    feeder = FrameFeeder(frames_host, B, device)

    def make_stream(dtype_name, vol=None, ovl=None):
        # seeded non-degenerate weights (hive_amd/dpt/init.py): PyTorch's default initialisation predicts a constant 7.25 m, i.e. a
        # TSDF scene with no surface (free space only); these give depth maps of 1-7 m with surfaces inside the volume
        model = depth_mod.build_model(None, device=device, dtype=torch_dtype[dtype_name], engine=args.engine, init_seed=1234)
        return depth_mod.DepthFusionStream(model, vol or volume, K, overlap=overlap if ovl is None else ovl)

    # strong: the job is frames [0, K B) of the wrapping sequence, this rank takes its contiguous share of every stretch asked for;
    # weak: the job is frames [0, N (W + K) B), this rank owns the contiguous block [rank (W + K) B, + (W + K) B) (W warm-up steps first)
    def job_frames(strong, first_step, n_steps):
        if strong:
            ids = [(first_step * B + j) % T for j in range(n_steps * B)]
            lo, hi = hdist.shard_range(len(ids), rank, world)
            return ids[lo:hi], [b - a for a, b in (hdist.shard_range(len(ids), r, world) for r in range(world))]
        base = (rank * (args.warmup + args.steps) + first_step) * B
        ids = [(base + j) % T for j in range(n_steps * B)]
        return ids, [len(ids)] * world

    def batches(ids):
        """At most B frames per batch, in equal parts (60 frames -> 30 + 30, not 48 + 12: small batches fill the chip worse)."""
        n_b = max(1, -(-len(ids) // B))
        cuts = [len(ids) * i // n_b for i in range(n_b + 1)]
        return [ids[a:b] for a, b in zip(cuts[:-1], cuts[1:]) if b > a]

    def run_job(stream, strong, first_step, n_steps):
        """All of this rank's batches: upload (one batch ahead) -> depth -> integrate (or, exact mode, keep the depth maps)."""
        ids, counts = job_frames(strong, first_step, n_steps)
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
                stream.step(fr, poses[batch_ids])
            feeder.release(token, stream.side if not exact else None)  # the buffer is free once the sweeps that read its colours are done
            token = nxt
        if exact:  # all-gather the frames, integrate every frame of the job in sequence order into this rank's x-slab
            color = torch.cat([c for c, _ in kept]) if kept else torch.empty((0, H, W, 3), dtype=torch.uint8, device=device)
            depth = torch.cat([d for _, d in kept]) if kept else torch.empty((0, H, W), dtype=torch.float32, device=device)
            all_ids = [(first_step * B + j) % T for j in range(n_steps * B)]
            merger.integrate(color, depth, K, poses[all_ids], counts)

    def volume_check(vol, merged, frames_mine):
        """Proof of work, read from the TIMED volume right after the clock stopped: how many frames / integrate launches this rank's volume was handed since
        the reset in front of the job (hive_tsdf_stats) and its weight plane's maximum and sum.  Every update adds the observation weight 1, so the sum
        is exactly the number of (frame, voxel) updates the job made: sum over the job's frames of N_upd -- `expected_weight_sum` is added further down
        from the counting kernel's N_upd of exactly those frames (an independent volume, the single-frame kernel).  At N > 1 the sums of the ranks'
        volumes are in the merged volume (reduce-scatter of the sums), so weight_sum is already the whole job's; frames are summed over the ranks."""
        frames, launches = vol.stats()
        w = (merged if merged is not None else vol).device_tensors()[1]
        chk = {"frames_integrated": frames, "integrate_launches": launches, "weight_max": float(w.max().item()), "weight_sum": int(w.double().sum().item()),
               "voxels_observed": int((w > 0).sum().item())}
        del w
        if world > 1:
            t = torch.tensor([float(frames), float(launches)], dtype=torch.float64, device="cpu" if hdist._host_staged() else device)
            per_rank = [torch.zeros_like(t) for _ in range(world)]
            torch.distributed.all_gather(per_rank, t)
            chk["frames_per_rank"] = [int(x[0].item()) for x in per_rank]
            chk["launches_per_rank"] = [int(x[1].item()) for x in per_rank]
            chk["frames_integrated"] = sum(chk["frames_per_rank"])
            chk["integrate_launches"] = sum(chk["launches_per_rank"])
            if exact:  # every rank's slab saw every frame: the job's frames are one rank's count
                chk["frames_integrated"] = chk["frames_per_rank"][0]
        chk["frames_expected"] = frames_mine
        return chk

    def timed_job(stream, strong, vol=None):
        """W untimed warm-up steps, then EXACTLY K timed steps between barrier + synchronize on both sides; at N > 1 the one merge of the shared volume is
        inside the timed region.  Returns (max-over-ranks seconds, frames of the whole job, sweep launches, sweep kernel ms of this rank, proof-of-work
        record of the timed volume)."""
        vol = vol or volume
        tctx = vol._ctx
        for s in range(0, args.warmup):
            run_job(stream, strong, s, 1)
        if strong and args.warmup > 0:  # the timed job's own batch sizes (this rank's share of K * B frames) also run once untimed:
            warmed = {len(b) for s in range(args.warmup) for b in batches(job_frames(strong, s, 1)[0])}  # first use sizes the activation arena
            for size in sorted({len(b) for b in batches(job_frames(strong, args.warmup, args.steps)[0])} - warmed):
                tok = feeder.prefetch(list(range(size)))
                stream.depth(feeder.acquire(tok))
                feeder.release(tok)
        if world > 1 and args.warmup > 0:  # warm-up of the collectives too (RCCL sets up its channels on the first large transfer)
            if exact:
                merger.gather()
            else:
                stream.join()
                hdist.fuse_sharded(vol)
        torch.cuda.synchronize()
        vol.reset()  # the timed job starts from an empty scene
        torch.cuda.synchronize()
        tctx.set_timing(True)
        hdist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if strong:
            run_job(stream, True, args.warmup, args.steps)
        else:
            for s in range(args.steps):
                run_job(stream, False, args.warmup + s, 1)
        stream.join()  # the merge (and the end of the job) behind the last sweeps
        merged = None
        if world > 1:
            merged = merger.gather() if exact else hdist.fuse_sharded(vol)
        torch.cuda.synchronize()
        hdist.barrier()
        elapsed = time.perf_counter() - t0
        n_launch, kernel_ms = tctx.kernel_time_total()
        tctx.set_timing(False)
        elapsed = hdist.max_over_ranks(elapsed, device=device if world > 1 else "cpu")
        total = args.steps * B * (1 if strong or world == 1 else world)
        check = volume_check(vol, merged if exact else None, total)
        del merged
        return elapsed, total, n_launch, kernel_ms, check

    stream = make_stream(args.dtype)
    elapsed, total_frames, n_launch, kernel_ms, volume_proof = timed_job(stream, headline_strong)

    if args.timed_only:
        if rank == 0:
            print(json.dumps({"value": total_frames / elapsed, "unit": "frames/s", "ms_per_step": elapsed / args.steps * 1e3,
                              "avg_integrate_us": kernel_ms / max(n_launch, 1) * 1e3, "integrate_launches": n_launch, "tsdf_overlap": overlap,
                              "timed_volume_check": volume_proof, "note": "--timed-only: no roofline / cpu_baseline / extra legs"}))
        return

    # ---- the other scaling mode at N > 1 (BASELINE configs[2] literally when the headline is the weak job) -------------------------------------
    other = None
    if world > 1 and extra_legs and not exact:
        o_elapsed, o_total, _, _, o_check = timed_job(stream, not headline_strong)
        other = {"scaling": "weak" if headline_strong else "strong", "value": o_total / o_elapsed, "unit": "frames/s", "frames_total": o_total,
                 "ms_total": o_elapsed * 1e3, "steps": args.steps, "frames_per_step": B, "timed_volume_check": o_check,
                 "note": ("BASELINE configs[2] literally: the SAME K x B frames split in contiguous blocks over the ranks, one merge of the shared volume inside the timed "
                          "region; value / (the N = 1 line's value) is the strong-scaling speed-up" if not headline_strong else
                          "N x K x B frames, a contiguous block of K steps per rank, one merge inside the timed region")}

    # ---- untimed: what the timed launches processed --------------------------------------------------------------------
    # A measurement volume of the same shape.  Per frame: N_upd, from the counting variant of the single-frame kernel.  Per sweep: N_union = voxels
    # updated by ANY frame of the sweep = weights that moved across the launch (every update adds obs_weight > 0), and the work list's length (the voxels
    # the sweep ran its tests on).  All depend on depth + pose only, not on the volume's state.
    x_range = merger.x_ranges[rank] if exact else None
    n_vox = volume.num_voxels
    storage = tuple(torch.empty(n_vox, dtype=torch.float32, device=device) for _ in range(3))  # caller-owned planes: the weight plane is read directly
    mvol = fusion.TSDFVolume(synthetic.room_bounds(), args.voxel, ctx=ctx, storage=storage, x_range=x_range)
    w_plane = storage[1]

    def measure(frame_sets, depth_of, time_kernel=False, replay_sets=4):
        """Over all frames: per-frame N_upd (the counting variant of the single-frame kernel); over the first `replay_sets` sets also: per-sweep (frames,
        N_union, work-list voxels), ms per frame of the TSDF leg (prep + work list + sweep), and with time_kernel the HIP-event time of the sweep launches."""
        n_upd, sweeps, leg_ms, k_ms, k_n = [], [], [], 0.0, 0
        for si, ids in enumerate(frame_sets):
            fr = torch.from_numpy(seq["color"][ids]).to(device)
            depth_m = depth_of(fr, ids)
            for j, i in enumerate(ids):
                n_upd.append(mvol.integrate(fr[j], depth_m[j], K, poses[i], return_n_updated=True))
            mvol.reset()  # (also: every weight a whole number again -- the sweep's fast colour update, as in the timed job)
            if si >= replay_sets:
                continue
            if time_kernel:
                ctx.set_timing(True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            mvol.integrate_batch(fr, depth_m, K, poses[ids])  # as the timed job runs it
            e1.record()
            e1.synchronize()
            leg_ms.append(e0.elapsed_time(e1) / len(ids))
            if time_kernel:
                n, ms = ctx.kernel_time_total()
                ctx.set_timing(False)
                k_ms, k_n = k_ms + ms, k_n + n
            a = 0
            for nf in mvol.last_batch_groups():  # the same sweeps again, one call each, with the weight plane compared across the launch
                before = w_plane.clone()
                mvol.integrate_batch(fr[a:a + nf], depth_m[a:a + nf], K, poses[ids[a:a + nf]])
                assert mvol.last_batch_groups() == [nf]
                sweeps.append((nf, int((w_plane != before).sum().item()), mvol.last_sweep_voxels()))
                a += nf
        return n_upd, sweeps, float(np.mean(leg_ms)), (k_ms / max(k_n, 1) * 1e3, k_n)

    stamp = source_stamp()

    def roofline(n_upd, sweeps, launch_us, scene_key):
        """HBM roofline of the sweep kernel.  `achieved` = SURVEY 8(d)'s ALGORITHMIC bytes per launch -- the per-frame figure 24 N_upd + 8 H W (every voxel a
        frame updates read and written once in three float planes, one read of the frame's depth + colour) x the frames one launch integrates --
        / the launch duration.  Four frames share one sweep, so a voxel all four update moves once where the per-frame figure counts it four times: the
        bytes the fused launch MUST move (24 N_union + 8 H W x frames) are reported beside it as `must_move` -- the stricter figure."""
        frames = sum(s[0] for s in sweeps)
        fpl = frames / max(len(sweeps), 1)
        n_union = float(np.mean([s[1] for s in sweeps]))
        wl = float(np.mean([s[2] for s in sweeps]))
        n_upd_mean = float(np.mean(n_upd))
        must = 24.0 * n_union + 8.0 * H * W * fpl
        alg = (24.0 * n_upd_mean + 8.0 * H * W) * fpl
        achieved = alg / (launch_us * 1e-6) / 1e9
        out = {"kernel": "integrate_multi_kernel" if fpl > 1 else "integrate_kernel", "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
               "frac": achieved / 8000.0, "traffic": None, "algorithmic_bytes_per_launch": alg,
               "algorithmic_bytes_note": "SURVEY 8(d): (24 N_upd + 8 H W) per frame x frames per launch",
               "frames_per_launch": fpl, "avg_launch_us": launch_us, "us_per_frame": launch_us / fpl, "n_upd_mean": n_upd_mean, "n_upd_fraction": n_upd_mean / n_vox,
               "must_move": {"bytes_per_launch": must, "gbs": must / (launch_us * 1e-6) / 1e9, "frac": must / (launch_us * 1e-6) / 1e9 / 8000.0, "n_union_mean": n_union,
                             "note": "bytes a fused launch must move: 24 N_union + 8 H W x frames (a voxel updated by several of the launch's frames moves once)"},
               "worklist": {"voxels_per_launch": wl, "updated_share": n_union / max(wl, 1.0),
                            "note": "voxels on the sweep's work list (64-voxel segments that survive the per-row frustum / depth clip): every one runs every frame's projection, "
                                    "texel gather and tests; updated_share = N_union / that"}}
        pmc = load_profile("r05_integrate_pmc.json") or load_profile("r04_integrate_pmc.json")
        if pmc and scene_key in pmc:
            if pmc.get("source_stamp") != stamp:
                out["traffic_note"] = (f"profiles/r0x_integrate_pmc.json was measured on TSDF kernel source {pmc.get('source_stamp')}, this build is {stamp}: "
                                       f"counter figures withheld (refresh the profile: tools/profile_r05.sh)")
            else:
                t = pmc[scene_key]
                out["traffic"] = t.get("hbm_bytes_per_launch")
                out["traffic_detail"] = {k: t.get(k) for k in ("read_bytes_lo", "read_bytes_hi", "write_bytes", "launches", "note") if k in t}
                v = t.get("valu")
                if v:
                    issue_us = v["SQ_INSTS_VALU"] * 4.0 / 1024.0 / (v["clock_ghz"] * 1e3)
                    out["valu_issue"] = {"insts_per_launch": v["SQ_INSTS_VALU"], "cycles_per_inst": 4, "simds": 1024, "clock_ghz": v["clock_ghz"], "min_us": issue_us,
                                         "frac_of_launch": issue_us / launch_us, "simd_busy_valu": v.get("simd_busy_valu"),
                                         "note": "profiles/r05_integrate_pmc.json (rocprofv3 --pmc of the same kernel source and scene)"}
        return out

    with torch.no_grad():
        if exact:
            timed_sets = batches([(args.warmup * B + j) % T for j in range(args.steps * B)])
        elif headline_strong:
            timed_sets = batches(job_frames(True, args.warmup, args.steps)[0])
        else:
            timed_sets = [job_frames(False, args.warmup + s, 1)[0] for s in range(args.steps)]
        # N_upd of EVERY frame of the timed job (this rank's): their sum is what the timed volume's weight plane must add up to; the sweep replays (N_union, work
        # list, kernel time) over the first four sets only (the sequence wraps: a few steps cover every frame of it)
        n_upd_job, sweeps_dpt, leg_dpt, (us_dpt, _) = measure(timed_sets, lambda fr, ids: stream.depth(fr)[0], time_kernel=True, replay_sets=4)
        n_upd_dpt = n_upd_job[:sum(len(t) for t in timed_sets[:4])]
        expected = float(sum(n_upd_job))
        if world > 1:  # (every rank counted its own frames -- exact mode: every frame on its own slab; the job's total is the sum)
            t = torch.tensor([expected], dtype=torch.float64, device="cpu" if hdist._host_staged() else device)
            torch.distributed.all_reduce(t)
            expected = float(t.item())
        volume_proof["expected_weight_sum"] = int(expected)
        volume_proof["expected_note"] = ("sum over the timed job's frames of N_upd, counted by the single-frame kernel's counting variant on a separate volume from the same "
                                         "depth maps (same batches through the network) and poses; every update adds weight 1, so the timed volume's weight plane must "
                                         "sum to exactly this")
        w_ok = volume_proof["weight_sum"] == volume_proof["expected_weight_sum"]
        if exact:  # (the ranks' depth maps come from other batch sizes than this recount's: split-K / Gram thresholds move a few last bits of a few pixels)
            w_ok = abs(volume_proof["weight_sum"] - volume_proof["expected_weight_sum"]) <= 1e-3 * max(volume_proof["expected_weight_sum"], 1)
        volume_proof["pass"] = bool(w_ok and volume_proof["frames_integrated"] == volume_proof["frames_expected"])
        # the launch duration the roofline uses is the kernel's OWN (HIP events, the same frames swept again with nothing else on the GPU);
        # inside the timed job the sweeps share the chip with the next batch's network (second stream, lowest priority) and take longer
        main_roof = roofline(n_upd_dpt, sweeps_dpt, us_dpt, "bench")
        main_roof["launches"] = n_launch
        main_roof["avg_launch_us_in_job"] = kernel_ms / max(n_launch, 1) * 1e3
        main_roof["frac_in_job"] = main_roof["frac"] * main_roof["avg_launch_us"] / max(main_roof["avg_launch_us_in_job"], 1e-9)
        main_roof["must_move"]["frac_in_job"] = main_roof["must_move"]["frac"] * main_roof["avg_launch_us"] / max(main_roof["avg_launch_us_in_job"], 1e-9)
        main_roof["overlap"] = ("the timed job runs the sweeps of batch i on a second, lowest-priority HIP stream under the network of batch i + 1: "
                                "avg_launch_us_in_job is their duration there, avg_launch_us the kernel alone") if overlap else "none (--no-overlap / exact merge): the sweeps run on the network's stream"
        main_roof["tsdf_leg_us_per_frame"] = leg_dpt * 1e3
        main_roof["scene"] = "DPT-Hybrid depth (seeded weights) of the timed frames"
        # the room scene: the same integrate on the ray-cast (analytic) depth of the sequence: CONSECUTIVE frames (2.4 degrees apart,
        # sweeps of four, as in the timed job), the walls of the room as the surface
        room_ids = list(range(min(32, T)))
        mvol.reset()
        n_upd_room, sweeps_room, leg_room, (us_room, n_room) = measure([room_ids], lambda fr, ids: torch.from_numpy(seq["depth"][ids]).to(device), time_kernel=True)
        room_roof = roofline(n_upd_room, sweeps_room, us_room, "room")
        room_roof["launches"] = n_room
        room_roof["tsdf_leg_us_per_frame"] = leg_room * 1e3
        room_roof["scene"] = f"analytic ray-cast depth of the same trajectory (the room's walls inside the volume), frames 0..{len(room_ids) - 1} consecutively"

        # ---- DPT alone: MFMA roofline of the network (hive_dpt_forward on one step's batch, HIP events on the compute stream) ---------
        from hive_amd.dpt.models import count_flops
        fr = torch.from_numpy(seq["color"][[j % T for j in range(len(timed_sets[0]))]]).to(device)
        stream.depth(fr)
        reps = 3
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            stream.depth(fr)
        e1.record()
        e1.synchronize()
        dpt_ms = e0.elapsed_time(e1) / reps
        flops = count_flops(H, W)
        tflops = flops["total"] * fr.shape[0] / (dpt_ms * 1e-3) / 1e12
        dpt_roof = {"kernel": "hive_dpt_forward (MFMA GEMM / attention / implicit-GEMM convolutions + glue, one C-ABI call)", "bound": "mfma", "achieved": tflops, "peak": 2500.0,
                    "unit": "TFLOP/s", "frac": tflops / 2500.0, "dtype": args.dtype, "flops_per_frame": flops["total"], "frames": int(fr.shape[0]), "ms_per_batch": dpt_ms,
                    "ms_per_frame": dpt_ms / fr.shape[0], "note": "algorithmic FLOPs (2 x MACs, hive_amd.dpt.models.count_flops) x frames / time of the whole network, "
                    "glue kernels included; per-kernel mfma_busy: profiles/r05_mfma_pmc.json"}
        busy = load_profile("r05_mfma_pmc.json") or load_profile("r04_mfma_pmc.json")
        if busy:
            dpt_roof["mfma_busy"] = busy.get("mfma_busy")

        # ---- marching cubes once, on the room volume (reported separately; runs once per sequence) ----------------------------------------
        mesh = None
        if not exact:
            mvol._extract()  # (a first call sizes the scratch and result buffers)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            n_v, n_f = mvol._extract()
            e1.record()
            e1.synchronize()
            mc_bytes = 4.0 * n_vox + 36.0 * n_v + 12.0 * n_f + 4.0 * n_v
            mesh = {"mesh_ms": e0.elapsed_time(e1), "vertices": n_v, "faces": n_f, "algorithmic_bytes": mc_bytes, "gbs": mc_bytes / (e0.elapsed_time(e1) * 1e-3) / 1e9,
                    "frac_of_hbm_peak": mc_bytes / (e0.elapsed_time(e1) * 1e-3) / 1e9 / 8000.0,
                    "note": "hive_tsdf_extract_mesh on the room volume after the 32 frames above, end to end on the device side (its count read-back included; scratch and "
                            "result buffers reused from the previous call); bytes = 4 N + 36 V + 12 F + 4 V (SURVEY 8d).  Runs once per sequence."}
    mvol.close()
    del mvol, w_plane, storage

    # ---- the same job with the sweeps on the network's stream (N = 1): the overlap's A/B inside the one line -------------------------------------
    no_overlap = None
    if world == 1 and extra_legs and overlap and args.engine == "hip":
        vol2 = fusion.TSDFVolume(synthetic.room_bounds(), args.voxel, ctx=ctx)
        stream2 = make_stream(args.dtype, vol=vol2, ovl=False)
        n_elapsed, n_total, n_n, n_ms, n_check = timed_job(stream2, False, vol=vol2)
        no_overlap = {"value": n_total / n_elapsed, "ms_per_step": n_elapsed / args.steps * 1e3, "avg_launch_us_in_job": n_ms / max(n_n, 1) * 1e3,
                      "weight_sum": n_check["weight_sum"], "frames_integrated": n_check["frames_integrated"]}
        vol2.close()
        del stream2, vol2
        torch.cuda.empty_cache()

    # ---- the reference's dtype: the same timed job in float16 (N = 1) ------------------------------------------------------------------------
    fp16 = None
    if world == 1 and extra_legs and args.engine == "hip":
        other_dtype = "fp16" if args.dtype == "bf16" else "bf16"
        del stream
        torch.cuda.empty_cache()
        stream = make_stream(other_dtype)
        f_elapsed, f_total, _, _, _ = timed_job(stream, False)
        fp16 = {"dtype": other_dtype, "value": f_total / f_elapsed, "ms_per_step": f_elapsed / args.steps * 1e3}
        # the reference's literal call pattern is ONE frame per forward (hive/dataset_adaptors.py:1406-1419): the drop-in network object at batches of 1 and 8,
        # device-resident frames, in the reference's dtype (the whole sweep: tools/batch_sweep.py -> profiles/r04_batch_sweep.json)
        small_batch = {}
        fmodel = stream.model
        few = torch.from_numpy(seq["color"][:8]).to(device)
        with torch.no_grad():
            for b in (1, 8):
                fr = few[:b].contiguous()
                for _ in range(3):
                    fmodel.forward_frames(fr, max_depth=10.0)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fmodel.forward_frames(fr, max_depth=10.0)
                e1.record()
                e1.synchronize()
                small_batch[f"batch_{b}"] = {"ms_per_frame": e0.elapsed_time(e1) / 20 / b, "frames_per_s": b * 20 / (e0.elapsed_time(e1) * 1e-3)}
        small_batch["dtype"] = other_dtype
        small_batch["note"] = "hive_dpt_forward alone (DPT-Hybrid, 480 x 640, frames resident on the device), 20 calls after 3"
        fp16["small_batch"] = small_batch
        del fmodel, few
    del stream
    torch.cuda.empty_cache()

    # ---- BASELINE configs[3], bounded ----------------------------------------------------------------------------------------------------------
    config4 = None
    if world == 1 and extra_legs and args.engine == "hip":
        try:
            config4 = config4_leg(device, ctx)
        except (_lib.HiveError, torch.cuda.OutOfMemoryError) as e:  # the library refusing, or the 13 GB volume not fitting beside what else runs on the card: say so in
            config4 = {"error": f"{type(e).__name__}: {e}"}          # the line; anything else (a bug in this file) propagates

    if rank != 0:
        return
    dim_list = [int(d) for d in (merger.dims if exact else volume.vol_dim)]
    dims = "x".join(str(d) for d in dim_list)
    vol_name = f"{dim_list[0]}^3" if len(set(dim_list)) == 1 else dims  # (the real dims: --voxel changes them)
    merge_note = ""
    scaling = "strong" if headline_strong else "weak"  # (N = 1: the weak job's one-rank case -- per-GPU work fixed as N grows)
    if world > 1:
        share = (f"the same {args.steps} x {B} frames split in contiguous blocks over the ranks" if headline_strong
                 else f"{world} x {args.steps} x {B} frames, a contiguous block of {args.steps} steps per rank")
        merge_note = (f", {share}, frames all-gathered, every rank integrates all frames into its x-slab (bit-identical to 1 GPU), slabs all-gathered"
                      if exact else f", frame-sharded ({share}), shared volume merged once by reduce-scatter of the 5 accumulator planes + all-gather of the 3 "
                                    f"volumes (RCCL), inside the timed region")
    out = {
        "metric": f"frames/sec (depth+TSDF integrate) @{W}x{H}, {vol_name} vol" + ("" if world == 1 else f", {world} GPUs, {scaling} scaling ({total_frames} frames)"),
        "value": total_frames / elapsed,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {
            "workload": f"synthetic {W}x{H}x{T} RGB (seeded room trajectory), host-resident uint8 frames uploaded in the timed region, DPT-Hybrid depth "
                        f"(seeded random weights: depth 1-7 m, {args.dtype}, {args.engine} engine) + {dims} TSDF integrate, {B} frames/step" + merge_note,
            "frames_per_step": B, "frames_total": total_frames, "image": [H, W], "volume": dims, "voxel_m": args.voxel,
            "n_upd_mean": main_roof["n_upd_mean"], "n_upd_fraction": main_roof["n_upd_fraction"], "merge": (args.merge if world > 1 else None),
            "tsdf_overlap": overlap, "scaling_mode": scaling,
        },
        "tsdf_source_stamp": stamp,
    }
    # key order: a log TAIL shows the end of the line (the driver's tail cut the middle out) -- the side legs first, then the other scaling mode, and
    # the contract's objects last: proof of work, the dominant kernel's roofline, the CPU baseline
    if config4 is not None:
        out["config4"] = config4
    out["mesh"] = mesh
    if fp16 is not None:
        out["small_batch"] = fp16["small_batch"]
    out["roofline_room"] = room_roof
    out["roofline_dpt"] = dpt_roof
    if fp16 is not None:
        out["value_" + fp16["dtype"]] = fp16["value"]
        out["ms_per_step_" + fp16["dtype"]] = fp16["ms_per_step"]
    if no_overlap is not None:
        out["value_no_overlap"] = no_overlap["value"]
        out["no_overlap"] = no_overlap
    if other is not None:
        out[other["scaling"]] = other
    out["timed_volume_check"] = volume_proof
    out["roofline"] = main_roof
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(seq, args.voxel, K)
    print(json.dumps(compact(out, args.notes)))


if __name__ == "__main__":
    main()
