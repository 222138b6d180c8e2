#!/usr/bin/env python3
"""Launch-ordered kernel times of one DPT-Hybrid forward (hive_dpt_forward) at the bench batch.
  run <batch> <dtype>   three forwards of `batch` 480 x 640 frames; the last one lies between two marker launches (a fill of 12345 floats)
  parse <dir>           <dir>/raw/**/kernel_trace.csv -> <dir>/forward.csv + a per-section summary on stdout"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(batch, dtype):
    import torch
    from hive_amd import depth as depth_mod
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[dtype]
    model = depth_mod.build_model(None, dtype=dt, init_seed=7)
    frames = torch.randint(0, 256, (batch, 480, 640, 3), dtype=torch.uint8, device="cuda")
    marker = torch.empty(12345, device="cuda")
    for i in range(3):
        if i == 2:
            torch.cuda.synchronize()
            marker.fill_(1.0)
        with torch.no_grad():
            model.forward_frames(frames, max_depth=10.0)
        if i == 2:
            marker.fill_(2.0)
    torch.cuda.synchronize()


def short(name):
    for a, b in (("(anonymous namespace)::", ""), ("_GLOBAL__N_1", ""), ("void ", "")):
        name = name.replace(a, b)
    return name.split("(")[0][:70]


def parse(out):
    f = max(glob.glob(os.path.join(out, "raw", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # the marker is the only at::native fill in the process after warm-up: take the last two fills
    fills = [i for i, r in enumerate(rows) if "at::native" in r["Kernel_Name"] and "ill" in r["Kernel_Name"]]
    a, b = fills[-2], fills[-1]
    seq = rows[a + 1:b]
    with open(os.path.join(out, "forward.csv"), "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["index", "kernel", "us", "grid", "workgroup"])
        for i, r in enumerate(seq):
            us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            w.writerow([i, short(r["Kernel_Name"]), f"{us:.1f}", r.get("Grid_Size_X", ""), r.get("Workgroup_Size_X", "")])
    tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seq) / 1e3
    span = (int(seq[-1]["End_Timestamp"]) - int(seq[0]["Start_Timestamp"])) / 1e3
    print(f"{len(seq)} launches, {tot / 1e3:.2f} ms of kernels, {span / 1e3:.2f} ms first start -> last end")




def gaps(trace_csv, tail_ms=400.0):
    """Idle intervals of the device (no kernel of any stream running) over the last `tail_ms` of a kernel trace: where a step loses time
    to the host.  Usage: python tools/layer_trace.py gaps <kernel_trace.csv> [tail_ms]"""
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(trace_csv))))
    t_end = max(r[1] for r in rows)
    rows = [r for r in rows if r[0] >= t_end - tail_ms * 1e6]
    busy_until, idle, big = rows[0][0], 0, []
    for s, e, name in rows:
        if s > busy_until:
            idle += s - busy_until
            if s - busy_until > 20000:
                big.append(((s - busy_until) / 1e3, short(name)))
        busy_until = max(busy_until, e)
    span = (rows[-1][1] - rows[0][0]) / 1e6
    print(f"last {span:.1f} ms: idle {idle / 1e6:.2f} ms ({100 * idle / 1e6 / span:.1f} %), {len(big)} gaps > 20 us")
    for g, n in sorted(big, reverse=True)[:15]:
        print(f"  {g:8.1f} us before {n}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), sys.argv[3])
    elif sys.argv[1] == "gaps":
        gaps(sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 400.0)
    else:
        parse(sys.argv[2])
